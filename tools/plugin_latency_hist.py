import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
os.environ.setdefault("VITTRACK_PRJ_DIR", "/root/repo")
from vittracker_amd.parameter import vit_dist as P
from vittracker_amd.tracker.vit_dist import get_tracker_class
name = sys.argv[1]
p = P.parameters(name); p.allow_synthetic_weights = True; p.debug = 0
rs = np.random.RandomState(0)
frames = [rs.randint(0, 256, (240, 320, 3)).astype(np.uint8) for _ in range(4)]
t = get_tracker_class()(p, "synthetic")
t.initialize(frames[0], {"init_bbox": [100.0, 80.0, 50.0, 40.0]})
for i in range(10): t.track(frames[i & 3])
ts = []
for i in range(2000):
    a = time.perf_counter(); t.track(frames[i & 3]); ts.append(time.perf_counter() - a)
ts = np.array(ts) * 1e3
print(name, os.environ.get("VT_BLOCKS_STACK"), "median %.3f mean %.3f max %.3f ms; calls > 1 ms: %d; > 10 ms: %d" % (np.median(ts), ts.mean(), ts.max(), (ts > 1).sum(), (ts > 10).sum()), "worst idx", np.argsort(ts)[-5:])
