#!/usr/bin/env python3
"""Where a plugin track() call spends its time (B = 1 device pipeline): upload, graph replay, GPU time, read-back."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("VITTRACK_PRJ_DIR", ROOT)
from vittracker_amd.parameter import vit_dist as P
from vittracker_amd.tracker.vit_dist import get_tracker_class
name = sys.argv[1] if len(sys.argv) > 1 else "vit_48_h32_noKD"
p = P.parameters(name); p.allow_synthetic_weights = True; p.debug = 0
rs = np.random.RandomState(0)
frames = [rs.randint(0, 256, (240, 320, 3)).astype(np.uint8) for _ in range(4)]
t = get_tracker_class()(p, "synthetic")
t.initialize(frames[0], {"init_bbox": [100.0, 80.0, 50.0, 40.0]})
for i in range(10): t.track(frames[i & 3])
bt = t._bt
N = 200
t0 = time.perf_counter()
for i in range(N): t.track(frames[i & 3])
print(name, "track() ms:", round((time.perf_counter() - t0) / N * 1e3, 3))
# pieces
tu = tr = tg = 0.0
for i in range(N):
    a = time.perf_counter(); fr = bt._upload(frames[i & 3][None]); b = time.perf_counter()
    g = bt._chunk_graph(fr.unsqueeze(0), to_host=True)[0]; g.replay(); c = time.perf_counter()
    torch.cuda.synchronize(); d = time.perf_counter()
    tu += b - a; tr += c - b; tg += d - c
print("upload ms", round(tu / N * 1e3, 3), " replay call ms", round(tr / N * 1e3, 3), " wait for GPU ms", round(tg / N * 1e3, 3))
