import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
os.environ.setdefault("VITTRACK_PRJ_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vittracker_amd.batched import BatchedVitTracker
from vittracker_amd.parameter import vit_dist as P
p = P.parameters("vit_48_h32_g128"); p.allow_synthetic_weights = True
B, H, W = 256, 480, 640
rs = np.random.RandomState(0)
frames = rs.randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)
boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
bt = BatchedVitTracker(p, B)
bt.initialize(frames[0], boxes)
dev = torch.from_numpy(frames).cuda()
for rep in range(2):
    for n in (1, 2, 3, 4, 6, 8):
        chunk = dev[[i & 1 for i in range(n)]].contiguous()
        bt.track_chunk(chunk, sync=False); bt.track_chunk(chunk, sync=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for f in range(96 // n):
            bt.track_chunk(chunk, sync=False)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(rep, n, f"{dt / ((96 // n) * n) * 1e3:.3f} ms per frame", flush=True)
        bt._chunk_graphs.clear()
