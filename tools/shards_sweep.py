#!/usr/bin/env python3
"""The whole tracker step (open-loop held boxes, 4 frames per graph launch) with N shards of 256 sequences on N streams, stepped
round-robin: python tools/shards_sweep.py [G128|G256] -- frames/s by N (one shard's crop runs beside the others' networks)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
os.environ.setdefault("VITTRACK_PRJ_DIR", ROOT)
import numpy as np
import torch
from vittracker_amd.batched import BatchedVitTracker
from vittracker_amd.parameter import vit_dist as P

geom = sys.argv[1] if len(sys.argv) > 1 else "G128"
p = P.parameters({"G128": "vit_48_h32_g128", "G256": "vit_48_h32_noKD"}[geom])
p.allow_synthetic_weights = True
p.debug = 0
B, H, W, frames = 256, 480, 640, 480
rs = np.random.RandomState(0)
fr = torch.from_numpy(rs.randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)).cuda()
boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
chunk = fr[[0, 1, 0, 1]].contiguous()
for N in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(N)]
    bts = []
    for s in streams:
        with torch.cuda.stream(s):
            bt = BatchedVitTracker(p, B)
            bt.initialize(fr[0], boxes)
            bt.hold_states(True)
            bts.append(bt)
    torch.cuda.synchronize()
    n = frames // 4
    for i in range(N * (8 + n)):
        if i == 8 * N:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        with torch.cuda.stream(streams[i % N]):
            bts[i % N].track_chunk(chunk, sync=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{geom} {N} shard(s) of {B}: {N * B * n * 4 / dt:,.0f} frames/s, {dt / (N * n * 4) * 1e6:.2f} us per frame of one shard", flush=True)
    del bts
