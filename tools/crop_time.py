#!/usr/bin/env python3
"""Time the device crop alone (B sequences, frames resident): python tools/crop_time.py [B]   (VITTRACK_LIB / VT_CROP_ROWS select forms)"""
import os
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vittracker_amd import native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 480, 640
rs = np.random.RandomState(0)
frames = torch.from_numpy(rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)).cuda()
boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
st = torch.tensor(boxes, dtype=torch.float64).cuda()
m = native.Model(64, 128, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=16, len_x=64))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
for T, factor in ((128, 4.0), (256, 4.0), (64, 2.0)):
    for _ in range(5):
        m.crop(frames, st, factor, T, MEAN, STD)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        m.crop(frames, st, factor, T, MEAN, STD)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    out_mb = B * 3 * T * T * 4 / 1e6
    print(f"crop T={T} factor={factor} B={B}: {us:7.1f} us per call (eager, incl. launch), output {out_mb:.0f} MB = {out_mb / us * 1e-6 * 1e6 / 1e3:.2f} TB/s")
