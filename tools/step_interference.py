#!/usr/bin/env python3
"""Does the crop slow the network kernels that follow it?  Three captured loops at B = 256 (G128), same weights and patches:
A: forward_u8(None, patch) x 4;  B: [crop_u8 -> forward_u8] x 4;  C: [crop_u8 into a scratch buffer -> forward_u8 on a FIXED patch] x 4.
Run under rocprofv3 --kernel-trace --stats to read the block kernel's duration in each (MODE=A|B|C selects one loop per process)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vittracker_amd import native, synth
mode = os.environ.get("MODE", "A")
geom = int(os.environ.get("GEOM", "128"))
B, H, W = 256, 480, 640
rs = np.random.RandomState(0)
frames = torch.from_numpy(rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)).cuda()
boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
st = torch.tensor(boxes, dtype=torch.float64).cuda()
m = native.Model(geom // 2, geom, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=(geom // 32) ** 2, len_x=(geom // 16) ** 2))
z = torch.from_numpy(synth.synth_inputs(0, B, geom // 2, geom)[0]).cuda()
m.set_template(z)
patch = torch.empty(B, geom, geom, 3, dtype=torch.uint8, device="cuda")
scratch = torch.empty_like(patch)
rf = torch.empty(B, dtype=torch.float64, device="cuda")
m.crop_u8(frames, st, 4.0, geom, out=patch, resize_factor=rf)
out = native.Outputs(B, geom // 16, "cuda")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=side):
    s = torch.cuda.current_stream()
    for _ in range(4):
        if mode == "B": m.crop_u8(frames, st, 4.0, geom, out=patch, resize_factor=rf, stream=s)
        if mode == "C": m.crop_u8(frames, st, 4.0, geom, out=scratch, resize_factor=rf, stream=s)
        m.forward_u8(None, patch, out=out, stream=s)
torch.cuda.current_stream().wait_stream(side)
for _ in range(5): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"MODE {mode} G{geom}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us per step")
