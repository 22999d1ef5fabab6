#!/usr/bin/env python3
"""Plugin track(): the frame upload (blocking copy from pageable memory, ~22 us for 230 KB) against a CPU copy into pinned host
memory that the crop kernel then reads over the bus (no DMA, no staging): host time per frame and crop kernel time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vittracker_amd import native
H, W, T = 240, 320, 128
rs = np.random.RandomState(0)
frames = [rs.randint(0, 256, (1, H, W, 3)).astype(np.uint8) for _ in range(4)]
dev = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
pin = torch.empty((1, H, W, 3), dtype=torch.uint8).pin_memory()
pin_np = pin.numpy()
N = 300
for name, fn in (("copy_ pageable -> device (blocking)", lambda a: dev.copy_(torch.from_numpy(a))),
                 ("np.copyto -> pinned", lambda a: np.copyto(pin_np, a)),
                 ("np.copyto -> pinned, 60 % of the rows", lambda a: np.copyto(pin_np[:, 40:184], a[:, 40:184]))):
    for i in range(20): fn(frames[i & 3])
    t0 = time.perf_counter()
    for i in range(N): fn(frames[i & 3])
    print(f"{name:45s} {(time.perf_counter() - t0) / N * 1e6:7.1f} us per frame")
m = native.Model(64, T, max_batch=1)
states = torch.tensor([[100.0, 80.0, 50.0, 40.0]], dtype=torch.float64, device="cuda")
x = torch.empty(1, 3, T, T, device="cuda"); rf = torch.empty(1, dtype=torch.float64, device="cuda")
mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for name, src in (("device frame", dev), ("pinned host frame (zero copy)", pin)):
    src.copy_(torch.from_numpy(frames[0])) if src is dev else np.copyto(pin_np, frames[0])
    torch.cuda.synchronize()
    for _ in range(5): m.crop(src, states, 4.0, T, mean, std, out=x, resize_factor=rf)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): m.crop(src, states, 4.0, T, mean, std, out=x, resize_factor=rf)
    e1.record(); e1.synchronize()
    print(f"crop kernel from {name:32s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (back to back)")
    if ref is None: ref = x.clone()
    else: print("same crop:", torch.equal(ref, x))
