#!/usr/bin/env python3
"""Host frames -> device: an explicit pinned staging buffer (CPU fill + async DMA) against one blocking copy from the caller's
pageable array, for one 230 KB frame and a 236 MB batch.  (BatchedVitTracker._upload uses the blocking copy.)"""
import time
import numpy as np
import torch

def t(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for shape, n in (((1, 240, 320, 3), 200), ((256, 480, 640, 3), 10)):
    a = np.random.randint(0, 256, shape).astype(np.uint8)
    dev = torch.empty(shape, dtype=torch.uint8, device="cuda")
    pin = torch.empty(shape, dtype=torch.uint8).pin_memory()
    def staged():
        pin.copy_(torch.from_numpy(a)); dev.copy_(pin, non_blocking=True); torch.cuda.current_stream().synchronize()
    def direct():
        dev.copy_(torch.from_numpy(a))
    for name, fn in (("pinned staging", staged), ("blocking copy from pageable", direct)):
        dt = t(fn, n)
        print(f"{a.nbytes / 1e6:8.2f} MB  {name:30s} {dt * 1e3:8.3f} ms  {a.nbytes / dt / 1e9:6.1f} GB/s")
