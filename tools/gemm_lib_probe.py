#!/usr/bin/env python3
"""What the vendor library (hipBLASLt / rocBLAS through torch) reaches on the ViT-Base GEMM shapes of BASELINE config 4 (bf16 in,
f32 accumulate): the yardstick for vb_gemm.h.  M = 256 frames x 320 tokens."""
import torch
M = 256 * 320
shapes = [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]
dev = "cuda"
for name, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for fn_name, fn in (("linear+bias", lambda: torch.nn.functional.linear(x, w, b)), ("matmul", lambda: x @ w.t())):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print(f"{name:5s} {fn_name:12s} M={M} K={K} N={N}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s")
