#!/usr/bin/env python3
"""Phase breakdown of stem_fused_kernel from in-kernel s_memtime stamps (VT_DBG_STAMPS=1)."""
import os
import sys
os.environ["VT_DBG_STAMPS"] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vittracker_amd import native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
geom = sys.argv[2] if len(sys.argv) > 2 else "G128"
tz, tx = {"G128": (64, 128), "G256": (128, 256)}[geom]
m = native.Model(tz, tx, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
z, x = synth.synth_inputs(1, B, tz, tx)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
for _ in range(3):
    m.stem(zd, xd)
torch.cuda.synchronize()
buf = np.zeros((B * 8 * 64,), dtype=np.uint64)
native._check(native.lib().vt_debug_stamps(m._h, B, buf.ctypes.data), "stamps")
st = buf[: B * 16 * 32].reshape(B, 16, 32).astype(np.int64)
if geom == "G256":   # stem_pipe_kernel: group A = [L1, barrier, fetch+L2, barrier] x 10; group B the same after one barrier
    d = np.diff(st, axis=2)
    print(f"stem_pipe B={B}: mean shader cycles between consecutive stamps, by wave (A = waves 0-7, B = waves 8-15)")
    print("stamp".ljust(8) + "".join(f"w{w:<2d}".rjust(7) for w in range(16)))
    for k in range(30):
        print(str(k).ljust(8) + "".join(f"{d[:, w, k].mean():7.0f}" for w in range(16)))
    sys.exit(0)
names = ["fetch+clear", "barrier"]
for t in range(6):
    names += [f"interval {t} work", f"interval {t} barrier"]
names += ["L3 (+pads, w4 req)", "barrier", "L4 + store"]
d = np.diff(st[:, :, : len(names) + 1], axis=2)
print(f"stem_fused B={B}: mean shader cycles per phase by wave (A = waves 0-7, B = waves 8-15)")
print("phase".ljust(22) + "".join(f"w{w:<2d}".rjust(7) for w in range(16)))
for k, n in enumerate(names):
    print(n.ljust(22) + "".join(f"{d[:, w, k].mean():7.0f}" for w in range(16)))
tot = st[:, :, len(names)] - st[:, :, 0]
print("total".ljust(22) + "".join(f"{tot[:, w].mean():7.0f}" for w in range(16)))
