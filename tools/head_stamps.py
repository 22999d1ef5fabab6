#!/usr/bin/env python3
"""Phase breakdown of head_seq3_kernel (G256 head) from in-kernel s_memtime stamps (VT_DBG_STAMPS=1): mean shader cycles per
phase and wave over the frames of one launch.

    python tools/head_stamps.py [B]
"""
import os
import sys
os.environ["VT_DBG_STAMPS"] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vittracker_amd import native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = native.Model(128, 256, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=64, len_x=256))
z, x = synth.synth_inputs(1, B, 128, 256)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
for _ in range(3):
    m.forward(zd, xd)
torch.cuda.synchronize()
buf = np.zeros((B * 8 * 64,), dtype=np.uint64)
native._check(native.lib().vt_debug_stamps(m._h, B, buf.ctypes.data), "stamps")
st = buf.reshape(B, 8, 64).astype(np.int64)
names = ["clear", "stage tokens"]
for t in ("ctr", "offset", "size"):
    names += [f"{t} conv1", "  barrier", f"{t} conv2", "  barrier", f"{t} conv3", "  barrier", f"{t} conv4", "  barrier", f"{t} 1x1"]
names[-1:] = [f"size 1x1", "  barrier"]
d = np.diff(st[:, :, : len(names) + 1], axis=2)
print(f"head_seq3 B={B}: mean shader cycles per phase by wave")
print("phase".ljust(16) + "".join(f"w{w}".rjust(7) for w in range(8)))
for k, nme in enumerate(names):
    print(nme.ljust(16) + "".join(f"{d[:, w, k].mean():7.0f}" for w in range(8)))
tot = st[:, :, len(names)] - st[:, :, 0]
print("total".ljust(16) + "".join(f"{tot[:, w].mean():7.0f}" for w in range(8)))
