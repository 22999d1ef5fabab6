#!/usr/bin/env python3
"""Phase breakdown of the transformer-block kernel from in-kernel s_memtime stamps
(VT_DBG_STAMPS=1): average shader cycles per phase, per wave index."""
import os
import sys
os.environ["VT_DBG_STAMPS"] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vittracker_amd import native, synth

geom = sys.argv[1] if len(sys.argv) > 1 else "G128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tz, tx = {"G128": (64, 128), "G256": (128, 256)}[geom]
m = native.Model(tz, tx, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
z, x = synth.synth_inputs(1, B, tz, tx)
tok = m.stem(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
for _ in range(3):
    m.blocks(tok)
torch.cuda.synchronize()
buf = np.zeros((B, 8, 64), dtype=np.uint64)
native._check(native.lib().vt_debug_stamps(m._h, B, buf.ctypes.data), "stamps")
fine = int(os.environ.get("VT_DBG_SKIP_TILE", "-1")) <= -2
if fine and geom == "G128":
    names = ["load+sync"] + sum([[f"b{k} ln1", f"b{k} qk (72)", f"b{k} v (36)", f"b{k} barrier1", f"b{k} scores (60)", f"b{k} softmax",
                                  f"b{k} pv (60)", f"b{k} proj (36)", f"b{k} barrier2", f"b{k} ln2", f"b{k} fc1 g0 (48)",
                                  f"b{k} fc1 g1+gelu0 (48)", f"b{k} fc1 g2+gelu1 (48)", f"b{k} barrier3", f"b{k} fc2 g0+gelu2 (48)",
                                  f"b{k} fc2 g1 (48)", f"b{k} fc2 g2 (48)"] for k in range(3)], []) + ["barrier4+tail"]
elif geom == "G128":   # WLDS variant: a barrier splits the MLP
    # no stamp behind a block's last barrier: a block's first phase includes the previous block's closing barrier (block 0: the load)
    names = sum([[f"b{k} (load|barrier4)+ln1+qkv", f"b{k} barrier1", f"b{k} attn+proj", f"b{k} barrier2", f"b{k} ln2+fc1+gelu01",
                   f"b{k} barrier3", f"b{k} fc2+gelu2"] for k in range(3)], []) + ["barrier4+tail"]
else:
    names = sum([[f"b{k} (load)ln1+qkv", f"b{k} barrier1", f"b{k} attn+proj", f"b{k} barrier2", f"b{k} mlp"] for k in range(3)], []) + ["tail"]
if os.environ.get("STAMP_RAW"):
    names = [f"s{k}" for k in range(int(os.environ["STAMP_RAW"]))]
bal = os.environ.get("VT_BLOCKS_BAL", "1") != "0"
nw = (8 if bal else 5) if geom == "G128" else 4
buf = buf.reshape(-1)[: B * nw * 64].reshape(B, nw, 64)
st = buf[:, :nw, :len(names) + 1].astype(np.int64)
d = np.diff(st, axis=2)
print(f"{geom} B={B}: mean shader cycles per phase, by wave index (total per wave in last row)")
print("phase".ljust(16) + "".join(f"wave{w:>2d}".rjust(10) for w in range(nw)))
for k, n in enumerate(names[:d.shape[2]]):
    print(n.ljust(26) + "".join(f"{d[:, w, k].mean():10.0f}" for w in range(nw)))
tot = (st[:, :, d.shape[2]] - st[:, :, 0])
print("total".ljust(16) + "".join(f"{tot[:, w].mean():10.0f}" for w in range(nw)))
