#!/usr/bin/env python3
"""Per-block, per-frame error of the block kernel against the numpy oracle with scaled-up attention logits."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from oracle import vt_oracle_np as onp
from vittracker_amd import native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
fac = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
sd = synth.synth_state_dict(21, len_z=16, len_x=64)
for blk in range(3):
    sd[f"blocks.{blk}.attn.qkv.weight"][:96] *= fac
    sd[f"blocks.{blk}.attn.qkv.bias"][:96] *= fac
z, x = synth.synth_inputs(21, B, 64, 128)
ref = onp.forward(sd, z, x, want_acts=True)["acts"]
m = native.Model(64, 128, max_batch=B)
m.load_state_dict(sd)
tok = m.stem(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
print("tokens err", float(np.abs(tok.cpu().numpy() - ref["tokens"]).max()), " |ref| max", float(np.abs(ref["tokens"]).max()))
for nb in (1, 2, 3):
    feat, resid = m.blocks(tok, nblocks=nb, want_resid=True)
    d = np.abs(resid.cpu().numpy() - ref[f"block{nb - 1}"])
    print(f"block {nb - 1}: max err {d.max():.3e}  |ref| max {np.abs(ref[f'block{nb - 1}']).max():.2f}  per frame:",
          " ".join(f"{d[b].max():.1e}" for b in range(B)), " worst token tile:", int(np.unravel_index(d.argmax(), d.shape)[1]) // 16)
