#!/usr/bin/env python3
"""Race screen for one library build: the large-batch step replayed on FIXED inputs must be bit-identical every time, and a
permuted batch must permute the outputs.  Localises a mismatch to (stage, frame, token tile).

    python tools/race_check.py [--lib build_variants/x.so] [--geom G128] [--B 256] [--reps 40]
"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--geom", default="G128")
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
for k in ("VT_STEM_FUSED", "VT_STEM_PIPE", "VT_HEAD_FUSED"):
    os.environ.setdefault(k, "1")
os.environ.setdefault("VT_BLOCKS_TILE", "0")
os.environ.setdefault("VT_HEAD_SPLIT", "0")
import torch
from vittracker_amd import native, synth
if a.lib:
    native.LIB_PATH = os.path.join(ROOT, a.lib) if not os.path.isabs(a.lib) else a.lib
tz, tx = {"G128": (64, 128), "G256": (128, 256)}[a.geom]
lz, lx = (tz // 16) ** 2, (tx // 16) ** 2
m = native.Model(tz, tx, max_batch=a.B)
m.load_state_dict(synth.synth_state_dict(0, len_z=lz, len_x=lx))
z, x = synth.synth_inputs(9, a.B, tz, tx)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
tok = m.stem(zd, xd).clone()
bad = {"stem": 0, "blocks": 0, "head": 0, "forward": 0}
tiles = {}
feat0 = m.blocks(tok).clone()
out0 = m.forward(zd, xd)
ref = {k: getattr(out0, k).clone() for k in ("score_map", "size_map", "offset_map", "hann_boxes")}
for r in range(a.reps):
    t2 = m.stem(zd, xd)
    if not torch.equal(t2, tok):
        bad["stem"] += 1
    f = m.blocks(tok)
    if not torch.equal(f, feat0):
        bad["blocks"] += 1
        d = (f != feat0).view(a.B, lx // 16, 16 * 48).any(dim=2)        # (frame, search token tile)
        for fr, tl in d.nonzero().tolist():
            tiles[tl] = tiles.get(tl, 0) + 1
    o = m.forward(zd, xd)
    if not all(torch.equal(getattr(o, k), ref[k]) for k in ref):
        bad["forward"] += 1
torch.cuda.synchronize()
print(f"{a.lib or 'in-tree'} {a.geom} B={a.B} reps={a.reps}: mismatching replays {bad}; blocks mismatches by search-token tile {dict(sorted(tiles.items()))}")
sys.exit(1 if any(bad.values()) else 0)
