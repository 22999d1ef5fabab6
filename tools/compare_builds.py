#!/usr/bin/env python3
"""ViT-Base: two builds of the library must produce bit-identical outputs (k-loop variants accumulate every output element in the
same order: k-tile by k-tile, k-step 0 before k-step 1).   python tools/compare_builds.py build_variants/ph8.so [B,B,...]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %(root)r)
import torch
from vittracker_amd import native, synth
if %(lib)r: native.LIB_PATH = %(lib)r
res = {}
sd = synth.synth_vitb_state_dict(26)
for B in %(sizes)r:
    m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
    m.load_state_dict(sd)
    z, x = synth.synth_inputs(B + 1, B, 128, 256)
    o = m.forward(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
    h = hashlib.sha256()
    for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
        h.update(getattr(o, k).cpu().numpy().tobytes())
    res[B] = h.hexdigest()[:16]
    m.close()
print("RESULT " + json.dumps(res))
"""
other = os.path.abspath(sys.argv[1])
sizes = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "5", "37", "96", "256"))]
out = {}
for name, lib in (("cur", ""), ("other", other)):
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib, "sizes": sizes}], capture_output=True, text=True, timeout=1200)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    if p.returncode or not line:
        print(name, "FAILED", p.stdout[-300:], p.stderr[-800:]); sys.exit(2)
    out[name] = json.loads(line[0][7:])
same = out["cur"] == out["other"]
print(out)
print("builds agree bit for bit:", same)
sys.exit(0 if same else 1)
