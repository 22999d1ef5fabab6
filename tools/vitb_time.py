#!/usr/bin/env python3
"""ViT-Base: replay the captured step a few times (no checks) -- for rocprofv3 runs of timing experiments (VB_DBG)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vittracker_amd import native, synth
if os.environ.get("VT_LIB"):            # timing experiments: another build of the library (build_variants/*.so)
    native.LIB_PATH = os.environ["VT_LIB"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
m.load_state_dict(synth.synth_vitb_state_dict(26))
z, x = synth.synth_inputs(0, B, 128, 256)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
g, out = m.capture(zd, xd)
for _ in range(6):
    g.launch()
torch.cuda.synchronize()
