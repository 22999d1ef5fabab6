#!/usr/bin/env python3
"""Per-stage times of the f16 build (libvittrack_hip_f16.so) at G128 / G256, B = 256: stem, blocks, head, whole step (eager)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from vittracker_amd import native, synth
B = 256
for geom, (tz, tx) in (("G128", (64, 128)), ("G256", (128, 256))):
    m = native.Model(tz, tx, max_batch=B, precision="f16")
    m.load_state_dict(synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
    z, x = synth.synth_inputs(0, B, tz, tx)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    tok = m.stem(zd, xd); feat = m.blocks(tok); out = native.Outputs(B, m.feat_sz, "cuda")
    def t(fn, n=200):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    g, _ = m.capture(zd, xd, out)
    print(f"f16 {geom} B={B}: stem {t(lambda: m.stem(zd, xd)):.2f}  blocks {t(lambda: m.blocks(tok, feat=feat)):.2f}  head {t(lambda: m.head(feat, out)):.2f}  graph step {t(lambda: g.launch()):.2f} us", flush=True)
    g = None; m.close()
