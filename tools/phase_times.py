#!/usr/bin/env python3
"""Phase-cost probe: times the stem / head stages with parts of the kernels switched off through
the VT_SKIP_* diagnostic masks (results are wrong by design; only the durations matter)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CODE = r"""
import sys, torch, numpy as np
sys.path.insert(0, %r)
from vittracker_amd import native, synth
geom = sys.argv[1]; B = int(sys.argv[2])
tz, tx = {"G128": (64, 128), "G256": (128, 256)}[geom]
m = native.Model(tz, tx, max_batch=B); m.load_state_dict(synth.synth_state_dict(0, len_z=(tz//16)**2, len_x=(tx//16)**2))
z, x = synth.synth_inputs(1, B, tz, tx); zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
s = torch.cuda.Stream(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, n=50):
    with torch.cuda.stream(s):
        for _ in range(5): fn()
        e0.record()
        for _ in range(n): fn()
        e1.record()
    e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tok = m.stem(zd, xd); feat = m.blocks(tok); out = native.Outputs(B, m.feat_sz, "cuda")
L = native.lib()
print("stem %%.1f  head %%.1f" %% (t(lambda: L.vt_stem(m._h, native._ptr(zd), native._ptr(xd), B, native._stream(s), native._ptr(tok))),
                                  t(lambda: m.head(feat, out, stream=s))))
""" % ROOT

def run(env, geom="G128", B=256):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", CODE, geom, str(B)], env=e, capture_output=True, text=True)
    return (r.stdout.strip() or r.stderr.strip()[-300:])

if __name__ == "__main__":
    geom = sys.argv[1] if len(sys.argv) > 1 else "G128"
    cases = [("baseline", {}),
             ("stem_a: no L1", {"VT_SKIP_STEM_A": "1"}), ("stem_a: no L2", {"VT_SKIP_STEM_A": "2"}), ("stem_a: nothing", {"VT_SKIP_STEM_A": "3"}),
             ("stem_b: no zero", {"VT_SKIP_STEM_B": "1"}), ("stem_b: no load", {"VT_SKIP_STEM_B": "2"}), ("stem_b: no L3", {"VT_SKIP_STEM_B": "4"}),
             ("stem_b: no L4", {"VT_SKIP_STEM_B": "8"}), ("stem_b: nothing", {"VT_SKIP_STEM_B": "15"}),
             ("stem: a+b nothing", {"VT_SKIP_STEM_A": "3", "VT_SKIP_STEM_B": "15"}),
             ("head: no zero", {"VT_SKIP_HEAD": "1"}), ("head: no load", {"VT_SKIP_HEAD": "2"}), ("head: no conv1", {"VT_SKIP_HEAD": "4"}),
             ("head: no conv2", {"VT_SKIP_HEAD": "8"}), ("head: no conv3,4", {"VT_SKIP_HEAD": "16"}), ("head: nothing", {"VT_SKIP_HEAD": "31"})]
    for name, env in cases:
        print(f"{name:22s} {run(env, geom)}", flush=True)
