#!/usr/bin/env python3
"""ViT-Base: the captured step as 1..4 concurrent chains over frame slices (VT_GRAPH_CHAINS, VT_CHAIN_CUS, VT_CHAIN_DELAY_US; a case is chains:cus[:delay_us]): ms per step in ONE box
session, and chained graph == eager forward bit for bit.     python tools/vitb_chains.py [--B 256] [--cases 1:0,2:0,2:160,3:0,4:0]"""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys, json, types, hashlib
sys.path.insert(0, %(root)r)
import torch
from vittracker_amd import native, synth, bench_vitb
a = types.SimpleNamespace(gpus=1, batch=%(B)d, steps=%(steps)d, warmup=3, no_extra=True)
l = bench_vitb.measure(a)
B = %(B)d
m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
m.load_state_dict(synth.synth_vitb_state_dict(26))
z, x = synth.synth_inputs(5, B, 128, 256)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
e = m.forward(zd, xd)
ref = {k: getattr(e, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")}
g, out = m.capture(zd, xd)
g.launch(); torch.cuda.synchronize()
same = all(torch.equal(getattr(out, k), v) for k, v in ref.items())
print("RESULT " + json.dumps({"ms": l["ms_per_step"], "fps": l["value"], "same_as_eager": same}))
"""
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--cases", default="1:0,2:0,2:160,3:0,4:0")
    a = ap.parse_args()
    cases = [tuple(int(v) for v in c.split(":")) for c in a.cases.split(",")]
    res = {c: [] for c in cases}
    for _ in range(a.rounds):
        for c in cases:
            env = dict(os.environ, VT_GRAPH_CHAINS=str(c[0]), VT_CHAIN_CUS=str(c[1]), VT_CHAIN_DELAY_US=str(c[2] if len(c) > 2 else 0))
            p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "B": a.B, "steps": a.steps}], capture_output=True, text=True, timeout=900, env=env)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if p.returncode or not line:
                print(f"{c}: FAILED rc={p.returncode} {p.stdout[-300:]} {p.stderr[-800:]}"); continue
            res[c].append(json.loads(line[0][7:]))
    for c, rows in res.items():
        if rows:
            print(f"vitb chains {c[0]} cus/chain {c[1] or 'auto':>4} delay {c[2] if len(c) > 2 else 0:>4} us: ms/step median {statistics.median(r['ms'] for r in rows):.3f} min {min(r['ms'] for r in rows):.3f}  "
                  f"frames/s max {max(r['fps'] for r in rows):.0f}  graph == eager: {all(r['same_as_eager'] for r in rows)}")
if __name__ == "__main__":
    main()
