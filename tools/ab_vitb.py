#!/usr/bin/env python3
"""ViT-Base A/B of library builds in ONE box session: ms per step (graph replay, B = 256) for the in-tree library and every
build_variants/*.so, interleaved rounds, with the fixture check of bench_vitb (boxes differ by several per cent).

    python tools/ab_vitb.py [--rounds 2] [--only name,name]
"""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys, json, types
sys.path.insert(0, %(root)r)
from vittracker_amd import native
if %(path)r:
    native.LIB_PATH = %(path)r
from vittracker_amd import bench_vitb
a = types.SimpleNamespace(gpus=1, batch=256, steps=%(steps)d, warmup=3, no_extra=True)
l = bench_vitb.measure(a)
print("RESULT " + json.dumps({"ms": l["ms_per_step"], "fps": l["value"], "err": l["check"]["max_abs_err"]["score_map"]}))
"""
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    vdir = os.path.join(ROOT, "build_variants")
    variants = {"cur": ""}
    if os.path.isdir(vdir):
        for f in sorted(os.listdir(vdir)):
            if f.endswith(".so"):
                variants[f[:-3]] = os.path.join(vdir, f)
    if a.only:
        variants = {k: v for k, v in variants.items() if k in a.only.split(",")}
    res = {k: [] for k in variants}
    for _ in range(a.rounds):
        for name, path in variants.items():
            p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "path": path, "steps": a.steps}], capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if p.returncode or not line:
                print(f"{name}: FAILED rc={p.returncode} {p.stdout[-200:]} {p.stderr[-500:]}")
                continue
            res[name].append(json.loads(line[0][7:]))
    for name, rows in res.items():
        if rows:
            print(f"vitb {name:>12s}: ms/step median {statistics.median(r['ms'] for r in rows):.3f}  min {min(r['ms'] for r in rows):.3f}   frames/s max {max(r['fps'] for r in rows):.0f}   err {rows[0]['err']:.2e}")
if __name__ == "__main__":
    main()
