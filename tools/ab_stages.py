#!/usr/bin/env python3
"""A/B of library builds in ONE box session (boxes differ by ~3 %; MI355X_MICROARCH.md 'DVFS give-back' 5): per-stage and
whole-step times (HIP events, bench.py's Runner) for the in-tree library and every build_variants/*.so, interleaved rounds.

    python tools/ab_stages.py [--geom G128,G256] [--B 256] [--rounds 3] [--only name,name]
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CHILD = r"""
import sys, json
sys.path.insert(0, %(root)r)
from vittracker_amd import native
if %(path)r:
    native.LIB_PATH = %(path)r
import bench
r = bench.Runner(%(geom)r, %(B)d, steps_per_graph=4)
try:
    chk = r.check_against_golden()
except SystemExit as e:          # timing experiments with wrong-by-design builds (--no-check)
    if not %(nocheck)d: raise
    chk = {"max_abs_err": {k: float("nan") for k in ("score_map", "size_map", "offset_map")}}
r.prewarm(0.3)
st = r.stage_times(%(iters)d)
t = r.time_us(lambda: r.graph_s.launch(r.stream), %(iters)d) / r.S
st["step"] = t
st["err"] = max(chk["max_abs_err"][k] for k in ("score_map", "size_map", "offset_map"))
print("RESULT " + json.dumps(st))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="G128,G256")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-check", action="store_true", help="time builds whose results are wrong by design (phase-skip experiments)")
    a = ap.parse_args()
    vdir = os.path.join(ROOT, "build_variants")
    variants = {"cur": ""}
    if os.path.isdir(vdir):
        for f in sorted(os.listdir(vdir)):
            if f.endswith(".so") and not f.endswith("_f16.so"):
                variants[f[:-3]] = os.path.join(vdir, f)
    if a.only:
        variants = {k: v for k, v in variants.items() if k in a.only.split(",")}
    for geom in a.geom.split(","):
        res = {k: [] for k in variants}
        for _ in range(a.rounds):
            for name, path in variants.items():
                code = CHILD % {"root": ROOT, "path": path, "geom": geom, "B": a.B, "iters": a.iters, "nocheck": int(a.no_check)}
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
                if p.returncode or not line:
                    print(f"{geom} {name}: FAILED rc={p.returncode} {p.stdout[-300:]} {p.stderr[-600:]}")
                    continue
                res[name].append(json.loads(line[0][7:]))
        for name, rows in res.items():
            if not rows:
                continue
            med = {k: statistics.median(r[k] for r in rows) for k in rows[0]}
            mn = {k: min(r[k] for r in rows) for k in rows[0]}
            print(f"{geom} B={a.B} {name:>14s}: " + "  ".join(f"{k} {med[k]:.2f} (min {mn[k]:.2f})" for k in ("stem", "blocks", "head", "step")) + f"  err {med['err']:.1e}")


if __name__ == "__main__":
    main()
