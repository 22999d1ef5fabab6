#!/bin/bash
# Gaps between consecutive kernels of the captured step (rocprofv3 --kernel-trace of bench.py, one stream): start[i + 1] - end[i] per kernel pair
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in ${@:-G128}; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$g -- python3 $R/bench.py --geom $g --steps 100 --warmup 20 --no-cpu --no-extra --streams 1 > $O/$g.log 2>&1
  python3 - $O/$g $g <<'P'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]            # steady state
    gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        ka, kb = (r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0] for r in (a, b))
        gaps[ka + " -> " + kb].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
        durs[ka].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    for k, v in gaps.items():
        v.sort(); print(f"{sys.argv[2]} gap {k:60s} n {len(v):4d}  median {v[len(v)//2]:6.2f} us  min {v[0]:6.2f}")
    for k, v in durs.items():
        v.sort(); print(f"{sys.argv[2]} dur {k:60s} n {len(v):4d}  median {v[len(v)//2]:6.2f} us")
P
  tail -1 $O/$g.log | cut -c1-200
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
