#!/bin/bash
# rocprofv3 kernel stats of the real tracker step (tracking/track_batch_demo.py, 256 sequences): top kernels.
# Arguments: geometries (G128 G256; default both) and variant builds (build_variants/<name>.so, run beside the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trackprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GEOMS=""; VARS="cur"
for a in "$@"; do case $a in G128|G256) GEOMS="$GEOMS $a" ;; *) VARS="$VARS $a" ;; esac; done
for g in ${GEOMS:-G128 G256}; do
 for v in $VARS; do
  unset VITTRACK_LIB; [ $v != cur ] && export VITTRACK_LIB=$R/build_variants/$v.so
  rm -rf $O/$g; echo "--- $g $v"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$g -- python3 $R/tracking/track_batch_demo.py --batch 256 --geom $g --frames 40 > $O/$g.log 2>&1
  python3 - $O/$g $g <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:6]:
        print(f"{sys.argv[2]}  {r['Name'].split('(')[0].replace('void ','')[:60]:60s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:7.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
P
 done
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
