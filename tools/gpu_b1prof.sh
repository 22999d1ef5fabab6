#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/b1prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in G128 G256; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$g -- python3 $R/bench.py --geom $g --batch 1 --steps 200 --warmup 20 --no-cpu --no-extra > $O/$g.log 2>&1
python3 - $O/$g <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if int(r['Calls'])>100: print('  ',r['Name'][:66], r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
grep -o '"ms_per_step": [0-9.]*' $O/$g.log | head -1
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
