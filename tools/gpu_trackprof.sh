#!/bin/bash
# per-kernel times of the whole batched tracker step (tracking/track_batch_demo.py) at batch 256
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trackprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for y in vit_48_h32_g128 vit_48_h32_noKD; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$y -- python3 $R/tracking/track_batch_demo.py --batch 256 --config $y > $O/$y.log 2>&1
python3 - $O/$y <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>1.0: print('  %-80s calls %5s avg %8.1f us %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
P
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
