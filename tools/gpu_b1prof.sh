#!/bin/bash
# per-kernel times of the plugin tracker's B = 1 device step (kernel trace of tools/plugin_profile.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/b1prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for y in vit_48_h32_g128 vit_48_h32_noKD; do
python3 $R/tools/plugin_profile.py $y > $O/$y.plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$y -- python3 $R/tools/plugin_profile.py $y > $O/$y.log 2>&1
python3 - $O/$y <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    tot = 0
    for r in csv.DictReader(open(f)):
        if int(r['Calls'])>=400:
            n = int(r['Calls']) / 410.0
            print('  ',r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2)); tot += float(r['AverageNs'])/1e3 * round(n)
    print('   sum per step', round(tot, 1))
P
cat $O/$y.plain.log | tail -2
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
