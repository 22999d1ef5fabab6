#!/bin/bash
# quick vit_48 check: parity tests of the fp32 path + kernel-trace stats at G128 and G256 (B=256)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/q48; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_f16cache.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
for g in G128 G256; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$g -- python3 $R/bench.py --geom $g --steps 100 --warmup 20 --no-cpu --no-extra > $O/prof_$g.log 2>&1
python3 - $O/prof_$g <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if int(r['Calls'])>50: print('  ',r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
grep -o '"value": [0-9.]*' $O/prof_$g.log | head -1
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
