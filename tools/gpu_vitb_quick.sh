#!/bin/bash
# ViT-Base quick loop: parity tests, the bench line, per-kernel averages (rocprofv3 kernel stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbq; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vitb.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 600 python bench.py --config vitb --steps 10 --warmup 3 --no-extra > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').readline()); print('vitb', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['frac_bf16_peak_whole_step'], d['check']['max_abs_err'])" || tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/vitb_time.py > $O/prof.log 2>&1
python3 - $O/prof <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows:
        if float(r['TotalDurationNs'])/tot>0.01: print('  %-70s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
P
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
