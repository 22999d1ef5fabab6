#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h; rm -rf $O; mkdir -p $O
cd $R
for args in "" "--gpus 1 --force-gather" "--gpus 1 --force-gather --gather-every 1" "--gpus 1 --force-gather --streams 2" "--streams 2" "--record-graphs"; do
  timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu --no-extra $args 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$args'.ljust(52), d['value'], d['ms_per_step'], d.get('gather_exposed_us_per_step'), (d.get('gather') or {}).get('records_checked'))
" | tee -a $O/gather.txt
done
timeout 2400 python -m pytest tests/test_gpu_harness.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
