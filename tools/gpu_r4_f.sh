#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f; rm -rf $O; mkdir -p $O
cd $R
for args in "" "--gpus 1 --force-gather" "--gpus 1 --force-gather --steps-per-graph 8" "--gpus 1 --force-gather --steps-per-graph 16" "--steps-per-graph 16" "--gpus 1 --force-gather --streams 2" "--streams 2"; do
  timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu --no-extra $args 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$args'.ljust(52), d['value'], d['ms_per_step'], d.get('gather_exposed_us_per_step'))
" | tee -a $O/gather.txt
done
timeout 600 python tools/ab_stages.py --geom G256 --rounds 2 2>&1 | grep -v amdgpu.ids | tee $O/ab256.txt
timeout 300 python tools/race_check.py --geom G256 --B 200 --reps 10 2>&1 | grep -v amdgpu.ids | tee -a $O/ab256.txt
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
