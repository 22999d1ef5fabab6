#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_harness.py -m gpu -q -x 2>&1 | tail -3
for args in "--steps 20 --warmup 5" "--steps 7 --warmup 3" "--steps 200 --warmup 20" "--steps 20 --warmup 5 --streams 2" "--steps 20 --warmup 5 --gather-every 4"; do
  timeout 300 python bench.py --gpus 1 --force-gather $args --no-cpu --no-extra 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', d['value'], d['ms_per_step'], d.get('gather_exposed_us_per_step'), d['gather']['units_per_gather'], d['gather']['records_checked'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain 20', d['value'], d['ms_per_step'])"
