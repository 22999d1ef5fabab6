#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_pipeline.py tests/test_gpu_f16cache.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 300 python tools/race_check.py --geom G256 --B 256 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python bench.py --geom G256 --no-cpu --no-extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('G256', d['value'], d['ms_per_step'])"
