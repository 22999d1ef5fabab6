#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/f16_stages.py 2>&1 | grep -v amdgpu.ids | tee $O/f16_stages.txt
timeout 1500 python -m pytest tests/test_gpu_f16cache.py tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python bench.py --config vit48_f16cache --no-cpu 2>/dev/null | tail -1 | cut -c1-1500 | tee $O/bench_f16.json
