#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.txt
timeout 300 python tools/head_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
timeout 900 python tools/ab_stages.py --geom G256 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
