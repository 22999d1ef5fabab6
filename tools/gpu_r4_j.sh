#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4j; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/race_check.py 2>&1 | grep -v amdgpu.ids | tee $O/race.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 900 python tools/ab_stages.py --geom G128 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
timeout 300 python tools/stem_stamps.py 256 2>&1 | grep -v amdgpu.ids | head -22 | tee $O/stamps.txt
