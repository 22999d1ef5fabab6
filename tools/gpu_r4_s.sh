#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tracker.py tests/test_gpu_harness.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
for cfg in vit_48_h32_g128 vit_48_h32_noKD; do timeout 600 python tracking/track_batch_demo.py --config $cfg --batch 256 --frames 400 2>&1 | grep -v amdgpu.ids | grep "frames already" | tee -a $O/demo.txt; done
