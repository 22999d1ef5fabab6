#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_safety.py tests/test_gpu_harness.py tests/test_gpu_tracker.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tracking/track_batch_demo.py --batch 256 --frames 48 2>&1 | grep -v amdgpu.ids
timeout 600 python tracking/track_batch_demo.py --config vit_48_h32_g128 --batch 256 --frames 48 2>&1 | grep -v amdgpu.ids
