#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tracker.py tests/test_gpu_harness.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.txt
for cfg in vit_48_h32_g128; do
for lib in "" $R/build_variants/prev.so "" $R/build_variants/prev.so; do
echo "== $cfg lib=${lib:-cur}" | tee -a $O/demo.txt
VITTRACK_LIB=$lib timeout 600 python tracking/track_batch_demo.py --config $cfg --batch 256 --frames 200 2>&1 | grep -v amdgpu.ids | grep "frames already\|track_chunk, 4\|two shards" | tee -a $O/demo.txt
done
done
