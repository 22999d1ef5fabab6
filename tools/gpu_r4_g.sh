#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g; rm -rf $O; mkdir -p $O
cd $R
python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/gather_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $O/gather_probe.txt
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_harness.py tests/test_gpu_tracker.py tests/test_gpu_safety.py -m gpu -q -x --deselect tests/test_gpu_harness.py::test_force_gather_form_is_as_fast_as_the_plain_form 2>&1 | tail -5 | tee $O/pytest.txt
