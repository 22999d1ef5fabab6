#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/race_check.py --geom G256 --B 256 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/race256.txt
timeout 300 python tools/race_check.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/race128.txt
timeout 900 python tools/stress_two_streams.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/stress.txt
