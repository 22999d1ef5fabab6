#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/race_check.py 2>&1 | grep -v amdgpu.ids | tee -a $O/race.txt
timeout 900 python tools/ab_stages.py --geom G128 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
echo "guests at default priority:" | tee -a $O/ab.txt
VT_DBG_SKIP_TILE=-3 timeout 900 python tools/ab_stages.py --geom G128 --rounds 2 --only cur 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 200 python tools/block_stamps.py G128 256 2>&1 | grep -v amdgpu.ids | tee $O/stamps_g128.txt
VT_DBG_SKIP_TILE=-2 timeout 200 python tools/block_stamps.py G128 256 2>&1 | grep -v amdgpu.ids | tee $O/fstamps_g128.txt
