#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python tools/ab_stages.py --geom G128 --rounds 2 --no-check 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
for v in build_variants/exp_GUESTS0.so build_variants/exp_OWNERS0.so; do
echo "== $v" | tee -a $O/stamps.txt
VITTRACK_LIB=$R/$v timeout 200 python tools/block_stamps.py G128 256 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
