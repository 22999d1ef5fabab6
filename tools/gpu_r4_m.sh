#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O
cd $R
for v in exp1 exp2 exp3; do
  echo "== $v" | tee -a $O/stamps.txt
  VITTRACK_LIB=$R/build_variants/$v.so timeout 300 python tools/head_stamps.py 2>&1 | grep -v amdgpu.ids | grep "conv1\|total" | tee -a $O/stamps.txt
done
timeout 900 python tools/ab_stages.py --geom G256 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
