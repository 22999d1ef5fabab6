#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O
cd $R
for v in sk4 sk16 sk20; do
  echo "== $v" | tee -a $O/stamps.txt
  VITTRACK_LIB=$R/build_variants/$v.so timeout 300 python tools/head_stamps.py 2>&1 | grep -v amdgpu.ids | grep "ctr conv1\|size conv1\|total" | tee -a $O/stamps.txt
done
