#!/bin/bash
# race screen of library builds (in-tree + build_variants/*.so), then per-stage A/B and block stamps of the in-tree build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b; rm -rf $O; mkdir -p $O
cd $R
for v in "" $(ls build_variants/*.so 2>/dev/null); do
  timeout 300 python tools/race_check.py --lib "$v" 2>&1 | grep -v amdgpu.ids | tee -a $O/race.txt
done
timeout 300 python tools/race_check.py --geom G256 --B 200 --reps 10 2>&1 | grep -v amdgpu.ids | tee -a $O/race.txt
timeout 900 python tools/ab_stages.py --geom G128,G256 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
timeout 200 python tools/block_stamps.py G128 256 2>&1 | grep -v amdgpu.ids | tee $O/stamps_g128.txt
