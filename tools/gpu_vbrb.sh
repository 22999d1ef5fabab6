#!/bin/bash
# Tile-order experiment for the ViT-Base GEMMs: VB_RB_<epilogue>=rows (see vitb.hip launch_gemm).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbrb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, env assignments...
  local name=$1; shift
  env "$@" true
  ( export "$@"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/tools/vitb_time.py > $O/$name.log 2>&1 )
  echo "== $name ($*)"; grep gemm_kernel $O/$name/*/*kernel_stats.csv | awk -F, '{print $1, $4}' | sed 's/.*gemm_kernel//'
}
run base VB_NONE=1
run rb4 VB_RB_3=4 VB_RB_1=4 VB_RB_4=4
run rb8 VB_RB_3=8 VB_RB_1=8 VB_RB_4=16
run rb16 VB_RB_3=16 VB_RB_1=16 VB_RB_4=32
run rb32 VB_RB_3=32 VB_RB_1=32 VB_RB_2=32
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
