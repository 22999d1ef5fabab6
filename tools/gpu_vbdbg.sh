#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbdbg; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in ${VB_DBG_LIST:-0 1 3 4 8 12}; do
  VB_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$d -- python3 $R/tools/vitb_time.py > $O/d$d.log 2>&1
  echo "== VB_DBG=$d"; grep gemm_kernel $O/d$d/*/*kernel_stats.csv | awk -F, '{print $1, $4}' | sed 's/.*gemm_kernel//' 
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
