#!/bin/bash
# ViT-Base GEMMs on a persistent grid of 256 / 128 / 64 workgroups, whole and as the DMA skeleton (VB_DBG=12: no MFMA, no epilogue):
# does a k-tile's time depend on how many CUs stream at once (shared L2 / fabric) or not (the CU's own issue path)?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbcus; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 0 12 4; do for c in 256 128 64; do
  VB_DBG=$d VB_MAX_CUS=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d${d}_c$c -- python3 $R/tools/vitb_time.py ${VBCUS_B:-256} > $O/d${d}_c$c.log 2>&1
  echo "== VB_DBG=$d VB_MAX_CUS=$c"; grep "gemm_kernel<256, 256" $O/d${d}_c$c/*/*kernel_stats.csv | awk -F, '{print $1, $4}' | sed 's/.*gemm_kernel//'
done; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
