#!/bin/bash
# ViT-Base fused qkv + attention kernel: its time in variant builds (build_variants/<name>.so) or under env settings (NAME=VALUE arguments)
# against the in-tree default, rocprofv3 kernel stats, one chain
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/qaexp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VT_GRAPH_CHAINS=1
for v in cur "$@"; do
  unset VT_LIB
  case $v in
    cur) ;;
    *=*) export "$v" ;;
    *) export VT_LIB=$R/build_variants/$v.so ;;
  esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $R/tools/vitb_time.py > $O/$v.log 2>&1
  grep -h "qkv_attn" $O/$v/*/*kernel_stats.csv | awk -F'",' '{print $2}' | awk -F, -v n=$v '{printf "%-18s qkv_attn avg %.1f us (min %.1f max %.1f)\n", n, $3/1000, $5/1000, $6/1000}'
  case $v in *=*) unset "${v%%=*}" ;; esac
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
