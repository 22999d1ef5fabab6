#!/bin/bash
# ViT-Base: the step captured as N chains of frame slices (VT_GRAPH_CHAINS), each chain's persistent GEMMs on VT_CHAIN_CUS workgroups
# (0 = CUs / N), ms per step by tools/ab_vitb.py (in-tree library only), one box session
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbchains; rm -rf $O; mkdir -p $O
cd $R
for setting in "${@:-base}"; do
  ( for kv in $setting; do [ $kv != base ] && export $kv; done
    timeout 600 python tools/ab_vitb.py --rounds 1 --only cur 2>&1 | grep "vitb" | sed "s/^/$setting  /" ) | tee -a $O/out.txt
done
