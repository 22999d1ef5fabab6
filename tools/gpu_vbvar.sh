#!/bin/bash
# ViT-Base GEMM kernel times (rocprofv3 kernel stats, VB_DBG from the environment, default 8 = no epilogue) for the in-tree library
# and every build_variants/*.so -- timing experiments on k-loop variants in one box session.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbvar; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VB_DBG=${VB_DBG:-8}
for lib in cur $(ls $R/build_variants/*.so 2>/dev/null); do
  n=$(basename $lib .so)
  if [ $n = cur ]; then unset VT_LIB; else export VT_LIB=$lib; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/tools/vitb_time.py > $O/$n.log 2>&1
  python3 - $O/$n $n <<'P'
import csv,sys,glob,re
names={0:"patch",1:"qk",2:"proj/fc2",3:"fc1",4:"conv1",5:"v"}
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    row={}
    for r in csv.DictReader(open(f)):
        m=re.search(r"gemm_kernel<256, 256, 2, 4, (\d), (\d)>",r["Name"])
        if m: row[names[int(m.group(2))]]=round(float(r["AverageNs"])/1e3,1)
    print("%-14s"%sys.argv[2], {k:row.get(k) for k in ("qk","v","proj/fc2","fc1","conv1")})
P
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
