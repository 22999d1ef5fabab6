#!/bin/bash
# ViT-Base GEMM kernel times with workgroup phase groups (VB_DESYNC_<epi>=<us>[:groups]), one rocprofv3 run per setting
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbdesync; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "${@:-base}"; do
  i=$((i+1))
  ( for kv in $setting; do [ $kv != base ] && export $kv; done
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -- python3 $R/tools/vitb_time.py > $O/s$i.log 2>&1 )
  python3 - $O/s$i "$setting" <<'P'
import csv,sys,glob,re
names={0:"patch",1:"qk",2:"proj/fc2",3:"fc1",4:"conv1",5:"v"}
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    row={}; tot=0
    for r in csv.DictReader(open(f)):
        tot+=float(r["TotalDurationNs"])
        m=re.search(r"gemm_kernel<256, 256, 2, 4, (\d), (\d)>",r["Name"])
        if m: row[names[int(m.group(2))]]=(round(float(r["AverageNs"])/1e3,1), round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
    print("%-40s"%sys.argv[2], {k:row.get(k) for k in ("qk","v","proj/fc2","fc1")}, "all kernels %.2f ms/step" % (tot/6e6))
P
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
