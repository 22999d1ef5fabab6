#!/bin/bash
# where do stem_fused's LDS bank conflicts come from?  The conflict counter of the stem kernel with layers 3 / 4 as BF3 (default) and
# on fp32 MFMAs (VT_STEM_BF3=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4t; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export VT_STEM_BF3=$v
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/bf3_$v -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --streams 1 > $O/log$v.txt 2>&1
  f=$(find $O/bf3_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "stem" in k or "head" in k or "blocks" in k:
        print("VT_STEM_BF3=" + sys.argv[2], k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
  rm -rf $O/bf3_$v
done
