#!/bin/bash
# ViT-Base kernel timing experiments: rocprofv3 kernel stats of tools/vitb_time.py (one chain) for the in-tree build and for each argument --
# a variant build (build_variants/<name>.so) or an environment setting (NAME=VALUE).  Prints the kernels whose name matches $KPAT
# (default: every vb kernel above 1 % of the step).     usage: [KPAT=regex] tools/gpu_vbexp.sh [variant | NAME=VALUE]...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vbexp; rm -rf $O; mkdir -p $O
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
  python3 - $O/$v "$v" "${KPAT:-}" <<'P'
import csv, glob, re, sys
d, name, pat = sys.argv[1:4]
for f in glob.glob(d + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        k = r["Name"].split("(")[0].replace("void ", "")
        share = float(r["TotalDurationNs"]) / tot
        if (pat and re.search(pat, k)) or (not pat and share > 0.01):
            print(f"{name:18s} {k[:60]:60s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  share {100*share:5.1f} %")
    print(f"{name:18s} all kernels: {tot/1e6/7:.3f} ms per step (7 replays incl. capture run)")
P
  case $v in *=*) unset "${v%%=*}" ;; esac
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
