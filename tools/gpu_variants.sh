#!/bin/bash
# compiler-flag variants of libvittrack_hip.so (build_variants/*.so, built on the dev box): per-kernel times at G128 / G256
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/variants; rm -rf $O; mkdir -p $O
cp $R/vittracker_amd/csrc/libvittrack_hip.so /tmp/base.so
cd /tmp && export TMPDIR=/tmp
for v in base $(ls $R/build_variants | sed 's/\.so$//'); do
  if [ $v = base ]; then cp /tmp/base.so $R/vittracker_amd/csrc/libvittrack_hip.so; else cp $R/build_variants/$v.so $R/vittracker_amd/csrc/libvittrack_hip.so; fi
  echo "== $v"
  for g in G128 G256; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_$g -- python3 $R/bench.py --geom $g --steps 100 --warmup 20 --no-cpu --no-extra > $O/${v}_$g.log 2>&1
    python3 - $O/${v}_$g <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    print('   '+'  '.join(f"{r['Name'].split('(')[0].replace('void ','')[5:32]} {float(r['AverageNs'])/1e3:.2f}" for r in csv.DictReader(open(f)) if int(r['Calls'])>50))
P
  done
done
cp /tmp/base.so $R/vittracker_amd/csrc/libvittrack_hip.so
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
