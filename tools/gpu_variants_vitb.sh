#!/bin/bash
# A/B of library builds in build_variants/*.so for the ViT-Base path: frames/s of bench.py --config vitb, one box session
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/vittracker_amd/csrc/libvittrack_hip.so /tmp/base.so
cd $R
for v in base $(ls $R/build_variants | sed 's/\.so$//') base; do
  if [ $v = base ]; then cp /tmp/base.so $R/vittracker_amd/csrc/libvittrack_hip.so; else cp $R/build_variants/$v.so $R/vittracker_amd/csrc/libvittrack_hip.so; fi
  timeout 200 python bench.py --config vitb --steps 20 --warmup 3 --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'])"
done
cp /tmp/base.so $R/vittracker_amd/csrc/libvittrack_hip.so
