#!/bin/bash
# kernel stats of the whole tracker step (device crop -> network on the cached template -> state update; tracking/track_batch_demo.py):
# per-frame launches, 4 frames per launch and two shards on two streams in one run -> profiles/r4_trackstep_<config>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trackprof; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in vit_48_h32_g128 vit_48_h32_noKD; do
  timeout 600 python tracking/track_batch_demo.py --config $cfg --batch 256 --frames 200 2>&1 | grep -v amdgpu.ids | tee $O/demo_$cfg.txt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 tracking/track_batch_demo.py --config $cfg --batch 256 --frames 200 > $O/prof_$cfg.log 2>&1
  find $O/prof_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r4_trackstep_${cfg}_kernel_stats.csv
  rm -rf $O/prof_$cfg
done
