#!/bin/bash
# round 4: bench.py's default IS one stream now
# (kernel stats and counters are taken on ONE stream -- a kernel alone on the chip, which is what bench.py's roofline.avg_launch_us
# times; the default bench line runs two shards on two streams, whose kernels overlap)
# Round-4 evidence in one box session: rocprofv3 kernel stats (G128, G256, ViT-Base), PMC passes (G128, G256 -> summaries +
# profiles/pmc_traffic.json; ViT-Base FETCH_SIZE / WRITE_SIZE -> per-step HBM bytes), default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4prof; rm -rf $O; mkdir -p $O
COMMIT=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
for g in G128 G256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$g -- python3 $R/bench.py --geom $g --steps 100 --warmup 20 --no-cpu --no-extra --streams 1 > $O/stats_$g.log 2>&1
  cp $O/stats_$g/*/*kernel_stats.csv $O/r4_$(echo $g | tr A-Z a-z)_kernel_stats.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vitb -- python3 $R/tools/vitb_time.py > $O/stats_vitb.log 2>&1
cp $O/stats_vitb/*/*kernel_stats.csv $O/r4_vitb_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
bash tools/pmc.sh gpurun_out/r4prof/pmc_g128 > /dev/null 2>&1
bash tools/pmc.sh gpurun_out/r4prof/pmc_g256 --geom G256 > /dev/null 2>&1
python3 tools/pmc_traffic.py gpurun_out/r4prof/pmc_g128 G128_B256 $COMMIT > /dev/null
python3 tools/pmc_traffic.py gpurun_out/r4prof/pmc_g256 G256_B256 $COMMIT > /dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/r4prof/pmc_g128/summary.txt $O/r4_g128_pmc_summary.txt
cp gpurun_out/r4prof/pmc_g256/summary.txt $O/r4_g256_pmc_summary.txt
# (ViT-Base PMC passes: the vb_* sources did not change in round 4 -- profiles/r3_vitb_pmc_traffic.json stays valid)
find $O -name "*counter_collection.csv" -size +2000k -delete
cd $R
timeout 900 python bench.py > $O/r4_bench.json 2> $O/r4_bench.err; tail -c 600 $O/r4_bench.json
