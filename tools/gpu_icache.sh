#!/bin/bash
# Instruction-cache counters per kernel of the bench step (one --pmc pass; no trace domains): requests, hits, misses and the mean
# instruction-fetch latency (SQ_IFETCH_LEVEL accumulated / SQ_IFETCH).   usage: tools/gpu_icache.sh [G128|G256]...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in ${@:-G128 G256}; do
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/$g -- python3 $R/bench.py --geom $g --steps 20 --warmup 5 --no-cpu --no-extra --streams 1 > $O/$g.log 2>&1
  echo "== $g"; python3 $R/tools/pmc_summary.py $O/$g | grep -E "^==|ICACHE|IFETCH|WAVE_CYCLES|BUSY_CYCLES" | grep -v -E "rocclr|at::"
done
