#!/bin/bash
# round-2 checkpoint A: GPU tests, bench line, kernel-trace stats at G128 and G256 (B=256)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2a; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g128 -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu --no-extra > $O/prof_g128.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g256 -- python3 $R/bench.py --geom G256 --steps 50 --warmup 10 --no-cpu --no-extra > $O/prof_g256.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.db" -delete
