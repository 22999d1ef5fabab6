#!/bin/bash
# ViT-Base checkpoint: GPU parity tests, bench line, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-vitb}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vitb.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 150 python bench.py --config vitb --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config vitb --steps 10 --warmup 2 --no-extra > $O/prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
