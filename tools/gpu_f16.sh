#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-f16}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16cache.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 900 python bench.py --config vit48_f16cache --warmup 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
