#!/bin/bash
# vit_48: the multi-step-graph test, then the default bench line and the config-5 line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/b48; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_step or golden" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err; tail -2 $O/bench.err
timeout 600 python bench.py --config vit48_f16cache --no-cpu > $O/bench_f16.json 2> $O/bench_f16.err; echo "bench rc=$?" >> $O/bench_f16.err; tail -2 $O/bench_f16.err
python3 - <<'P'
import json
for f in ("bench.json","bench_f16.json"):
    d=json.loads(open("gpurun_out/b48/"+f).readline())
    print(f, d["value"], d["ms_per_step"], d.get("stages_us"), json.dumps(d.get("also"))[:900])
P
