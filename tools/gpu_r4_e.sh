#!/bin/bash
# all GPU tests + default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err; tail -2 $O/bench.err
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/r4e/bench.json").readline())
print(d["value"], d["ms_per_step"], d.get("stages_us"), json.dumps(d.get("roofline"))[:1200])
print(json.dumps(d.get("also"))[:2500])
P
