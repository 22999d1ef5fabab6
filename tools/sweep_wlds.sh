#!/bin/bash
# block-kernel weight path (LDS-staged vs per-wave L2 reads) across batch sizes
for w in 1 0; do for B in 256 512 1024; do
  VT_BLOCKS_WLDS=$w timeout 200 python bench.py --batch $B --no-cpu --steps 100 2>/dev/null > /tmp/sw.json
  python - "$w" <<'PY'
import sys, json
d = json.load(open("/tmp/sw.json"))
print("wlds", sys.argv[1], "B", d["config"]["batch_per_gpu"], round(d["value"]), d["ms_per_step"], d["stages_us"])
PY
done; done
