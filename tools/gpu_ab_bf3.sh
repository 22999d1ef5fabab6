for i in 1 2; do
python bench.py --no-cpu --no-configs > gpurun_out/ab_def_$i.json 2> gpurun_out/ab_def_$i.err
VT_HEAD_BF3=1 python bench.py --no-cpu --no-configs > gpurun_out/ab_bf3_$i.json 2> gpurun_out/ab_bf3_$i.err
done
python - <<'PY'
import json
for n in ("def_1","bf3_1","def_2","bf3_2"):
    try:
        d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["stages_us"], d["roofline"]["frac"], d["roofline"].get("probe_clock_mhz"), d["also"].get("G256_frames_per_s"))
    except Exception as e: print(n, "ERR", e)
PY
