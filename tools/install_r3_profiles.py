#!/usr/bin/env python3
"""Copy the files tools/gpu_round3_profiles.sh left under gpurun_out/r3prof/ into profiles/ (tracked) and stamp the ViT-Base traffic
JSON with the kernel-source hash bench.py checks.  Usage: python tools/install_r3_profiles.py <commit>"""
import csv, json, os, shutil, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
O = os.path.join(ROOT, "gpurun_out", "r3prof")
P = os.path.join(ROOT, "profiles")
for f in ("r3_g128_kernel_stats.csv", "r3_g256_kernel_stats.csv", "r3_vitb_kernel_stats.csv", "r3_g128_pmc_summary.txt",
          "r3_g256_pmc_summary.txt", "pmc_traffic.json", "r3_bench.json"):
    shutil.copy(os.path.join(O, f), os.path.join(P, f))
t = json.load(open(os.path.join(O, "r3_vitb_pmc_traffic.json")))
t.update({"_kernel_source_hash": bench.kernel_source_hash(prefixes=("vb_", "vitb")), "_forwards_in_run": 6, "_commit": sys.argv[1] if len(sys.argv) > 1 else "?",
          "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/vitb_time.py (6 graph replays, B = 256); bytes = 2 x FETCH_SIZE + WRITE_SIZE per dispatch"})
json.dump(t, open(os.path.join(P, "r3_vitb_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
d = json.loads(open(os.path.join(P, "r3_bench.json")).readline())
print(d["value"], d["ms_per_step"], d["frac_fp32_peak_whole_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["stages_us"])
print({k: v for k, v in d["also"].items() if not isinstance(v, (dict, str))})
tj = json.load(open(os.path.join(P, "pmc_traffic.json")))
for k in ("G128_B256", "G256_B256"):
    print(k, tj[k]["_kernel_source_hash"], "current", bench.kernel_source_hash())
