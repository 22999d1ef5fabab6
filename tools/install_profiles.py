#!/usr/bin/env python3
"""Copy the files tools/gpu_profiles.sh left under gpurun_out/r<N>prof/ into profiles/ (tracked) and stamp the ViT-Base traffic JSON
with the kernel-source hash bench.py checks.  Usage: python tools/install_profiles.py <round> <commit>"""
import json, os, shutil, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
N = sys.argv[1] if len(sys.argv) > 1 else "5"
commit = sys.argv[2] if len(sys.argv) > 2 else "?"
O = os.path.join(ROOT, "gpurun_out", f"r{N}prof")
P = os.path.join(ROOT, "profiles")
names = [f"r{N}_g128_kernel_stats.csv", f"r{N}_g256_kernel_stats.csv", f"r{N}_vitb_kernel_stats.csv", f"r{N}_trackstep_g128_kernel_stats.csv",
         f"r{N}_trackstep_g256_kernel_stats.csv", f"r{N}_g128_pmc_summary.txt", f"r{N}_g256_pmc_summary.txt", "pmc_traffic.json", f"r{N}_bench.json",
         f"r{N}_trackstep_g128_pmc_traffic.json", f"r{N}_trackstep_g256_pmc_traffic.json"]
for f in names:
    src = os.path.join(O, f)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f))
    else:
        print("missing", f)
tp = os.path.join(O, f"r{N}_vitb_pmc_traffic.json")
if os.path.exists(tp):
    t = json.load(open(tp))
    t.update({"_kernel_source_hash": bench.kernel_source_hash(prefixes=("vb_", "vitb")), "_forwards_in_run": 6, "_commit": commit,
              "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/vitb_time.py (6 graph replays, B = 256); bytes = 2 x FETCH_SIZE + WRITE_SIZE per dispatch"})
    json.dump(t, open(os.path.join(P, f"r{N}_vitb_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
bp = os.path.join(P, f"r{N}_bench.json")
if os.path.exists(bp):
    d = json.loads(open(bp).readline())
    print(d["value"], d["ms_per_step"], d["frac_fp32_peak_whole_step"], d.get("roofline", {}).get("frac"), d.get("stages_us"))
tj = json.load(open(os.path.join(P, "pmc_traffic.json")))
for k in ("G128_B256", "G256_B256"):
    print(k, tj[k]["_kernel_source_hash"], "current", bench.kernel_source_hash())
