#!/bin/bash
# The round's evidence in ONE box session (generalises rounds 3-4's one-off scripts):
#   gpurun --timeout 2400 -- bash tools/gpu_profiles.sh <round> <commit>
# -> gpurun_out/r<round>prof/: rocprofv3 kernel stats (G128, G256, ViT-Base; one stream: a kernel alone on the chip, which is what
# bench.py's roofline.avg_launch_us times), PMC passes (G128, G256 -> summaries + pmc_traffic.json; ViT-Base FETCH_SIZE / WRITE_SIZE ->
# per-dispatch HBM bytes), the tracker-step kernel stats, the default bench line.  `python tools/install_profiles.py <round> <commit>`
# then copies the summaries into profiles/ (tracked).
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-5}
COMMIT=${2:-unknown}
O=$R/gpurun_out/r${N}prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in G128 G256; do
  gl=$(echo $g | tr A-Z a-z)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$g -- python3 $R/bench.py --geom $g --steps 100 --warmup 20 --no-cpu --no-extra --streams 1 > $O/stats_$g.log 2>&1
  cp $O/stats_$g/*/*kernel_stats.csv $O/r${N}_${gl}_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/track_$g -- python3 $R/tracking/track_batch_demo.py --batch 256 --geom $g --frames 120 --one-stream --device-frames-only --hold-boxes > $O/track_$g.log 2>&1
  cp $O/track_$g/*/*kernel_stats.csv $O/r${N}_trackstep_${gl}_kernel_stats.csv
  # HBM bytes per launch of the tracker step's kernels (FETCH_SIZE / WRITE_SIZE, one pass each; held boxes: the 120-360 px windows of a tracker that follows a target)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/trackpmc_${g}_$c -- python3 $R/tracking/track_batch_demo.py --batch 256 --geom $g --frames 24 --one-stream --device-frames-only --hold-boxes > $O/trackpmc_${g}_$c.log 2>&1
  done
  python3 - $O $g $N <<'P'
import csv, glob, sys, collections, json
O, g, N = sys.argv[1:4]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{O}/trackpmc_{g}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                tot[k][c] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": n[k] += 1
res = {k: {"dispatches": n[k], "fetch_mb_per_dispatch": round(2 * v["FETCH_SIZE"] * 1024 / max(1, n[k]) / 1e6, 2), "write_mb_per_dispatch": round(v["WRITE_SIZE"] * 1024 / max(1, n[k]) / 1e6, 2)}
       for k, v in tot.items() if "rocclr" not in k and "at::" not in k}
res["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over tracking/track_batch_demo.py --batch 256 --one-stream --hold-boxes; fetch = 2 x FETCH_SIZE KiB (gfx950: 128-byte requests counted as 64), write = WRITE_SIZE KiB, MB per launch"
json.dump(res, open(f"{O}/r{N}_trackstep_{g.lower()}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
print(g, json.dumps(res, indent=1)[:1500])
P
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vitb -- python3 $R/tools/vitb_time.py > $O/stats_vitb.log 2>&1
cp $O/stats_vitb/*/*kernel_stats.csv $O/r${N}_vitb_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
bash tools/pmc.sh gpurun_out/r${N}prof/pmc_g128 > /dev/null 2>&1
bash tools/pmc.sh gpurun_out/r${N}prof/pmc_g256 --geom G256 > /dev/null 2>&1
python3 tools/pmc_traffic.py gpurun_out/r${N}prof/pmc_g128 G128_B256 $COMMIT > /dev/null
python3 tools/pmc_traffic.py gpurun_out/r${N}prof/pmc_g256 G256_B256 $COMMIT > /dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/r${N}prof/pmc_g128/summary.txt $O/r${N}_g128_pmc_summary.txt
cp gpurun_out/r${N}prof/pmc_g256/summary.txt $O/r${N}_g256_pmc_summary.txt
# ViT-Base: HBM bytes per dispatch (two passes: FETCH_SIZE, WRITE_SIZE; 6 replays + 1 capture run in vitb_time.py)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/vitb_$c -- python3 $R/tools/vitb_time.py > $O/vitb_$c.log 2>&1
done
python3 - $O $N <<'P'
import csv,glob,sys,collections,json
O,N=sys.argv[1],sys.argv[2]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"{O}/vitb_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c:
                k=r["Kernel_Name"].split("(")[0].replace("void ","")
                tot[k][c]+=float(r["Counter_Value"])
                if c=="FETCH_SIZE": n[k]+=1
res={}
for k,v in tot.items():
    if "rocclr" in k or "at::" in k: continue
    res[k]={"dispatches":n[k],"fetch_kib_per_dispatch":round(v["FETCH_SIZE"]/max(1,n[k]),1),"write_kib_per_dispatch":round(v["WRITE_SIZE"]/max(1,n[k]),1),
            "hbm_bytes_per_dispatch":int((2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024/max(1,n[k]))}
json.dump(res,open(f"{O}/r{N}_vitb_pmc_traffic.json","w"),indent=1,sort_keys=True)
print(json.dumps(res,indent=1)[:3000])
P
find $O -name "*counter_collection.csv" -size +2000k -delete
cd $R
timeout 1200 python bench.py > $O/r${N}_bench.json 2> $O/r${N}_bench.err; tail -c 600 $O/r${N}_bench.json
