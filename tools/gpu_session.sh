#!/bin/bash
# One parameterised GPU-box session script (round 5; replaces the one-shot gpu_r4_*.sh family).
#   gpurun --timeout N -- bash tools/gpu_session.sh <task> [task ...]
# Every task writes under gpurun_out/s6/<task>/ and prints a short summary.  Tasks:
#   vitb_tests      tests/test_gpu_vitb.py (+ the ViT-Base variants test)
#   vitb_ab         tools/ab_vitb.py: in-tree library against build_variants/*.so, interleaved, one box
#   vitb_nofold     the ViT-Base tests and one timing with VB_LN_FOLD=0 (separate LayerNorm kernel)
#   vitb_prof       rocprofv3 kernel stats of tools/vitb_time.py
#   vitb_traffic    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/vitb_time.py -> traffic json
#   tests           the whole -m gpu suite
#   quick48         parity subset + per-stage times of the vit_48 path (G128, G256)
#   ab48 [names]    tools/ab_stages.py against build_variants/*.so
#   bench           python bench.py (the driver's line) -> bench.json
#   bench_noextra   python bench.py --no-extra
#   race            tools/race_check.py at G128 and G256
#   trackstep       tracking/track_batch_demo.py --batch 256 at both geometries
#   generic         tests/test_gpu_generic.py
#   u8tests         round 6: the uint8-patch path (tests/test_gpu_patch_u8.py) + the tracker-level suites that step through vt_track_step
#   trackstep_ab    tracking/track_batch_demo.py --batch 256 with the uint8 patch (default) and VT_TRACK_U8=0 (fp32 crop), both geometries, one box
#   trackprof       rocprofv3 kernel stats of the tracker step (uint8 patch and VT_TRACK_U8=0), both geometries
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for task in "$@"; do
  O=$R/gpurun_out/s6/$task; rm -rf $O; mkdir -p $O
  echo "=== $task"
  case $task in
    vitb_tests)
      timeout 1200 python -m pytest tests/test_gpu_vitb.py tests/test_gpu_variants.py -k vitb -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
      tail -5 $O/pytest.txt ;;
    vitb_ab)
      timeout 1500 python tools/ab_vitb.py --rounds 2 > $O/ab.txt 2>&1; cat $O/ab.txt ;;
    vitb_nofold)
      VB_LN_FOLD=0 timeout 900 python -m pytest tests/test_gpu_vitb.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
      tail -3 $O/pytest.txt
      VB_LN_FOLD=0 timeout 600 python tools/ab_vitb.py --rounds 1 --only cur > $O/ab.txt 2>&1; cat $O/ab.txt ;;
    vitb_prof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/vitb_time.py > $O/prof.log 2>&1)
      python3 - $O/prof <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows:
        if float(r['TotalDurationNs'])/tot>0.003: print('  %-70s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
P
      find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete ;;
    vitb_traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/vitb_time.py > $O/$c.log 2>&1)
      done
      # per-kernel HBM bytes from the two passes, written under $O only (the tracked profiles/ files are installed by tools/install_profiles.py)
      python3 - $O <<'P'
import csv, glob, sys, collections, json
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                tot[k][c] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": n[k] += 1
res = {k: {"dispatches": n[k], "fetch_kib_per_dispatch": round(v["FETCH_SIZE"] / max(1, n[k]), 1), "write_kib_per_dispatch": round(v["WRITE_SIZE"] / max(1, n[k]), 1),
           "hbm_bytes_per_dispatch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / max(1, n[k]))}
       for k, v in tot.items() if "rocclr" not in k and "at::" not in k}
json.dump(res, open(f"{O}/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1)[:3000])
P
      find $O -name "*.db" -delete ;;
    tests)
      timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt ;;
    quick48)
      timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q -k "not vitb" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
      timeout 600 python tools/ab_stages.py --only cur > $O/stages.txt 2>&1; cat $O/stages.txt ;;
    ab48)
      timeout 1500 python tools/ab_stages.py > $O/ab.txt 2>&1; cat $O/ab.txt ;;
    bench)
      timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err; head -c 2500 $O/bench.json ;;
    bench_noextra)
      timeout 600 python bench.py --no-extra > $O/bench.json 2> $O/bench.err; head -c 2500 $O/bench.json ;;
    race)
      timeout 600 python tools/race_check.py > $O/g128.txt 2>&1; tail -3 $O/g128.txt
      timeout 600 python tools/race_check.py --geom G256 --B 256 > $O/g256.txt 2>&1; tail -3 $O/g256.txt ;;
    trackstep)
      timeout 600 python tracking/track_batch_demo.py --batch 256 --geom G128 > $O/g128.txt 2>&1; tail -4 $O/g128.txt
      timeout 600 python tracking/track_batch_demo.py --batch 256 --geom G256 > $O/g256.txt 2>&1; tail -4 $O/g256.txt ;;
    generic)
      timeout 900 python -m pytest tests/test_gpu_generic.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt ;;
    u8tests)
      timeout 1500 python -m pytest tests/test_gpu_patch_u8.py tests/test_gpu_pipeline.py tests/test_gpu_tracker.py tests/test_gpu_harness.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -15 $O/pytest.txt ;;
    trackstep_ab)
      for g in G128 G256; do for u in 1 0; do
        VT_TRACK_U8=$u timeout 600 python tracking/track_batch_demo.py --batch 256 --geom $g > $O/${g}_u8_$u.txt 2>&1; echo "--- $g VT_TRACK_U8=$u"; tail -3 $O/${g}_u8_$u.txt
      done; done ;;
    trackprof)
      for g in G128 G256; do for u in 1 0; do
        rm -rf $O/prof_${g}_$u
        (cd /tmp && VT_TRACK_U8=$u timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${g}_$u -- python3 $R/tracking/track_batch_demo.py --batch 256 --geom $g --frames 40 --one-stream > $O/prof_${g}_$u.log 2>&1)
        echo "--- $g VT_TRACK_U8=$u"
        python3 - $O/prof_${g}_$u <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:7]:
        print(f"  {r['Name'].split('(')[0].replace('void ','')[:70]:70s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:7.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
P
      done; done
      find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete ;;
    *) echo "unknown task $task" ;;
  esac
done
