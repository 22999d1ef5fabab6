#!/bin/bash
# Counters of the crop kernels alone (tools/crop_ab.py's child: uint8 and fp32 forms at T = 128 / 256, 256 frames), one rocprofv3 --pmc pass per set,
# for the default form and VT_CROP_BAND=0 (crop_fast_kernel): is the address path (TA) what the kernel waits for?  (NOTES R6-2)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/crop_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp CROP_AB_CHILD=1 VT_CROP_BYTES=0
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
 "TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
)
for band in 4 0; do
  export VT_CROP_BAND=$band
  i=0
  for s in "${SETS[@]}"; do
    timeout 300 rocprofv3 --pmc $s --output-format csv -d $O/b${band}_set$i -- python3 $R/tools/crop_ab.py > $O/b${band}_set$i.log 2>&1
    i=$((i+1))
  done
done
python3 - $O <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
for band in (4, 0):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"{O}/b{band}_set*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "crop" not in k: continue
            key = (k, r.get("Grid_Size", ""))
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
    print(f"== VT_CROP_BAND={band}: per launch")
    for key in sorted(tot):
        if max(n[key].values()) < 50: continue
        print(f"  {key[0]}  grid {key[1]}")
        print("    " + "  ".join(f"{c} {tot[key][c] / n[key][c]:.3g}" for c in sorted(tot[key])))
P
find $O -name "*.db" -delete
