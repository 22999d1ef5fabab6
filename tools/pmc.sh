#!/bin/bash
# Collect hardware counters for one bench.py configuration, one rocprofv3 --pmc pass per counter
# set (counters only: no trace domains, as the GPU pool requires).  Usage:
#   tools/pmc.sh <outdir> [bench.py args...]
#   PMC_VITB=1 tools/pmc.sh <outdir> [frames]      the same passes over tools/vitb_time.py (the captured ViT-Base step)
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for s in "${SETS[@]}"; do
  if [ -n "${PMC_VITB:-}" ]; then
    timeout 300 rocprofv3 --pmc $s --output-format csv -d "$R/$OUT/set$i" -- python3 "$R/tools/vitb_time.py" "$@" > "$R/$OUT/set$i.log" 2>&1
  else
    timeout 300 rocprofv3 --pmc $s --output-format csv -d "$R/$OUT/set$i" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu --no-extra --streams 1 "$@" > "$R/$OUT/set$i.log" 2>&1
  fi
  i=$((i+1))
done
python3 "$R/tools/pmc_summary.py" "$R/$OUT" | tee "$R/$OUT/summary.txt"
