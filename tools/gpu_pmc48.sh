#!/bin/bash
# vit_48 fp32: PMC passes at G128 and G256 (B=256) -> summaries + profiles/pmc_traffic.json entries
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/pmc.sh gpurun_out/pmc_g128 > /dev/null 2>&1
bash tools/pmc.sh gpurun_out/pmc_g256 --geom G256 > /dev/null 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_in.json
python3 tools/pmc_traffic.py gpurun_out/pmc_g128 G128_B256 ${1:-unknown} > /dev/null
python3 tools/pmc_traffic.py gpurun_out/pmc_g256 G256_B256 ${1:-unknown} > /dev/null
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
tail -5 gpurun_out/pmc_g128/summary.txt; tail -5 gpurun_out/pmc_g256/summary.txt
