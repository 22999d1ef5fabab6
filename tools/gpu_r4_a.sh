#!/bin/bash
# round 4, step A: parity subset on the current build, A/B of per-stage times against build_variants/*.so, block-kernel stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_f16cache.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
timeout 900 python tools/ab_stages.py --geom G128,G256 --rounds 3 > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 200 python tools/block_stamps.py G128 256 > $O/stamps_g128.txt 2>&1; cat $O/stamps_g128.txt
VT_DBG_SKIP_TILE=-2 timeout 200 python tools/block_stamps.py G128 256 > $O/fstamps_g128.txt 2>&1; cat $O/fstamps_g128.txt
