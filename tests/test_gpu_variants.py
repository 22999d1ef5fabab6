"""GPU parity of the non-default kernel variants.

The library picks, per geometry, the fastest of several kernels for each stage (fused per-frame stem /
two-kernel stem, fused head / per-tower head + decode kernel, balanced 8-wave block kernel / one wave
per tile with or without LDS-staged weights).  The variants stay selectable through VT_* environment
switches read at vt_create, and G256 uses some of them by default, so each one is held to the same
golden vectors.  One subprocess per combination: the switches are cached when the model is created.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CODE = r"""
import sys, glob, os
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import GEOMS, golden_files, load_case
from vittracker_amd import native
assert torch.cuda.is_available()
worst = 0.0
for path in golden_files():
    g, sd, z, x = load_case(path)
    tz, tx = GEOMS[str(g["geom"])]
    m = native.Model(tz, tx, max_batch=int(g["B"])); m.load_state_dict(sd)
    out = m.forward(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
    for k in ("score_map", "size_map", "offset_map"):
        worst = max(worst, float(np.abs(getattr(out, k).cpu().numpy() - g[k]).max()))
    np.testing.assert_allclose(out.pred_boxes.cpu().numpy(), g["pred_boxes"][:, 0], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out.hann_boxes.cpu().numpy(), g["hann_boxes"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out.conf.cpu().numpy(), g["conf"], atol=1e-4, rtol=0)
assert worst < 1e-4, worst    # same bound as tests/test_gpu_parity.py (TOL_MAP)
print("OK", worst)
""" % {"root": ROOT}

VARIANTS = {
    # the fixtures' batches are small: by default they run the multi-workgroup forms, so the large-batch forms are forced here
    "large_batch_forms_forced": {"VT_STEM_FUSED": "1", "VT_STEM_PIPE": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},
    "tile_parallel_blocks_forced": {"VT_BLOCKS_TILE": "1"},
    "two_kernel_stem": {"VT_STEM_FUSED": "0", "VT_STEM_FUSE": "0"},
    "two_kernel_stem_joint_bands": {"VT_STEM_FUSED": "0", "VT_STEM_FUSE": "1"},
    "g256_stem_a_instead_of_stem_pipe": {"VT_STEM_PIPE": "0"},
    "per_tower_head": {"VT_HEAD_FUSED": "0", "VT_HEAD_SPLIT": "0"},
    "g256_head_conv1_split_forced": {"VT_HEAD_FUSED": "0", "VT_HEAD_SPLIT": "1"},
    "blocks_wave_per_tile_lds_weights": {"VT_BLOCKS_BAL": "0", "VT_BLOCKS_WLDS": "1", "VT_BLOCKS_TILE": "0"},
    "blocks_wave_per_tile_l2_weights": {"VT_BLOCKS_BAL": "0", "VT_BLOCKS_WLDS": "0", "VT_BLOCKS_TILE": "0"},
    "everything_off": {"VT_STEM_FUSED": "0", "VT_STEM_FUSE": "0", "VT_STEM_PIPE": "0", "VT_HEAD_FUSED": "0", "VT_BLOCKS_BAL": "0", "VT_BLOCKS_TILE": "0"},
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_matches_reference_golden(name):
    env = dict(os.environ, **VARIANTS[name])
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])
