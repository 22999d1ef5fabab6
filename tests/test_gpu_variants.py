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

import numpy as np
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
    "stem_stream_forced": {"VT_STEM_STREAM": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},     # the streaming stem (G256 default at large batches), both geometries
    "stem_stream_off": {"VT_STEM_STREAM": "0", "VT_STEM_FUSED": "1", "VT_STEM_PIPE": "1"},
    "two_kernel_stem": {"VT_STEM_FUSED": "0", "VT_STEM_FUSE": "0"},
    "two_kernel_stem_joint_bands": {"VT_STEM_FUSED": "0", "VT_STEM_FUSE": "1"},
    "g256_stem_a_instead_of_stem_pipe": {"VT_STEM_PIPE": "0"},
    "per_tower_head": {"VT_HEAD_FUSED": "0", "VT_HEAD_SPLIT": "0"},
    # the head's towers on the bf16 matrix pipe with exact three-piece operands (vt_head3.h; the default: all four layers at G128,
    # conv1 of the one-workgroup-per-frame form at G256 = head_seq3), both kernel forms, and the fp32-MFMA towers they replaced
    # (VT_HEAD_BF3=0)
    "head_bf16x3_small_batch_form": {"VT_HEAD_BF3": "1"},
    "head_bf16x3_fused_form": {"VT_HEAD_BF3": "1", "VT_HEAD_FUSED": "1"},
    "head_fp32_mfma_small_batch_form": {"VT_HEAD_BF3": "0"},
    "head_fp32_mfma_fused_form": {"VT_HEAD_BF3": "0", "VT_HEAD_FUSED": "1"},
    # the frame-form block kernels' three levels (vt_blocks.h, vt_bf3.h): 2 (default) = every contraction of the G128 form / qkv, MLP, q k^T
    # and proj of the G256 form multiply exact three-piece bf16 splits (round 5: K, V^T published as pieces); 1 = qkv + MLP only
    # (round 4); 0 = fp32 MFMAs.  Level 2 is what "large_batch_forms_forced" runs.
    "blocks_qkv_mlp_only_bf16x3": {"VT_BLOCKS_BF3": "1", "VT_STEM_FUSED": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},
    "blocks_mlp_fp32_mfma": {"VT_BLOCKS_BF3": "0", "VT_STEM_FUSED": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},
    # layer 3 of the fused G128 stem on fp32 MFMAs (the default multiplies exact three-piece bf16 splits there too)
    "stem_fused_layer3_fp32_mfma": {"VT_STEM_BF3": "0", "VT_STEM_FUSED": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},
    "everything_on_fp32_mfma": {"VT_STEM_BF3": "0", "VT_BLOCKS_BF3": "0", "VT_HEAD_BF3": "0", "VT_STEM_FUSED": "1", "VT_HEAD_FUSED": "1", "VT_BLOCKS_TILE": "0"},
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


# A step captured as N parallel chains over frame slices (VT_GRAPH_CHAINS, vt_graph_capture_steps): the chains run concurrently,
# so every per-frame workspace -- the tile-form blocks' q / K / V^T / residual sets included -- must be sliced by the chain's
# first frame.  Graph replay == the eager step, bit for bit, at batches that select the tile form (small) and the frame form.
CHAINS_CODE = r"""
import sys, os
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import GEOMS
from vittracker_amd import native, synth
assert torch.cuda.is_available()
chains = int(os.environ["VT_GRAPH_CHAINS"])
for geom, batches in (("G128", (2, 7, 96, 540)), ("G256", (2, 7, 40))):
    tz, tx = GEOMS[geom]
    sd = synth.synth_state_dict(5, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    for B in batches:
        m = native.Model(tz, tx, max_batch=B); m.load_state_dict(sd)
        z, x = synth.synth_inputs(40 + B, B, tz, tx)
        zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
        ref = m.forward(zd, xd)
        ref = {k: getattr(ref, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")}
        graph, out = m.capture(zd, xd)
        for _ in range(3):
            graph.launch()
        torch.cuda.synchronize()
        # a chain steps B / chains frames: when that selects other kernel forms than the whole batch does (the forms follow the batch
        # size, DESIGN.md 4.6), the results agree to fp32 rounding instead of bit for bit
        same_forms = B <= 80 or B // chains > 176
        for k, v in ref.items():
            if same_forms:
                assert torch.equal(getattr(out, k), v), (geom, B, k)
            elif k in ("score_map", "size_map", "offset_map"):
                assert float((getattr(out, k) - v).abs().max()) < 2e-5, (geom, B, k)
        graph = None; m.close()
print("OK")
""" % {"root": ROOT}


@pytest.mark.parametrize("chains", ["2", "3"])
def test_graph_chains_match_eager(chains):
    env = dict(os.environ, VT_GRAPH_CHAINS=chains)
    r = subprocess.run([sys.executable, "-c", CHAINS_CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])


# ViT-Base (vitb.hip): the chains work on frame slices of every workspace (tower-major head buffers keep the batch's stride) and
# their persistent GEMMs launch on a share of the CUs; results are per-row, so the graph must equal the eager step bit for bit.
VITB_CHAINS_CODE = r"""
import sys
sys.path.insert(0, %(root)r)
import torch
from vittracker_amd import native, synth
for B in (5, 24):
    m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B + 3)
    m.load_state_dict(synth.synth_vitb_state_dict(26))
    z, x = synth.synth_inputs(11, B, 128, 256)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    e = m.forward(zd, xd)
    ref = {k: getattr(e, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")}
    g, out = m.capture(zd, xd)
    for _ in range(2):
        g.launch(); torch.cuda.synchronize()
        for k, v in ref.items():
            assert torch.equal(getattr(out, k), v), (B, k)
    m.close()
print("OK")
""" % {"root": ROOT}


@pytest.mark.parametrize("chains", ["2", "3"])
def test_vitb_graph_chains_match_eager(chains):
    env = dict(os.environ, VT_GRAPH_CHAINS=chains)
    r = subprocess.run([sys.executable, "-c", VITB_CHAINS_CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])


def test_vitb_default_two_full_grid_chains_match_eager():
    """Round 5 default: a captured ViT-Base step of >= 64 frames runs as two chains of half the frames, each chain's persistent GEMMs on
    a workgroup per CU (they fill each other's partly empty last tile rounds).  B = 72 (two chains of 36) and B = 63 (one chain) must
    both replay the eager step bit for bit."""
    code = VITB_CHAINS_CODE.replace("for B in (5, 24):", "for B in (72, 63):")
    env = {k: v for k, v in os.environ.items() if k not in ("VT_GRAPH_CHAINS", "VT_CHAIN_CUS", "VT_CHAIN_DELAY_US")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])


# The bf16-split forms (vt_bf3.h: qkv + MLP of the frame-form block kernels, the F = 8 head's towers) multiply the same fp32 operands
# as the fp32-MFMA forms they replace; on a full batch of random crops -- every frame-form kernel, 256 frames instead of a fixture's
# 2-5 -- the two must agree to the fp32 noise floor of the net (a few 1e-6 on the maps), far inside the 1e-4 / 1e-5 the golden
# tests hold either of them to.  This is the direct statement that the split is a faster way to issue fp32 products, not a
# precision change.
SPLIT_CODE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from vittracker_amd import native, synth
tz, tx, B = %(tz)d, %(tx)d, %(B)d
m = native.Model(tz, tx, max_batch=B)
m.load_state_dict(synth.synth_state_dict(3, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
z, x = synth.synth_inputs(17, B, tz, tx)
out = m.forward(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
torch.cuda.synchronize()
np.savez(%(path)r, score=out.score_map.cpu().numpy(), size=out.size_map.cpu().numpy(), offset=out.offset_map.cpu().numpy(),
         pred=out.pred_boxes.cpu().numpy())
print("OK")
"""


@pytest.mark.parametrize("geom", ["G128", "G256"])
def test_bf16_split_forms_agree_with_fp32_mfma_forms_on_a_full_batch(geom, tmp_path):
    tz, tx = {"G128": (64, 128), "G256": (128, 256)}[geom]
    res = {}
    for name, env in (("split", {}), ("fp32", {"VT_BLOCKS_BF3": "0", "VT_HEAD_BF3": "0"})):
        path = str(tmp_path / f"{name}.npz")
        code = SPLIT_CODE % {"root": ROOT, "tz": tz, "tx": tx, "B": 256, "path": path}
        e = {k: v for k, v in os.environ.items() if k not in ("VT_BLOCKS_BF3", "VT_HEAD_BF3")}
        r = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])
        res[name] = np.load(path)
    a, b = res["split"], res["fp32"]
    assert not np.array_equal(a["score"], b["score"]), "the switch selected the same kernels twice"
    # maps: sigmoid / clamp outputs in [1e-4, 1] and raw offsets of O(1); boxes: a grid cell index + offset, / 8 or / 16
    for k, tol in (("score", 1e-5), ("size", 1e-5), ("offset", 2e-5)):
        assert np.abs(a[k] - b[k]).max() <= tol, (k, float(np.abs(a[k] - b[k]).max()))
    # a box moves by more than the noise only where the two runs pick different argmax cells of near-tied scores
    same = np.abs(a["pred"] - b["pred"]).max(axis=1) <= 1e-5
    assert same.mean() >= 0.99, float(same.mean())



@pytest.mark.parametrize("geom", ["G128", "G256"])
def test_form_batch_makes_a_small_batch_run_the_large_batch_forms(geom):
    """vt_set_form_batch(256): a batch of 8 then runs the one-workgroup-per-frame forms a batch of 256 runs, so its outputs equal the
    first 8 frames of the 256-batch bit for bit (without it: the multi-workgroup forms, equal to fp32 rounding only)."""
    import torch
    from vittracker_amd import native, synth
    tz, tx = {"G128": (64, 128), "G256": (128, 256)}[geom]
    sd = synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    z, x = synth.synth_inputs(4, 256, tz, tx)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    big = native.Model(tz, tx, max_batch=256)
    big.load_state_dict(sd)
    ob = big.forward(zd, xd)
    small = native.Model(tz, tx, max_batch=8)
    small.load_state_dict(sd)
    plain = small.forward(zd[:8].contiguous(), xd[:8].contiguous())
    keys = ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")
    plain = {k: getattr(plain, k).clone() for k in keys}
    small.set_form_batch(256)
    formed = small.forward(zd[:8].contiguous(), xd[:8].contiguous())
    for k in keys:
        assert torch.equal(getattr(formed, k), getattr(ob, k)[:8]), (geom, k)
    assert any(not torch.equal(plain[k], getattr(ob, k)[:8]) for k in keys)
    assert float((plain["score_map"] - ob.score_map[:8]).abs().max()) < 1e-4
    with pytest.raises(native.VtError):
        small.set_form_batch(-1)
    # ORDER rules of include/vittrack.h: the template cache and captured graphs belong to the form batch they were made under
    z8, x8 = zd[:8].contiguous(), xd[:8].contiguous()
    small.set_template(z8)
    cached = small.forward(None, x8)
    for k in keys:
        assert torch.equal(getattr(cached, k), getattr(formed, k)), (geom, k)
    small.set_form_batch(0)
    with pytest.raises(native.VtError, match="form batch"):
        small.forward(None, x8)                      # the cache holds the 256-forms' operands
    small.set_template(z8)
    small.forward(None, x8)
    graph, _ = small.capture(z8, x8)
    small.set_form_batch(0)                          # unchanged value: accepted
    with pytest.raises(native.VtError, match="vt_graph_capture"):
        small.set_form_batch(256)                    # the captured graph keeps the forms of its capture
    small._graphs.discard(graph)
    del graph                                        # ... and only while it lives: the last graph destroyed, the value is free again
    small.set_form_batch(256)
    # a graph that outlives its model is still destroyed safely (vt_destroy orphans it)
    g2, _ = small.capture(z8, x8)
    small.close()
    with pytest.raises(native.VtError):
        g2.launch()
    del g2


# ViT-Base kernel-form switches (vitb.hip): the default step folds LayerNorm into qkv / fc1 and runs the qkv projection inside the
# attention kernel; VB_LN_FOLD=0 keeps the separate LayerNorm kernel, VB_FUSED_QKV=0 the qk GEMM + v GEMM + attention kernels,
# VB_QA_HGROUP the fused kernel's item order.  Every form is held to the reference's golden vectors and stage activations at the
# default form's tolerances (the child runs tests/test_gpu_vitb.py's golden, stage and batch-invariance tests under the switch).
@pytest.mark.parametrize("env_kv", ["VB_FUSED_QKV=0", "VB_LN_FOLD=0", "VB_LN_FOLD=0 VB_FUSED_QKV=0", "VB_QA_HGROUP=12", "VB_QA_HGROUP=4"])
def test_vitb_kernel_form_switches_hold_the_golden_vectors(env_kv):
    env = dict(os.environ, **dict(kv.split("=") for kv in env_kv.split()))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_vitb.py"), "-m", "gpu", "-x", "-q", "-k",
                        "golden or each_stage or batch_invariance"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
