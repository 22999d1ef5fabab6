"""Pin the CPU oracles against the reference's own outputs (tests/golden/*.npz).

The fixtures were produced by running the reference model code itself
(tests/golden/make_golden.py); these tests are what makes the oracle trustworthy before it is
used to judge the HIP kernels.  Tolerances: the oracle and the reference both compute in fp32
but in different summation orders, so maps agree to ~1e-5 absolute; bboxes must be identical up
to that noise because every fixture's argmax margin is >= 1e-3.
"""
import numpy as np
import pytest

from conftest import golden_files, load_case
from oracle import vt_oracle_np as onp

TOL = 2e-5


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_numpy_oracle_matches_reference(path):
    g, sd, z, x = load_case(path)
    out = onp.forward(sd, z, x, want_acts=True)
    for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
        np.testing.assert_allclose(out[k], g[k], atol=TOL, rtol=0, err_msg=k)
    assert onp.top2_margin(g["score_map"]).min() > 5e-4  # fixtures must stay far from argmax flips
    np.testing.assert_allclose(onp.hann2d(g["score_map"].shape[-1]), g["hann_window"], atol=1e-7)
    # per-stage activations where the fixture carries them
    a = out["acts"]
    if "act_norm" in g:
        for i in range(4):
            np.testing.assert_allclose(a["stem_z"][i], g[f"act_stem{i}_z"], atol=TOL, err_msg=f"stem{i}_z")
            np.testing.assert_allclose(a["stem_x"][i], g[f"act_stem{i}_x"], atol=TOL, err_msg=f"stem{i}_x")
        for i in range(3):
            np.testing.assert_allclose(a[f"block{i}"], g[f"act_block{i}"], atol=TOL, err_msg=f"block{i}")
        np.testing.assert_allclose(a["norm"], g["act_norm"], atol=TOL)
        for t in ("ctr", "offset", "size"):
            for i in range(4):
                np.testing.assert_allclose(a[f"head_{t}"][i], g[f"act_head_{t}{i + 1}"], atol=TOL,
                                           err_msg=f"head_{t}{i + 1}")


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_fp64_truth_brackets_reference(path):
    """The float64 run of the oracle is the yardstick: the reference's fp32 output sits within
    ~1e-5 of it, which is the noise floor any fp32 implementation (ours included) shares."""
    g, sd, z, x = load_case(path)
    out = onp.forward(sd, z, x, dtype=np.float64)
    for k in ("score_map", "size_map", "offset_map"):
        assert np.abs(out[k] - g[k]).max() < TOL, k


@pytest.mark.parametrize("path", golden_files()[:3], ids=lambda p: p.split("/")[-1][:-4])
def test_torch_oracle_matches_reference(path):
    import torch
    from oracle import vt_oracle_torch as ot
    g, sd, z, x = load_case(path)
    m = ot.build_from_state(sd)
    with torch.no_grad():
        out = m(torch.from_numpy(z), torch.from_numpy(x))
    for k in ("score_map", "size_map", "offset_map", "pred_boxes"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=TOL, rtol=0, err_msg=k)


def test_numpy_oracle_matches_reference_on_uint8_patches():
    """The reference model on Preprocessor-normalised uint8 patches (tests/golden/make_golden_u8.py): the oracle on the same
    patches, normalised with the same arithmetic, within the usual fp32 noise -- pins the oracle for the uint8-patch path."""
    from conftest import load_u8_case, u8_golden_files
    from vittracker_amd import synth
    files = u8_golden_files()
    assert len(files) == 2, "uint8-patch fixtures missing"
    for path in files:
        g, sd, z, patches = load_u8_case(path)
        for recip in (False, True):      # the CPU's true division (how the fixture was made) and the GPU's reciprocal multiply
            out = onp.forward(sd, z, synth.normalise_patches(patches, reciprocal=recip))
            for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
                np.testing.assert_allclose(out[k], g[k], atol=TOL, rtol=0, err_msg=f"{k} reciprocal={recip}")
        assert onp.top2_margin(g["score_map"]).min() > 4e-4


def test_numpy_oracle_matches_reference_at_other_widths():
    """Other points of the config surface build_ostrack_dist accepts (CHANNELS / HEADS / HEAD.NUM_CHANNELS = 64 / 2 / 64 and 32 / 4 / 16:
    tests/golden/make_golden_cfg.py): the oracle is shape-generic, the fixtures pin it there too."""
    from conftest import cfg_golden_files, load_cfg_case
    files = cfg_golden_files()
    assert len(files) == 2, "config-surface fixtures missing"
    for path in files:
        g, sd, z, x, (C, heads, W) = load_cfg_case(path)
        out = onp.forward(sd, z, x, num_heads=heads)
        for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
            np.testing.assert_allclose(out[k], g[k], atol=TOL, rtol=0, err_msg=f"{k} C={C} heads={heads} W={W}")


def test_clip_box_and_hann_known_answers():
    import os
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "ref_clip_box.npz"))
    for b, c in zip(g["boxes"].tolist(), g["clipped"].tolist()):
        assert onp.clip_box(b, int(g["H"]), int(g["W"]), int(g["margin"])) == pytest.approx(c, abs=0)
    h = np.load(os.path.join(GOLDEN_DIR, "ref_hann.npz"))
    for n in (8, 16, 20):
        np.testing.assert_allclose(onp.hann2d(n), h[f"hann{n}"], atol=1e-7)


def test_mac_counts_match_survey():
    """hook-counted MACs of the reference model."""
    g256 = onp.macs_per_frame(128, 256)
    assert (g256["stem"], g256["blocks"], g256["head"], g256["total"]) == (13271040, 56033280, 15266816, 84571136)
    g128 = onp.macs_per_frame(64, 128)
    assert (g128["stem"], g128["blocks"], g128["head"], g128["total"]) == (3317760, 8478720, 3816704, 15613184)


# ---------------------------------------------------------------------------------------- ViT-Base (config 4)
def test_vitb_torch_oracle_matches_reference():
    """oracle/vitb_oracle_torch.py against the outputs and activations of the reference's own build_ostrack model
    (tests/golden/make_golden_vitb.py).  fp32 both sides; different op order (conv vs unfold etc.) -> ~1e-5."""
    import torch
    from conftest import load_vitb_case, vitb_golden_files
    from oracle import vitb_oracle_torch as ob
    files = vitb_golden_files()
    assert files, "ViT-Base fixtures missing"
    for path in files:
        g, sd, z, x = load_vitb_case(path)
        m = ob.build_from_state(sd)
        acts = {}
        with torch.no_grad():
            out = m(torch.from_numpy(z), torch.from_numpy(x), acts)
        for k in ("score_map", "size_map", "offset_map", "pred_boxes"):
            np.testing.assert_allclose(out[k].numpy(), g[k], atol=5e-5, rtol=0, err_msg=k)
        assert min(g["margin_raw"].min(), g["margin_hann"].min()) > 0.03
        if "act_norm" in g:
            rows = g["act_rows"]
            for k in ["tokens", "norm"] + [f"block{i}" for i in range(12)]:
                np.testing.assert_allclose(acts[k][:1, rows].numpy(), g["act_" + k], atol=2e-4, rtol=0, err_msg=k)
    mac = ob.macs_per_frame()
    assert abs(sum(mac.values()) - 30.9e9) / 30.9e9 < 0.02          # ~30.9 GMAC / frame
