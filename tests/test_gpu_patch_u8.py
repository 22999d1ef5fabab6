"""GPU: the uint8-patch path.  vt_crop_u8 writes the array the reference's sample_target returns
(lib/train/data/processing_utils.py:12-79), the stems read it and apply Preprocessor.process (lib/test/tracker/data_utils.py:11-17)
folded into their first layer.  Bars: the integer patch bit-exact against the host statement of sample_target and the reference's
own crop fixtures; the network on patches within 1e-5 of the reference model's outputs (tests/golden/ref_u8_*.npz) and of
vt_crop + vt_forward; vt_track_step == vt_crop_u8 + vt_forward_u8 + vt_update_state_record bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, REPO, load_u8_case, u8_golden_files

pytestmark = pytest.mark.gpu

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
TOL = 1e-5          # maps and boxes: uint8 path against the reference / against the fp32-crop path


def _model(geom, B, seed=0, **kw):
    from vittracker_amd import native, synth
    m = native.Model(geom // 2, geom, max_batch=B, **kw)
    m.load_state_dict(synth.synth_state_dict(seed, len_z=(geom // 32) ** 2, len_x=(geom // 16) ** 2))
    return m


def _host_patch(frame, box, factor, T):
    from vittracker_amd.host_ops import sample_target
    patch, rf, _ = sample_target(frame, list(box), factor, output_sz=T)
    return patch, rf


@pytest.mark.parametrize("T,factor", [(256, 4.0), (128, 2.0), (64, 2.0), (20, 2.0), (30, 4.0)])
def test_crop_u8_is_the_patch_sample_target_returns(T, factor):
    """Boxes inside, across every border, tiny, huge, fractional; the tracker's sizes, a ragged one (T = 20) and one only crop_kernel
    takes (T = 30): every byte of the (T,T,3) patch and the resize factor."""
    import torch
    rs = np.random.RandomState(0)
    H, W = 120, 160
    boxes = [[40, 30, 20, 24], [-5, -8, 30, 30], [130, 90, 40, 36], [0, 0, 8, 8], [60.5, 41.5, 11, 7],
             [150, 110, 30, 30], [10.25, 77.75, 5.5, 3.25], [70, 50, 90, 80], [W - 3.5, H - 3.5, 3, 3], [W - 1, H - 1, 1, 1]]
    boxes += [[rs.uniform(-10, W), rs.uniform(-10, H), rs.uniform(2, 60), rs.uniform(2, 60)] for _ in range(14)]
    B = len(boxes)
    frames = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    m = _model(256, B)
    patch, rf = m.crop_u8(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), factor, T)
    patch, rf = patch.cpu().numpy(), rf.cpu().numpy()
    for b in range(B):
        want, want_rf = _host_patch(frames[b], boxes[b], factor, T)
        assert rf[b] == want_rf
        np.testing.assert_array_equal(patch[b], want, err_msg=f"box {boxes[b]}")
    m.close()


def test_crop_u8_geometry_matches_reference_fixture():
    """vt_crop_u8 against the crops the REFERENCE's own sample_target produced (tests/golden/ref_crop_geometry.npz; out_size = crop
    side, where the fixed-point resize is the identity): the patch IS the fixture, byte for byte."""
    import torch
    g = np.load(os.path.join(GOLDEN_DIR, "ref_crop_geometry.npz"))
    H, W = int(g["image_hw"][0]), int(g["image_hw"][1])
    im = np.random.RandomState(int(g["image_seed"])).randint(0, 256, (H, W, 3)).astype(np.uint8)
    m = _model(256, 1)
    fr = torch.from_numpy(im[None]).cuda()
    for i in range(int(g["n"])):
        want = g[f"crop_{i}"]
        st = torch.tensor(g["boxes"][i][None], dtype=torch.float64).cuda()
        patch, rf = m.crop_u8(fr, st, float(g["factors"][i]), want.shape[0])
        assert float(rf[0]) == 1.0
        np.testing.assert_array_equal(patch[0].cpu().numpy(), want, err_msg=f"case {i} box {g['boxes'][i]}")
    m.close()


def test_the_fast_crop_form_is_the_one_this_gpu_runs():
    """The device self test falls back to the byte-load twin on ANY mismatch, which would hide a bug of the fast form (the v_perm /
    v_dot2 / mul_hi path) behind a slower step: on gfx950 the fast form must be the one selected (round-5 advisor)."""
    from vittracker_amd import native
    _model(128, 1).close()
    assert native.crop_form() == 1


@pytest.mark.parametrize("env", [{"VT_CROP_BYTES": "1"}, {"VT_CROP_FAST": "0"}, {"VT_CROP_BAND": "0"}, {"VT_CROP_BAND": "-4"}, {"VT_CROP_BAND": "-2"},
                                 {"VT_CROP_BAND": "-4", "VT_CROP_ALIGNED": "0"}])
def test_other_crop_kernel_forms_write_the_same_patch(env):
    """The byte-load twin, crop_kernel (VT_CROP_FAST=0), crop_fast_kernel at every size (VT_CROP_BAND=0) and crop_band_kernel forced at
    every batch (-4 / -2: four / two items per thread; VT_CROP_ALIGNED=0: byte-aligned windows) -- the library picks between the last
    two by how many workgroups the batch gives a CU -- each in a child (a process reads the switches once), uint8 patch AND fp32 crop."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_patch_u8 as T
rs = np.random.RandomState(5)
H, W = 37, 53
boxes = [[rs.uniform(-10, W), rs.uniform(-10, H), rs.uniform(2, 40), rs.uniform(2, 40)] for _ in range(20)] + [[W - 3.5, H - 3.5, 3, 3], [0, 0, W, H], [W - 1, H - 1, 1, 1]]
frames = rs.randint(0, 256, (len(boxes), H, W, 3)).astype(np.uint8)
m = T._model(128, len(boxes))
import test_gpu_pipeline as P
for S in (64, 128, 256, 20, 30):
    patch, rf = m.crop_u8(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), 2.0, S)
    crop, rf2 = m.crop(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), 2.0, S, T.MEAN, T.STD)
    patch, crop = patch.cpu().numpy(), crop.cpu().numpy()
    for b in range(len(boxes)):
        want, want_rf = T._host_patch(frames[b], boxes[b], 2.0, S)
        assert np.array_equal(patch[b], want) and float(rf[b]) == want_rf, (S, boxes[b])
        assert np.array_equal(crop[b], P._host_crop(frames[b], boxes[b], 2.0, S)[0]) and float(rf2[b]) == want_rf, (S, boxes[b])
# the batch's LAST frame is the one whose windows may end at the buffer's end (crop_band_kernel: the aligned form everywhere but in a band that
# reads the frame's last row): every kind of box takes the last slot once
for rot in (2, 21, 22):
    bx, fr = boxes[rot:] + boxes[:rot], np.concatenate([frames[rot:], frames[:rot]])
    for S in (128, 256):
        patch, rf = m.crop_u8(torch.from_numpy(fr).cuda(), torch.tensor(bx, dtype=torch.float64).cuda(), 2.0, S)
        patch = patch.cpu().numpy()
        for b in (len(bx) - 2, len(bx) - 1):
            want, want_rf = T._host_patch(fr[b], bx[b], 2.0, S)
            assert np.array_equal(patch[b], want) and float(rf[b]) == want_rf, (rot, S, bx[b])
print("FORM-OK")
""" % (REPO, os.path.join(REPO, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FORM-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_crop_u8_too_small_box_poisons_the_resize_factor():
    import torch
    m = _model(128, 2)
    fr = torch.full((2, 32, 32, 3), 7, dtype=torch.uint8, device="cuda")
    st = torch.tensor([[5, 5, 0, 0], [5, 5, 8, 8]], dtype=torch.float64).cuda()
    patch, rf = m.crop_u8(fr, st, 4.0, 64)
    assert torch.isnan(rf[0]) and int(patch[0].max()) == 0 and float(rf[1]) == 2.0 and int(patch[1].max()) == 7
    m.close()


# ------------------------------------------------------------------------------------------------ the network on patches
# (geom, B): each selects another stem form -- stem_a (small batches), stem_fused (G128 large), stem_stream (G256 large)
FORMS = [(128, 3), (128, 96), (128, 200), (256, 2), (256, 180)]


@pytest.mark.parametrize("path", u8_golden_files(), ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("form_batch", [0, 256])
def test_forward_u8_matches_the_reference_on_normalised_patches(path, form_batch):
    """The reference model's outputs on Preprocessor-normalised uint8 patches (make_golden_u8.py) against vt_forward_u8 on the raw
    patches: small-batch forms and (vt_set_form_batch(256)) the one-workgroup-per-frame forms the tracker step runs at 256."""
    import torch
    from vittracker_amd import native
    g, sd, z, patches = load_u8_case(path)
    geom = patches.shape[1]
    m = native.Model(geom // 2, geom, max_batch=patches.shape[0])
    m.load_state_dict(sd)
    if form_batch:
        m.set_form_batch(form_batch)
    assert m.patch_u8_supported(patches.shape[0])
    zd, pd = torch.from_numpy(z).cuda(), torch.from_numpy(patches).cuda()
    for cached in (False, True):
        if cached:
            m.set_template(zd)
        out = m.forward_u8(None if cached else zd, pd)
        for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
            got = getattr(out, k).cpu().numpy()
            err = float(np.abs(got - g[k].reshape(got.shape)).max())          # the reference's pred_boxes are (B,1,4)
            assert err < TOL, (k, cached, err)
    m.close()


@pytest.mark.parametrize("geom,B", FORMS)
def test_forward_u8_matches_forward_on_the_normalised_crop(geom, B):
    """vt_forward_u8(patch) against vt_forward(Preprocessor.process(patch)) -- the reference's three rounded fp32 operations per
    value, computed with torch on the GPU as the reference does -- on a full batch of every stem form: token rows, maps, boxes."""
    import torch
    from vittracker_amd import native, synth
    m = _model(geom, B, seed=3)
    assert m.patch_u8_supported(B)
    patches = torch.from_numpy(synth.synth_patches(11, B, geom)).cuda()
    z = torch.from_numpy(synth.synth_inputs(11, B, geom // 2, geom)[0]).cuda()
    mean, std = torch.tensor(MEAN).view(1, 3, 1, 1).cuda(), torch.tensor(STD).view(1, 3, 1, 1).cuda()
    x = (((patches.float().permute(0, 3, 1, 2) / 255.0) - mean) / std).contiguous()          # data_utils.py:13-14
    np.testing.assert_array_equal(x.cpu().numpy(), synth.normalise_patches(patches.cpu().numpy()))
    m.set_template(z)
    want = m.forward(None, x)
    got = m.forward_u8(None, patches)
    want_z = m.forward(z, x)
    got_z = m.forward_u8(z, patches)
    for a, b in ((got, want), (got_z, want_z)):
        for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
            err = float((getattr(a, k) - getattr(b, k)).abs().max())
            assert err < TOL, (k, err)
    # the stage itself: search rows of the token matrix
    tok = m.stem(z, x)
    tok8 = torch.zeros_like(tok)
    m.stem_u8(patches, tok8)
    assert float((tok8[:, m.len_z:] - tok[:, m.len_z:]).abs().max()) < TOL
    assert float(tok8[:, :m.len_z].abs().max()) == 0.0          # template rows untouched
    m.close()


@pytest.mark.parametrize("geom,B", [(128, 3), (128, 200), (256, 2), (256, 180)])
def test_f16_build_reads_patches_too(geom, B):
    """The f16-contraction build (BASELINE config 5) runs the same uint8 stem forms (layer 1 is fp32 vector work in either build):
    vt_forward_u8 against the fp32 build's vt_forward on the normalised crop, at the f16 build's tolerance (maps 6e-3, boxes 2e-3)."""
    import torch
    from vittracker_amd import native, synth
    sd = synth.synth_state_dict(3, len_z=(geom // 32) ** 2, len_x=(geom // 16) ** 2)
    m32 = native.Model(geom // 2, geom, max_batch=B)
    m32.load_state_dict(sd)
    m16 = native.Model(geom // 2, geom, max_batch=B, precision="f16")
    m16.load_state_dict(sd)
    assert m16.patch_u8_supported(B)
    patches = torch.from_numpy(synth.synth_patches(13, B, geom)).cuda()
    z = torch.from_numpy(synth.synth_inputs(13, B, geom // 2, geom)[0]).cuda()
    x = torch.from_numpy(synth.normalise_patches(patches.cpu().numpy())).cuda()
    want, got = m32.forward(z, x), m16.forward_u8(z, patches)
    for k in ("score_map", "size_map", "offset_map"):
        assert float((getattr(got, k) - getattr(want, k)).abs().max()) < 6e-3, k
    same = (got.pred_boxes - want.pred_boxes).abs().amax(1) < 2e-3          # frames whose argmax agrees (f16 noise may flip a near tie)
    assert float(same.float().mean()) >= 0.9
    m16.set_template(z)
    cached = m16.forward_u8(None, patches)
    assert torch.equal(cached.score_map, got.score_map) and torch.equal(cached.pred_boxes, got.pred_boxes)      # the template cache is exact
    m32.close(); m16.close()


def test_generic_geometry_reads_patches_bit_identically():
    """A geometry without tuned kernels (112 / 224 px: 245 tokens) runs vt_generic.h, whose uint8 stem layer normalises every tap with
    the reference's own three rounded operations: vt_forward_u8 == vt_forward on the normalised crop, bit for bit."""
    import torch
    from vittracker_amd import native, synth
    tz, tx, B = 112, 224, 3
    m = native.Model(tz, tx, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(6, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
    assert m.patch_u8_supported(B)
    patches = torch.from_numpy(synth.synth_patches(14, B, tx)).cuda()
    z = torch.from_numpy(synth.synth_inputs(14, B, tz, tx)[0]).cuda()
    x = torch.from_numpy(synth.normalise_patches(patches.cpu().numpy())).cuda()
    want, got = m.forward(z, x), m.forward_u8(z, patches)
    for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
        assert torch.equal(getattr(got, k), getattr(want, k)), k
    m.set_template(z)
    cached = m.forward_u8(None, patches)
    assert torch.equal(cached.score_map, want.score_map)
    m.close()


def test_stem_pipe_reads_patches():
    """VT_STEM_STREAM=0 at G256 and a large batch selects stem_pipe + stem_b in the fp32 build (the f16 build's default there): its uint8
    form against vt_forward on the normalised crop, in a child (the switch is read at vt_create... per process for safety)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from vittracker_amd import native, synth
geom, B = 256, 180
m = native.Model(geom // 2, geom, max_batch=B)
m.load_state_dict(synth.synth_state_dict(3, len_z=64, len_x=256))
patches = torch.from_numpy(synth.synth_patches(15, B, geom)).cuda()
z = torch.from_numpy(synth.synth_inputs(15, B, geom // 2, geom)[0]).cuda()
x = torch.from_numpy(synth.normalise_patches(patches.cpu().numpy())).cuda()
m.set_template(z)
want, got = m.forward(None, x), m.forward_u8(None, patches)
for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes"):
    err = float((getattr(got, k) - getattr(want, k)).abs().max())
    assert err < 1e-5, (k, err)
print("PIPE-OK")
""" % REPO
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VT_STEM_STREAM="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "PIPE-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_forward_u8_with_another_normalisation():
    """vt_set_normalization refolds layer 1: another mean / std against vt_forward on the crop normalised with them."""
    import torch
    from vittracker_amd import synth
    geom, B = 128, 5
    m = _model(geom, B, seed=2)
    mean, std = [0.4, 0.5, 0.45], [0.2, 0.25, 0.3]
    m.set_normalization(mean, std)
    patches = torch.from_numpy(synth.synth_patches(12, B, geom)).cuda()
    z = torch.from_numpy(synth.synth_inputs(12, B, geom // 2, geom)[0]).cuda()
    x = (((patches.float().permute(0, 3, 1, 2) / 255.0) - torch.tensor(mean).view(1, 3, 1, 1).cuda()) / torch.tensor(std).view(1, 3, 1, 1).cuda()).contiguous()
    want, got = m.forward(z, x), m.forward_u8(z, patches)
    for k in ("score_map", "size_map", "offset_map", "pred_boxes"):
        assert float((getattr(got, k) - getattr(want, k)).abs().max()) < TOL, k
    m.close()


@pytest.mark.parametrize("geom,B", [(128, 3), (256, 2), (128, 200), (256, 180)])
def test_track_step_is_crop_u8_forward_u8_and_the_tail(geom, B):
    """vt_track_step == vt_crop_u8 + vt_forward_u8(z = None) + vt_update_state_record bit for bit (records, states, boxes, maps, the
    patch in the workspace), and within 1e-5 / a hundredth of a pixel of the fp32-crop step (vt_crop + vt_forward + tail)."""
    import torch
    from vittracker_amd import native, synth
    m = _model(geom, B, seed=4)
    H, W = 120, 160
    rs = np.random.RandomState(8)
    frames = torch.from_numpy(rs.randint(0, 256, (3, B, H, W, 3)).astype(np.uint8)).cuda()
    boxes = np.stack([[30 + (b % 40), 20 + (b % 30), 30 + (b % 7), 24 + (b % 5)] for b in range(B)]).astype(np.float64)
    res = {}
    for mode in ("step", "calls", "fp32"):
        states = torch.from_numpy(boxes).cuda()
        z, rf = m.crop(frames[0], states, 2.0, geom // 2, MEAN, STD)
        m.set_template(z)
        x = torch.empty(B, 3, geom, geom, device="cuda")
        patch = torch.empty(B, geom, geom, 3, dtype=torch.uint8, device="cuda")
        out = native.Outputs(B, geom // 16, "cuda")
        rec = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        recs = []
        for f in (1, 2):
            if mode == "step":
                m.track_step(frames[f], states, 4.0, MEAN, STD, x, rf, out, record=rec)
                patch = x.view(torch.uint8).flatten()[: B * geom * geom * 3].view(B, geom, geom, 3).clone()
            elif mode == "calls":
                m.crop_u8(frames[f], states, 4.0, geom, out=patch, resize_factor=rf)
                m.forward_u8(None, patch, out=out)
                m.update_state_record(out.hann_boxes, out.conf, rf, states, rec, geom, H, W, margin=10)
            else:
                m.crop(frames[f], states, 4.0, geom, MEAN, STD, out=x, resize_factor=rf)
                m.forward(None, x, out=out)
                m.update_state_record(out.hann_boxes, out.conf, rf, states, rec, geom, H, W, margin=10)
            recs.append((rec.clone(), states.clone(), out.hann_boxes.clone(), out.pred_boxes.clone(), out.score_map.clone(), patch.clone()))
        res[mode] = recs
    for a, b in zip(res["step"], res["calls"]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    for a, b in zip(res["step"], res["fp32"]):
        assert float((a[4] - b[4]).abs().max()) < TOL and float((a[2] - b[2]).abs().max()) < TOL
        assert float((a[0] - b[0]).abs().max()) < 1e-2          # image pixels (the box scaled by the crop side)
    assert torch.isfinite(res["step"][-1][0]).all() and not torch.equal(res["step"][0][1], res["step"][1][1])
    m.close()


def test_track_step_with_another_mean_falls_back_to_the_fp32_crop(monkeypatch):
    """(mean3, std3) that are not the folded normalisation: the step runs vt_crop + vt_forward (always correct), bit for bit."""
    import torch
    from vittracker_amd import native
    geom, B = 128, 4
    m = _model(geom, B, seed=4)
    H, W = 120, 160
    frames = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)).cuda()
    boxes = np.array([[30 + b, 20 + b, 30, 24] for b in range(B)], np.float64)
    mean, std = [0.5, 0.5, 0.5], [0.25, 0.25, 0.25]
    res = []
    for fused in (True, False):
        states = torch.from_numpy(boxes).cuda()
        z, rf = m.crop(frames[0], states, 2.0, geom // 2, mean, std)
        m.set_template(z)
        x = torch.empty(B, 3, geom, geom, device="cuda")
        out = native.Outputs(B, geom // 16, "cuda")
        rec = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        if fused:
            m.track_step(frames[1], states, 4.0, mean, std, x, rf, out, record=rec)
        else:
            m.crop(frames[1], states, 4.0, geom, mean, std, out=x, resize_factor=rf)
            m.forward(None, x, out=out)
            m.update_state_record(out.hann_boxes, out.conf, rf, states, rec, geom, H, W, margin=10)
        res.append((rec.clone(), out.score_map.clone(), x.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    m.close()


@pytest.mark.parametrize("geom,B", [(128, 3), (256, 2), (128, 200), (256, 130)])
def test_open_loop_step_leaves_the_states_and_writes_the_same_record(geom, B):
    """vt_set_open_loop: the step crops around the caller's boxes, the record (box + confidence) is the closed-loop step's bit for bit,
    states_dev is not written; switching it off restores the closed loop.  Small- and large-batch head forms (both run the tail)."""
    import torch
    from vittracker_amd import native
    m = _model(geom, B, seed=4)
    H, W = 120, 160
    frames = torch.from_numpy(np.random.RandomState(8).randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)).cuda()
    boxes = np.stack([[30 + (b % 40), 20 + (b % 30), 30 + (b % 7), 24 + (b % 5)] for b in range(B)]).astype(np.float64)
    got = {}
    for mode in ("closed", "open", "closed_again"):
        m.set_open_loop(mode == "open")
        states = torch.from_numpy(boxes).cuda()
        z, rf = m.crop(frames[0], states, 2.0, geom // 2, MEAN, STD)
        m.set_template(z)
        x = torch.empty(B, 3, geom, geom, device="cuda")
        out = native.Outputs(B, geom // 16, "cuda")
        rec = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        m.track_step(frames[1], states, 4.0, MEAN, STD, x, rf, out, record=rec)
        torch.cuda.synchronize()
        got[mode] = (rec.clone(), states.clone())
    assert torch.equal(got["open"][0], got["closed"][0]) and torch.equal(got["closed_again"][0], got["closed"][0])
    assert torch.equal(got["open"][1].cpu(), torch.from_numpy(boxes))                    # untouched
    assert torch.equal(got["closed"][1], got["closed"][0][:, :4]) and torch.equal(got["closed_again"][1], got["closed"][1])
    assert not torch.equal(got["closed"][1].cpu(), torch.from_numpy(boxes))
    m.close()


def test_hold_states_is_the_open_loop_in_eager_steps_and_chunk_graphs():
    """BatchedVitTracker.hold_states(True): every step -- eager and inside a captured chunk graph -- searches around the held boxes (the
    same frame gives the same record every time, the states never move); hold_states(False) resumes the closed loop."""
    import torch
    os.environ.setdefault("VITTRACK_PRJ_DIR", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    p = P.parameters("vit_48_h32_g128")
    p.allow_synthetic_weights = True
    p.debug = 0
    B, H, W = 6, 120, 160
    rs = np.random.RandomState(3)
    fr = torch.from_numpy(rs.randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)).cuda()
    boxes = np.stack([[30 + 5 * b, 20 + 4 * b, 30 + b, 24 + b] for b in range(B)]).astype(np.float64)
    bt = BatchedVitTracker(p, B)
    bt.initialize(fr[0], boxes)
    bt.hold_states(True)
    held = bt.states.clone()
    chunk = fr[[1, 1, 0]].contiguous()
    r1 = bt.track_chunk(chunk, sync=True)
    r2 = bt.track_chunk(chunk, sync=True)
    assert torch.equal(bt.states, held)
    c1, c2 = np.asarray(r1["confidence"]), np.asarray(r2["confidence"])
    assert np.array_equal(c1, c2) and np.array_equal(c1[0], c1[1]) and not np.array_equal(c1[0], c1[2])      # same frame, same boxes -> same result
    e1 = bt.track(fr[1], sync=True)
    assert torch.equal(bt.states, held)
    assert np.array_equal(np.asarray(e1["confidence"].cpu()).ravel(), c1[0].ravel())
    bt.hold_states(False)
    bt.track_chunk(chunk, sync=True)
    assert not torch.equal(bt.states, held)
