"""GPU: the rows either side of the network: device crop / resize / normalise,
device state update, and the lock-step batched tracker -- each against the host statement of the same
reference code (vittracker_amd/host_ops.py, the tracker plugin)."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _nat(geom=(128, 256), B=8):
    from vittracker_amd import native, synth
    m = native.Model(geom[0], geom[1], max_batch=B)
    m.load_state_dict(synth.synth_state_dict(0, len_z=(geom[0] // 16) ** 2, len_x=(geom[1] // 16) ** 2))
    return m


def _host_crop(frame, box, factor, T):
    """sample_target + Preprocessor.process, float32 like the reference (data_utils.py:13-14)."""
    from vittracker_amd.host_ops import sample_target
    crop, rf, _ = sample_target(frame, list(box), factor, output_sz=T)
    mean = np.array(MEAN, np.float32).reshape(3, 1, 1)
    std = np.array(STD, np.float32).reshape(3, 1, 1)
    t = crop.astype(np.float32).transpose(2, 0, 1)
    # torch on a GPU computes `img / 255.0` as img * float32(1/255) (division by a CPU scalar)
    return ((t * (np.float32(1.0) / np.float32(255.0))) - mean) / std, rf


@pytest.mark.parametrize("T,factor", [(256, 4.0), (128, 2.0), (64, 2.0)])
def test_device_crop_equals_host_crop(T, factor):
    """Boxes inside, across every border, tiny, huge, fractional (round-half-even cases included)."""
    import torch
    rs = np.random.RandomState(0)
    H, W = 120, 160
    boxes = [[40, 30, 20, 24], [-5, -8, 30, 30], [130, 90, 40, 36], [0, 0, 8, 8], [60.5, 41.5, 11, 7],
             [150, 110, 30, 30], [10.25, 77.75, 5.5, 3.25], [70, 50, 90, 80]]
    B = len(boxes)
    frames = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    m = _nat(B=B)
    crops, rf = m.crop(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), factor, T, MEAN, STD)
    crops, rf = crops.cpu().numpy(), rf.cpu().numpy()
    for b in range(B):
        want, want_rf = _host_crop(frames[b], boxes[b], factor, T)
        assert rf[b] == want_rf
        np.testing.assert_array_equal(crops[b], want, err_msg=f"box {boxes[b]}")


def test_device_crop_random_boxes_on_odd_frame_sizes():
    """The crop kernel reads a sample's two source columns as one unaligned 8-byte load: random boxes (inside, across borders, at the
    corners, at the very end of the batch's memory) on frames whose row length is not a multiple of anything."""
    import torch
    rs = np.random.RandomState(11)
    H, W, T = 37, 53, 64
    boxes = [[rs.uniform(-10, W), rs.uniform(-10, H), rs.uniform(2, 40), rs.uniform(2, 40)] for _ in range(45)]
    boxes += [[W - 6, H - 5, 6, 5], [W - 3.5, H - 3.5, 3, 3], [0, 0, W, H], [W - 1, H - 1, 1, 1], [W - 12, H - 9, 12.5, 9.5]]
    B = len(boxes)
    frames = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    m = _nat(B=B)
    for factor in (2.0, 4.0):
        crops, rf = m.crop(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), factor, T, MEAN, STD)
        crops, rf = crops.cpu().numpy(), rf.cpu().numpy()
        for b in range(B):
            want, want_rf = _host_crop(frames[b], boxes[b], factor, T)
            assert rf[b] == want_rf
            np.testing.assert_array_equal(crops[b], want, err_msg=f"box {boxes[b]} factor {factor}")


def test_byte_load_crop_form_equals_the_host_crop():
    """vt_create's self test crops a known frame with the 8-byte unaligned-load kernel and with its byte-load twin and falls back
    to the twin on any difference (a device without unaligned access).  Here the twin is forced (VT_CROP_BYTES=1, read by the
    self test; a process decides once, so this runs in a child) and held to the host statement bit for bit at the tracker's sizes
    and at one that is a multiple of nothing."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_pipeline as T
rs = np.random.RandomState(5)
H, W = 37, 53
boxes = [[rs.uniform(-10, W), rs.uniform(-10, H), rs.uniform(2, 40), rs.uniform(2, 40)] for _ in range(20)] + [[W - 3.5, H - 3.5, 3, 3], [0, 0, W, H], [W - 1, H - 1, 1, 1]]
frames = rs.randint(0, 256, (len(boxes), H, W, 3)).astype(np.uint8)
m = T._nat(B=len(boxes))
for S in (64, 128, 20):
    crops, rf = m.crop(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), 2.0, S, T.MEAN, T.STD)
    crops = crops.cpu().numpy()
    for b in range(len(boxes)):
        want, want_rf = T._host_crop(frames[b], boxes[b], 2.0, S)
        assert np.array_equal(crops[b], want), (S, boxes[b])
print("BYTES-OK")
""" % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, VT_CROP_BYTES="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "BYTES-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("form", ["0", "2", "4"])
def test_every_crop_kernel_form_equals_the_host_crop(form):
    """VT_CROP_FAST selects the tracker step's crop kernel: 1 (default; the tests above) = crop_fast_kernel with one group of 256 items per
    workgroup, 2 / 4 = its software-pipelined multi-group walks, 0 = crop_kernel (which also serves every size that is not a multiple
    of 4).  A process reads the switch once, so each form runs in a child; sizes: the tracker's, one with a ragged last group (T = 20
    -> 100 items) and one that only crop_kernel takes (T = 30)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_pipeline as T
rs = np.random.RandomState(7)
H, W = 61, 83
boxes = [[rs.uniform(-10, W), rs.uniform(-10, H), rs.uniform(2, 60), rs.uniform(2, 60)] for _ in range(12)] + [[W - 3.5, H - 3.5, 3, 3], [0, 0, W, H], [W - 1, H - 1, 1, 1], [-4, -4, 9, 9]]
frames = rs.randint(0, 256, (len(boxes), H, W, 3)).astype(np.uint8)
m = T._nat(B=len(boxes))
for S, factor in ((64, 2.0), (128, 4.0), (256, 4.0), (20, 2.0), (30, 2.0)):
    crops, rf = m.crop(torch.from_numpy(frames).cuda(), torch.tensor(boxes, dtype=torch.float64).cuda(), factor, S, T.MEAN, T.STD)
    crops, rf = crops.cpu().numpy(), rf.cpu().numpy()
    for b in range(len(boxes)):
        want, want_rf = T._host_crop(frames[b], boxes[b], factor, S)
        assert rf[b] == want_rf and np.array_equal(crops[b], want), (S, boxes[b])
print("FORM-OK")
""" % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, VT_CROP_FAST=form)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FORM-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_device_crop_geometry_matches_reference_fixture():
    """vt_crop's crop / pad geometry against the fixtures the REFERENCE's sample_target produced
    (tests/golden/ref_crop_geometry.npz).  With out_size == crop side the fixed-point resize is the
    identity, and with mean 0 / std 1 the output is pixel * (1/255): the uint8 crop is recovered exactly."""
    import torch
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "ref_crop_geometry.npz"))
    H, W = int(g["image_hw"][0]), int(g["image_hw"][1])
    im = np.random.RandomState(int(g["image_seed"])).randint(0, 256, (H, W, 3)).astype(np.uint8)
    m = _nat(B=1)
    fr = torch.from_numpy(im[None]).cuda()
    for i in range(int(g["n"])):
        want = g[f"crop_{i}"]
        T = want.shape[0]
        st = torch.tensor(g["boxes"][i][None], dtype=torch.float64).cuda()
        crop, rf = m.crop(fr, st, float(g["factors"][i]), T, [0, 0, 0], [1, 1, 1])
        assert float(rf[0]) == 1.0
        got = np.rint(crop[0].cpu().numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
        np.testing.assert_array_equal(got, want, err_msg=f"case {i} box {g['boxes'][i]}")


def test_device_crop_poisons_a_too_small_box_and_the_batched_tracker_raises():
    import torch
    m = _nat(B=2)
    fr = torch.zeros(2, 32, 32, 3, dtype=torch.uint8, device="cuda")
    st = torch.tensor([[5, 5, 0, 0], [5, 5, 8, 8]], dtype=torch.float64).cuda()
    crop, rf = m.crop(fr, st, 4.0, 64, MEAN, STD)
    assert torch.isnan(rf[0]) and torch.isnan(crop[0]).all() and not torch.isnan(crop[1]).any() and float(rf[1]) == 2.0
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_g128")
    p.allow_synthetic_weights, p.debug = True, 0
    bt = BatchedVitTracker(p, 2)
    with pytest.raises(Exception, match="Too small bounding box"):
        bt.initialize(np.zeros((2, 32, 32, 3), np.uint8), [[5, 5, 0, 0], [5, 5, 8, 8]])


def test_device_state_update_equals_host_tail():
    import torch
    from vittracker_amd.host_ops import clip_box
    from oracle import vt_oracle_np as onp
    rs = np.random.RandomState(1)
    B, H, W, S = 64, 480, 640, 256
    states = np.concatenate([rs.uniform(-20, 600, (B, 2)), rs.uniform(5, 200, (B, 2))], 1)
    hann = rs.uniform(0, 1, (B, 4)).astype(np.float32)
    rf = S / np.ceil(np.sqrt(states[:, 2] * states[:, 3]) * 4.0)
    m = _nat(B=B)
    st = torch.from_numpy(states.copy()).cuda()
    m.update_state(torch.from_numpy(hann).cuda(), torch.from_numpy(rf).cuda(), st, S, H, W, margin=10)
    got = st.cpu().numpy()
    for b in range(B):
        pred = (torch.from_numpy(hann[b]).view(1, 4).mean(dim=0) * S / float(rf[b])).tolist()   # tracker :107-109
        want = clip_box(onp.map_box_back(list(states[b]), pred, float(rf[b]), S), H, W, margin=10)
        np.testing.assert_allclose(got[b], want, rtol=1e-6, atol=1e-6)


def test_batched_tracker_equals_independent_plugin_trackers():
    """B sequences in lock-step with device-resident state == B separate Vit_dist plugin objects."""
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    from vittracker_amd.tracker.vit_dist import get_tracker_class
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_noKD")
    p.allow_synthetic_weights = True
    p.debug = 0
    B, n, H, W = 4, 5, 200, 260
    rs = np.random.RandomState(3)
    vids = rs.randint(0, 256, (n, B, H, W, 3)).astype(np.uint8)
    boxes0 = [[60 + 10 * b, 50 + 5 * b, 40, 30 + 2 * b] for b in range(B)]
    bt = BatchedVitTracker(p, B)
    bt.initialize(vids[0], boxes0)
    singles = []
    for b in range(B):
        t = get_tracker_class()(p, "synthetic")
        t.initialize(vids[0, b], {"init_bbox": list(map(float, boxes0[b]))})
        singles.append(t)
    for f in range(1, n):
        out = bt.track(vids[f])
        for b in range(B):
            o = singles[b].track(vids[f, b])
            np.testing.assert_allclose(out["target_bbox"][b].numpy(), o["target_bbox"], rtol=1e-6, atol=1e-5)
            assert abs(float(out["confidence"][b]) - o["confidence"]) < 1e-6
    # no-sync mode: same final state after replaying the sequence without reading back per frame
    bt2 = BatchedVitTracker(p, B)
    bt2.initialize(vids[0], boxes0)
    for f in range(1, n):
        last = bt2.track(vids[f], sync=False)
    np.testing.assert_array_equal(last["target_bbox"].cpu().numpy(), out["target_bbox"].numpy())


@pytest.mark.parametrize("geom,B", [(128, 3), (256, 2), (128, 200), (256, 180)])
def test_track_step_equals_its_three_calls(geom, B, monkeypatch):
    """vt_track_step with the fp32 crop (VT_TRACK_U8=0, read at vt_create; the default since round 6 hands the stem a uint8 patch:
    tests/test_gpu_patch_u8.py) -- crop -> network on the cached template -> tail, which the decoding lane of every head form runs
    itself: the decode kernel on the small-batch path, the fused heads at 200 -- == vt_crop + vt_forward + vt_update_state_record,
    bit for bit."""
    import torch
    from vittracker_amd import native, synth
    monkeypatch.setenv("VT_TRACK_U8", "0")
    m = native.Model(geom // 2, geom, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(4, len_z=(geom // 32) ** 2, len_x=(geom // 16) ** 2))
    H, W = 120, 160
    rs = np.random.RandomState(8)
    frames = torch.from_numpy(rs.randint(0, 256, (3, B, H, W, 3)).astype(np.uint8)).cuda()
    boxes = np.stack([[30 + (b % 40), 20 + (b % 30), 30 + (b % 7), 24 + (b % 5)] for b in range(B)]).astype(np.float64)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    res = []
    for fused in (False, True):
        states = torch.from_numpy(boxes).cuda()
        z, rf = m.crop(frames[0], states, 2.0, geom // 2, mean, std)
        m.set_template(z)
        x = torch.empty(B, 3, geom, geom, device="cuda")
        out = native.Outputs(B, geom // 16, "cuda")
        rec = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        recs = []
        for f in (1, 2):
            if fused:
                m.track_step(frames[f], states, 4.0, mean, std, x, rf, out, record=rec)
            else:
                m.crop(frames[f], states, 4.0, geom, mean, std, out=x, resize_factor=rf)
                m.forward(None, x, out=out)
                m.update_state_record(out.hann_boxes, out.conf, rf, states, rec, geom, H, W, margin=10)
            recs.append((rec.clone(), states.clone(), out.hann_boxes.clone(), out.pred_boxes.clone(), out.score_map.clone()))
        res.append(recs)
    for a, b in zip(*res):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    assert torch.isfinite(res[1][-1][0]).all() and not torch.equal(res[1][0][1], res[1][1][1])
    m.close()


def test_frames_read_in_place_from_pinned_memory_equal_uploaded_frames(monkeypatch):
    """Small host frames are copied by the CPU into pinned slots that the crop kernel reads over the bus (no upload); larger ones are
    uploaded into device slots.  Same kernels on the same pixels: boxes and confidences must be identical, with and without a
    synchronisation per step (two slots, the writer of a slot waits for the step that last read it)."""
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_g128")
    p.allow_synthetic_weights = True
    p.debug = 0
    B, n, H, W = 2, 7, 120, 160
    rs = np.random.RandomState(5)
    vids = rs.randint(0, 256, (n, B, H, W, 3)).astype(np.uint8)
    boxes0 = [[40 + 10 * b, 30 + 5 * b, 30, 24 + 2 * b] for b in range(B)]
    res = {}
    for mode, limit in (("pinned", 2 << 20), ("uploaded", 0)):
        monkeypatch.setattr(BatchedVitTracker, "ZERO_COPY_MAX_BYTES", limit)
        for sync in (True, False):
            bt = BatchedVitTracker(p, B)
            bt.initialize(vids[0], boxes0)
            assert bt.frames[0].is_cuda == (mode == "uploaded")
            outs = []
            for f in range(1, n):
                o = bt.track(vids[f], sync=sync)
                if sync:
                    outs.append((o["target_bbox"].numpy().copy(), o["confidence"].numpy().copy()))
            if not sync:
                outs = [(o["target_bbox"].cpu().numpy(), o["confidence"].cpu().numpy())]
            res[(mode, sync)] = outs
    for sync in (True, False):
        for (b0, c0), (b1, c1) in zip(res[("pinned", sync)], res[("uploaded", sync)]):
            np.testing.assert_array_equal(b0, b1)
            np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(res[("pinned", False)][0][0], res[("pinned", True)][-1][0])


def test_track_chunk_equals_frame_by_frame_tracking():
    """n frames per graph launch (crop -> forward -> state update, n times, in one captured graph) give, frame by frame, exactly
    what track() gives -- from host frames and from a device buffer, and again after re-initialising (the captured graphs read
    the template cache, which initialize() refreshes in place)."""
    import torch
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.native import VtError
    from vittracker_amd.parameter import vit_dist as P
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_noKD")
    p.allow_synthetic_weights = True
    B, n, H, W = 3, 7, 180, 240
    rs = np.random.RandomState(5)
    vids = rs.randint(0, 256, (n, B, H, W, 3)).astype(np.uint8)
    boxes0 = [[50 + 12 * b, 40 + 6 * b, 44, 30 + 3 * b] for b in range(B)]
    ref = BatchedVitTracker(p, B)
    with pytest.raises(VtError, match="before initialize"):
        ref.track_chunk(vids[1:3])
    ref.initialize(vids[0], boxes0)
    want_b, want_c = [], []
    for f in range(1, n):
        o = ref.track(vids[f])
        want_b.append(o["target_bbox"].numpy().copy()); want_c.append(o["confidence"].numpy().copy())
    want_b, want_c = np.stack(want_b), np.stack(want_c)

    bt = BatchedVitTracker(p, B)
    bt.initialize(vids[0], boxes0)
    o1 = bt.track_chunk(vids[1:4])                       # host frames, 3 per launch
    dev = torch.from_numpy(vids[4:7]).cuda()
    o2 = bt.track_chunk(dev)                             # device frames, used in place
    np.testing.assert_array_equal(np.concatenate([o1["target_bbox"].numpy(), o2["target_bbox"].numpy()]), want_b)
    np.testing.assert_array_equal(np.concatenate([o1["confidence"].numpy(), o2["confidence"].numpy()]), want_c)
    assert bt.frame_id == 6
    # a second pass over the same buffers after re-initialising on another template: graphs are reused, results follow the new template
    boxes1 = [[80 + 5 * b, 60, 36, 36] for b in range(B)]
    ref.initialize(vids[2], boxes1)
    bt.initialize(vids[2], boxes1)
    w = np.stack([ref.track(vids[f])["target_bbox"].numpy().copy() for f in range(4, 7)])
    got = bt.track_chunk(dev)["target_bbox"].numpy()
    np.testing.assert_array_equal(got, w)
    assert len(bt._chunk_graphs) == 2
    with pytest.raises(ValueError):
        bt.track_chunk(vids[1:3, :2])


# Tracker-realistic crop sides -> the two network input sizes, incl. exact 2x / 4x downsamples, identity and upsamples
CV2_CASES = [(90, 128), (131, 128), (256, 128), (512, 128), (700, 128), (64, 128), (128, 128),
             (97, 256), (300, 256), (512, 256), (1024, 256), (333, 256), (701, 256), (128, 256)]


def test_resize_equals_cv2_when_the_box_has_opencv():
    """SURVEY 8(f)1 pin: `cv.resize` (default INTER_LINEAR on uint8, lib/train/data/processing_utils.py:68) against BOTH the numpy
    port (host_ops.resize_bilinear_u8) and the device kernel (vt_crop), bit for bit.  Runs only where OpenCV is importable; the
    build image and -- as far as known -- the GPU box do not have it, in which case the skip reason says so and the resize
    stays parity-unpinned."""
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"cv2 is not importable on this box ({type(e).__name__}: {e}): the bilinear resize stays unpinned against OpenCV")
    import torch
    from vittracker_amd.host_ops import resize_bilinear_u8
    rs = np.random.RandomState(9)
    m = _nat(B=1)
    for S, T in CV2_CASES:
        im = rs.randint(0, 256, (S + 24, S + 24, 3)).astype(np.uint8)
        x1 = y1 = 12                                               # the crop lies strictly inside the frame: no padding involved
        want = cv2.resize(im[y1:y1 + S, x1:x1 + S], (T, T))
        np.testing.assert_array_equal(resize_bilinear_u8(np.ascontiguousarray(im[y1:y1 + S, x1:x1 + S]), T, T), want, err_msg=f"port {S}->{T}")
        # a box whose crop is exactly that square: side = ceil(sqrt(w h) * factor) = S with w = h = S / 2, factor 2
        box = torch.tensor([[x1 + S / 4.0, y1 + S / 4.0, S / 2.0, S / 2.0]], dtype=torch.float64).cuda()
        crop, rf = m.crop(torch.from_numpy(im[None]).cuda(), box, 2.0, T, [0, 0, 0], [1, 1, 1])
        assert float(rf[0]) == T / S
        got = np.rint(crop[0].cpu().numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
        np.testing.assert_array_equal(got, want, err_msg=f"vt_crop {S}->{T}")
