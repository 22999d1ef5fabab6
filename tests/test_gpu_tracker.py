"""GPU: the tracker plugin end to end (initialize / track on a synthetic moving target) against a
tracker whose device step is the CPU oracle -- same crops, same tail, boxes must agree."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _params(yaml_name):
    from vittracker_amd.parameter import vit_dist as P
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters(yaml_name)
    p.allow_synthetic_weights = True     # no trained checkpoint exists (.MISSING_LARGE_BLOBS)
    p.debug = 0
    return p


def _video(n, H=240, W=320, seed=0):
    rs = np.random.RandomState(seed)
    bg = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    frames, boxes = [], []
    for t in range(n):
        f = bg.copy()
        x, y = 100 + 3 * t, 80 + 2 * t
        f[y:y + 40, x:x + 50] = rs.randint(0, 256, (40, 50, 3))
        frames.append(f)
        boxes.append([x, y, 50, 40])
    return frames, boxes


@pytest.mark.parametrize("host_crop", [False, True])
@pytest.mark.parametrize("yaml_name", ["vit_48_h32_noKD", "vit_48_h32_g128"])
def test_track_matches_oracle_driven_tracker(yaml_name, host_crop):
    import torch
    from oracle import vt_oracle_np as onp
    from vittracker_amd.host_ops import clip_box, sample_target
    from vittracker_amd.tracker.vit_dist import get_tracker_class

    p = _params(yaml_name)
    p.host_crop = host_crop       # False (default): whole frame step on the device; True: the reference's host crop + one graph replay
    trk = get_tracker_class()(p, "synthetic")
    frames, boxes = _video(6)
    assert trk.initialize(frames[0], {"init_bbox": boxes[0]}) is None
    sd = {k: v.numpy() for k, v in trk.network.state_dict().items()}
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    prep = lambda a: ((a.astype(np.float32).transpose(2, 0, 1)[None] / np.float32(255.0)) - mean) / std  # noqa: E731
    z_arr, _, _ = sample_target(frames[0], boxes[0], p.template_factor, output_sz=p.template_size)
    state = list(boxes[0])
    for f in frames[1:]:
        out = trk.track(f)                       # info=None, as run_video calls it (evaluation/tracker.py:228)
        assert set(out) == {"target_bbox", "confidence"} and isinstance(out["confidence"], float)
        # oracle-driven tail on the same state
        x_arr, rf, _ = sample_target(f, state, p.search_factor, output_sz=p.search_size)
        ref = onp.forward(sd, prep(z_arr), prep(x_arr))
        pred = (ref["hann_boxes"][0] * np.float32(p.search_size) / np.float32(rf)).tolist()
        want = clip_box(onp.map_box_back(state, pred, rf, p.search_size), f.shape[0], f.shape[1], margin=10)
        margin = onp.top2_margin(ref["score_map"] * onp.hann2d(trk.feat_sz))[0]
        if margin > 1e-3:
            assert out["target_bbox"] == pytest.approx(want, abs=2e-3), (margin, out, want)
            assert out["confidence"] == pytest.approx(float(ref["conf"][0]), abs=1e-4)
        state = out["target_bbox"]               # follow the HIP tracker's state either way
        assert all(isinstance(v, float) or isinstance(v, int) for v in state)


def test_missing_checkpoint_raises_like_reference():
    from vittracker_amd.tracker.vit_dist import get_tracker_class
    p = _params("vit_48_h32_noKD")
    p.allow_synthetic_weights = False
    with pytest.raises(FileNotFoundError):
        get_tracker_class()(p, "synthetic")


@pytest.mark.parametrize("host_crop", [False, True])
def test_save_all_boxes_follows_the_reference_map_back(host_crop):
    """lib/test/tracker/vit_dist.py:141-146: `all_boxes` = the windowed prediction scaled to crop pixels and mapped back relative
    to the ALREADY UPDATED state, un-clipped (not the clipped state).  The target sits at the frame border so that clipping
    changes the state."""
    import torch
    from oracle import vt_oracle_np as onp
    from vittracker_amd.host_ops import sample_target
    from vittracker_amd.tracker.vit_dist import get_tracker_class
    p = _params("vit_48_h32_g128")
    p.host_crop = host_crop
    p.save_all_boxes = True
    trk = get_tracker_class()(p, "synthetic")
    rs = np.random.RandomState(3)
    frames = [rs.randint(0, 256, (120, 160, 3)).astype(np.uint8) for _ in range(4)]
    init = [2, 3, 30, 24]
    out0 = trk.initialize(frames[0], {"init_bbox": init})
    assert out0 == {"all_boxes": init}
    sd = {k: v.numpy() for k, v in trk.network.state_dict().items()}
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    prep = lambda a: ((a.astype(np.float32).transpose(2, 0, 1)[None] / np.float32(255.0)) - mean) / std  # noqa: E731
    z_arr, _, _ = sample_target(frames[0], init, p.template_factor, output_sz=p.template_size)
    state, differs = list(init), 0
    for f in frames[1:]:
        out = trk.track(f)
        assert set(out) == {"target_bbox", "all_boxes"} and len(out["all_boxes"]) == 4
        x_arr, rf, _ = sample_target(f, state, p.search_factor, output_sz=p.search_size)
        ref = onp.forward(sd, prep(z_arr), prep(x_arr))
        new = out["target_bbox"]
        if onp.top2_margin(ref["score_map"] * onp.hann2d(trk.feat_sz))[0] > 1e-3:
            pb = torch.from_numpy(ref["hann_boxes"][:1].astype(np.float32)) * p.search_size / rf        # (1,4) float32, as :143
            half = 0.5 * p.search_size / rf
            cxp, cyp = new[0] + 0.5 * new[2], new[1] + 0.5 * new[3]                                     # the updated state (:159)
            cx, cy, w, h = pb.unbind(-1)
            want = torch.stack([cx + (cxp - half) - 0.5 * w, cy + (cyp - half) - 0.5 * h, w, h], -1).view(-1).tolist()
            assert out["all_boxes"] == pytest.approx(want, abs=2e-3)
            differs += any(abs(a - b) > 1e-3 for a, b in zip(out["all_boxes"], new))
        state = new
    assert differs > 0, "the case never separated all_boxes from the clipped state"
