"""GPU: the failure modes ADVICE r1 named -- each must now fail loudly (or be impossible) instead of
reading freed or out-of-bounds device memory."""
import os

import numpy as np
import pytest

from conftest import GEOMS, REPO

pytestmark = pytest.mark.gpu


def _model(geom="G128", B=4, depth=3):
    from vittracker_amd import native, synth
    tz, tx = GEOMS[geom]
    m = native.Model(tz, tx, depth=depth, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(0, depth=depth, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
    return m


def test_shape_validation_raises_before_any_launch():
    import torch
    from vittracker_amd import native
    m = _model("G128", B=4)
    z, x = torch.zeros(4, 3, 64, 64, device="cuda"), torch.zeros(4, 3, 128, 128, device="cuda")
    m.forward(z, x)
    with pytest.raises(native.VtError, match="expected z"):
        m.forward(torch.zeros(4, 3, 128, 128, device="cuda"), x)            # template at the search size
    with pytest.raises(native.VtError, match="expected z"):
        m.forward(z, x[:2])                                                  # batch mismatch
    with pytest.raises(native.VtError, match="max_batch"):
        m.forward(torch.zeros(5, 3, 64, 64, device="cuda"), torch.zeros(5, 3, 128, 128, device="cuda"))
    with pytest.raises(native.VtError, match="tokens must be"):
        m.blocks(torch.zeros(4, 81, 48, device="cuda"))
    with pytest.raises(native.VtError, match="feat must be"):
        m.head(torch.zeros(4, 16, 48, device="cuda"))
    with pytest.raises(native.VtError, match="states must be"):
        m.crop(torch.zeros(4, 32, 32, 3, dtype=torch.uint8, device="cuda"),
               torch.zeros(3, 4, dtype=torch.float64, device="cuda"), 2.0, 64, [0, 0, 0], [1, 1, 1])
    with pytest.raises(native.VtError, match="output buffer"):
        m.forward(z, x, native.Outputs(2, m.feat_sz, "cuda"))


def test_tracker_refuses_mismatched_test_and_data_sizes():
    from vittracker_amd import native
    from vittracker_amd.parameter import vit_dist as P
    from vittracker_amd.tracker.vit_dist import get_tracker_class
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_noKD")
    p.allow_synthetic_weights, p.debug = True, 0
    p.search_size = 128          # TEST.SEARCH_SIZE without DATA.SEARCH.SIZE
    with pytest.raises(native.VtError, match="differ from the model geometry"):
        get_tracker_class()(p, "synthetic")


def test_graph_is_invalidated_when_its_model_is_resized_or_closed():
    import torch
    from vittracker_amd import config, native
    from vittracker_amd.model import build_ostrack_dist
    c = config.fresh_cfg()
    config.update_config_from_file(os.path.join(REPO, "experiments/vit_dist/vit_48_h32_g128.yaml"), c)
    net = build_ostrack_dist(c, max_batch=2).cuda().eval()
    z, x = torch.zeros(2, 3, 64, 64, device="cuda"), torch.zeros(2, 3, 128, 128, device="cuda")
    g, out = net._native().capture(z, x)
    g.launch()
    torch.cuda.synchronize()
    # a larger-batch forward must not silently free the buffers the live graph replays over
    with pytest.raises(native.VtError, match="captured graph"):
        net.forward(torch.zeros(3, 3, 64, 64, device="cuda"), torch.zeros(3, 3, 128, 128, device="cuda"))
    g.launch()                                   # still valid
    net.reserve(3)                               # explicit re-size: the graph is invalidated ...
    with pytest.raises(native.VtError, match="capture it again"):
        g.launch()                               # ... and says so instead of touching freed memory
    o = net.forward(torch.zeros(3, 3, 64, 64, device="cuda"), torch.zeros(3, 3, 128, 128, device="cuda"))
    assert o["score_map"].shape == (3, 1, 8, 8)
    m = _model()
    g2, _ = m.capture(torch.zeros(4, 3, 64, 64, device="cuda"), torch.zeros(4, 3, 128, 128, device="cuda"))
    m.close()
    with pytest.raises(native.VtError, match="capture it again"):
        g2.launch()


@pytest.mark.parametrize("geom,depth", [("G128", 1), ("G128", 5), ("G128", 12), ("G256", 7), ("G256", 12)])
def test_depth_other_than_three(geom, depth):
    """vt_create accepts depth 1..12: the dynamic-LDS limits follow the depth (the LayerNorm vectors and
    biases of every block live in LDS), and the result still matches the oracle."""
    import torch
    from oracle import vt_oracle_np as onp
    from vittracker_amd import synth
    tz, tx = GEOMS[geom]
    m = _model(geom, B=2, depth=depth)
    sd = synth.synth_state_dict(0, depth=depth, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    z, x = synth.synth_inputs(5, 2, tz, tx)
    out = m.forward(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
    ref = onp.forward(sd, z, x, depth=depth)
    for k in ("score_map", "size_map", "offset_map"):
        np.testing.assert_allclose(getattr(out, k).cpu().numpy(), ref[k], atol=2e-4, rtol=0, err_msg=k)


def test_no_sync_tracking_with_frames_large_enough_to_outlive_the_python_call():
    """track(sync=False) with ~59 MB of frames per step: the H2D copy of step f is still in flight when the
    host stages step f+1.  Two pinned staging buffers with one event each keep frame f intact."""
    import torch
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    os.environ["VITTRACK_PRJ_DIR"] = REPO
    p = P.parameters("vit_48_h32_g128")
    p.allow_synthetic_weights, p.debug = True, 0
    B, n, H, W = 64, 6, 480, 640
    rs = np.random.RandomState(11)
    base = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    vids = [np.roll(base, 7 * f, axis=2) for f in range(n)]            # every frame differs everywhere
    boxes0 = [[200 + (b % 8) * 10, 150 + (b // 8) * 10, 60, 50] for b in range(B)]
    a = BatchedVitTracker(p, B)
    a.initialize(vids[0], boxes0)
    for f in range(1, n):
        ref = a.track(vids[f], sync=True)
    b = BatchedVitTracker(p, B)
    b.initialize(vids[0], boxes0)
    for f in range(1, n):
        last = b.track(vids[f], sync=False)
    np.testing.assert_array_equal(last["target_bbox"].cpu().numpy(), ref["target_bbox"].numpy())
    np.testing.assert_array_equal(last["confidence"].cpu().numpy(), ref["confidence"].numpy())
