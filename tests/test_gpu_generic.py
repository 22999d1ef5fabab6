"""Other geometries of the vit_48_h32 config surface (round 4 review: vt_create accepted exactly three shapes).

build_ostrack_dist builds from whatever DATA.TEMPLATE.SIZE / DATA.SEARCH.SIZE the YAML names (lib/models/vit_dist/vit_dist.py:159-198;
lib/utils/ce_utils.py:22-32 lists template feature sizes 8 / 12 / 7 / 14).  Every stride-16 geometry other than the two tuned ones runs
the shape-generic kernels of vt_generic.h -- checked here through the C ABI against the pinned numpy oracle (oracle/vt_oracle_np.py is
shape-generic and pinned on the reference's own outputs at the two tuned geometries): whole forward, every stage fed the oracle's
upstream activation, the template cache, a graph replay and the first-index decode.  Tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_MAP, TOL_ACT, TOL_BOX = 1e-4, 1e-4, 1e-5
GEOMS = [(112, 224), (192, 384), (96, 160)]        # token counts 245 (not a multiple of 16), 720, 136


def _setup(tz, tx, B, seed):
    import torch
    from vittracker_amd import native, synth
    sd = synth.synth_state_dict(seed, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    z, x = synth.synth_inputs(seed, B, tz, tx)
    m = native.Model(tz, tx, max_batch=B)
    m.load_state_dict(sd)
    return sd, z, x, m, torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()


@pytest.mark.parametrize("tz,tx", GEOMS)
def test_generic_geometry_matches_the_oracle(tz, tx):
    import torch
    from oracle import vt_oracle_np as onp
    B = 3
    sd, z, x, m, zd, xd = _setup(tz, tx, B, seed=21)
    F = tx // 16
    assert (m.len_z, m.len_x, m.feat_sz) == ((tz // 16) ** 2, F * F, F)
    ref = onp.forward(sd, z, x, want_acts=True)
    out = m.forward(zd, xd)
    for k in ("score_map", "size_map", "offset_map"):
        np.testing.assert_allclose(getattr(out, k).cpu().numpy(), ref[k], atol=TOL_MAP, rtol=0, err_msg=k)
    ok = onp.top2_margin(ref["score_map"]) > 1e-3
    np.testing.assert_allclose(out.pred_boxes.cpu().numpy()[ok], ref["pred_boxes"][ok, 0], atol=TOL_BOX)
    okh = onp.top2_margin(ref["score_map"] * onp.hann2d(F)) > 1e-3
    np.testing.assert_allclose(out.hann_boxes.cpu().numpy()[okh], ref["hann_boxes"][okh], atol=TOL_BOX)
    np.testing.assert_allclose(out.conf.cpu().numpy(), ref["conf"], atol=TOL_MAP)
    # the decode is exact on this path's own maps (first-index argmax)
    bbox, mx = m.cal_bbox(out.score_map, out.size_map, out.offset_map)
    assert torch.equal(bbox, out.pred_boxes) and torch.equal(mx, out.conf)
    # stages, each fed the oracle's upstream activation
    acts = ref["acts"]
    tok = m.stem(zd, xd)
    np.testing.assert_allclose(tok.cpu().numpy(), acts["tokens"], atol=TOL_ACT, rtol=0)
    for k in (1, 3):
        _, resid = m.blocks(torch.from_numpy(acts["tokens"]).cuda(), nblocks=k, want_resid=True)
        np.testing.assert_allclose(resid.cpu().numpy(), acts[f"block{k - 1}"], atol=TOL_ACT, rtol=0, err_msg=f"block{k - 1}")
    feat = m.blocks(torch.from_numpy(acts["block2"]).cuda(), nblocks=0)
    np.testing.assert_allclose(feat.cpu().numpy(), acts["norm"][:, -m.len_x:], atol=TOL_ACT, rtol=0)
    ho = m.head(torch.from_numpy(np.ascontiguousarray(acts["norm"][:, -m.len_x:])).cuda())
    for k in ("score_map", "size_map", "offset_map"):
        np.testing.assert_allclose(getattr(ho, k).cpu().numpy(), ref[k], atol=TOL_MAP, rtol=0, err_msg="head " + k)


def test_generic_geometry_template_cache_graph_and_batch_invariance():
    """The rest of the ABI on a generic geometry: vt_set_template + forward(None, x) == forward(z, x) bit for bit, a captured graph
    replays the eager result, and frame i of a batch equals the frame run alone."""
    import torch
    tz, tx, B = 112, 224, 4
    sd, z, x, m, zd, xd = _setup(tz, tx, B, seed=22)
    keys = ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")
    full = m.forward(zd, xd)
    ref = {k: getattr(full, k).clone() for k in keys}
    m.set_template(zd)
    cached = m.forward(None, xd)
    for k in keys:
        assert torch.equal(getattr(cached, k), ref[k]), k
    graph, go = m.capture(zd, xd)
    graph.launch()
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(getattr(go, k), ref[k]), k
    one = m.forward(zd[2:3].contiguous(), xd[2:3].contiguous())
    for k in keys:
        assert torch.equal(getattr(one, k)[0], ref[k][2]), k


def test_sizes_that_are_not_multiples_of_16_are_rejected():
    from vittracker_amd import native
    with pytest.raises(native.VtError, match="multiples of 16"):
        native.Model(100, 200, max_batch=1)
