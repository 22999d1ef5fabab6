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


def test_other_widths_match_the_reference():
    """Round 6: CHANNELS / HEADS / HEAD.NUM_CHANNELS other than the shipped 48 / 1 / 32 (build_ostrack_dist takes all three from the YAML,
    lib/models/vit_dist/vit_dist.py:159-164) run the shape-generic kernels at run-time widths: the REFERENCE model's outputs at 64 / 2 / 64
    (G256) and 32 / 4 / 16 (G128) -- tests/golden/make_golden_cfg.py -- within the fp32 path's tolerances; template cache, graph replay and
    the uint8-patch entry included."""
    import torch
    from conftest import cfg_golden_files, load_cfg_case
    from vittracker_amd import native, synth
    files = cfg_golden_files()
    assert len(files) == 2
    for path in files:
        g, sd, z, x, (C, heads, W) = load_cfg_case(path)
        B, tz, tx = z.shape[0], z.shape[2], x.shape[2]
        m = native.Model(tz, tx, channels=C, heads=heads, head_channels=W, max_batch=B)
        assert m.channels == C
        m.load_state_dict(sd)
        zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
        out = m.forward(zd, xd)
        for k in ("score_map", "size_map", "offset_map"):
            err = float(np.abs(getattr(out, k).cpu().numpy() - g[k]).max())
            assert err < 1e-4, (k, C, heads, W, err)
        for k in ("pred_boxes", "hann_boxes"):
            got = getattr(out, k).cpu().numpy()
            assert float(np.abs(got - g[k].reshape(got.shape)).max()) < 1e-5, (k, C, heads, W)
        m.set_template(zd)
        cached = m.forward(None, xd)
        graph, gout = m.capture(None, xd)
        graph.launch()
        torch.cuda.synchronize()
        for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf"):
            assert torch.equal(getattr(cached, k), getattr(out, k)) and torch.equal(getattr(gout, k), getattr(out, k)), k
        patches = torch.from_numpy(synth.synth_patches(3, B, tx)).cuda()
        xn = torch.from_numpy(synth.normalise_patches(patches.cpu().numpy())).cuda()
        assert torch.equal(m.forward_u8(zd, patches).score_map, m.forward(zd, xn).score_map)
        m.close()


def test_model_level_surface_builds_other_widths_from_the_yaml_fields():
    """build_ostrack_dist(cfg) with MODEL.BACKBONE.CHANNELS / HEADS and MODEL.HEAD.NUM_CHANNELS changed, load_state_dict, forward(z=, x=):
    the reference's model-level calls (lib/test/tracker/vit_dist.py:24-28, :95-98) at 64 / 2 / 64 against the reference's outputs."""
    import os
    import torch
    from conftest import REPO, cfg_golden_files, load_cfg_case
    from vittracker_amd import config
    from vittracker_amd.model import build_ostrack_dist
    path = [p for p in cfg_golden_files() if "c64h2w64" in p][0]
    g, sd, z, x, (C, heads, W) = load_cfg_case(path)
    c = config.fresh_cfg()
    config.update_config_from_file(os.path.join(REPO, "experiments/vit_dist/vit_48_h32_noKD.yaml"), c)
    c.MODEL.BACKBONE.CHANNELS, c.MODEL.BACKBONE.HEADS, c.MODEL.HEAD.NUM_CHANNELS = C, heads, W
    net = build_ostrack_dist(c, max_batch=z.shape[0])
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected
    net = net.cuda().eval()
    out = net.forward(z=torch.from_numpy(z).cuda(), x=torch.from_numpy(x).cuda())
    for k in ("score_map", "size_map", "offset_map"):
        assert float(np.abs(out[k].cpu().numpy() - g[k]).max()) < 1e-4, k
    assert float(np.abs(out["pred_boxes"].cpu().numpy().reshape(-1, 4) - g["pred_boxes"].reshape(-1, 4)).max()) < 1e-5


def test_unsupported_widths_are_rejected_with_a_message():
    from vittracker_amd import native
    for kw in (dict(channels=50), dict(channels=48, heads=5), dict(head_channels=20), dict(channels=2048, heads=1)):
        with pytest.raises(native.VtError, match="unsupported model"):
            native.Model(64, 128, max_batch=1, **kw)
