"""Host-side logic on CPU: config surface, parameter file, crop geometry, box mapping, window,
sharding + gather (gloo, world_size 2), and the fail-loudly contract."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, REPO
from oracle import vt_oracle_np as onp


# ---------------------------------------------------------------- config (lib/config/vit_dist/config.py)
def test_yaml_merge_and_strict_keys(tmp_path):
    from vittracker_amd import config
    c = config.fresh_cfg()
    assert (c.MODEL.BACKBONE.CHANNELS, c.MODEL.BACKBONE.HEADS, c.TEST.SEARCH_SIZE) == (768, 12, 320)   # defaults
    config.update_config_from_file(os.path.join(REPO, "experiments/vit_dist/vit_48_h32_noKD.yaml"), c)
    g = config.geometry(c)
    assert (g["channels"], g["heads"], g["head_channels"]) == (48, 1, 32)          # "h32" = head width, 1 attention head
    assert (g["template_size"], g["search_size"], g["feat_sz"], g["len_z"], g["len_x"]) == (128, 256, 16, 64, 256)
    assert c.TRAIN.LR == 0.0004 and c.TRAIN.AUX_WEIGHT == 1.0                      # set by the YAML / untouched default survives
    bad = tmp_path / "bad.yaml"
    bad.write_text("MODEL:\n  BACKBONE:\n    NOT_A_KEY: 1\n")
    with pytest.raises(ValueError, match="NOT_A_KEY not exist in config.py"):
        config.update_config_from_file(str(bad), c)
    c2 = config.fresh_cfg()
    config.update_config_from_file(os.path.join(REPO, "experiments/vit_dist/vit_48_h32_g128.yaml"), c2)
    assert config.geometry(c2)["len_z"] + config.geometry(c2)["len_x"] == 80


def test_parameters_file(monkeypatch):
    from vittracker_amd.parameter import vit_dist as P
    monkeypatch.setenv("VITTRACK_PRJ_DIR", REPO)
    monkeypatch.setenv("VITTRACK_SAVE_DIR", "/ckpt")
    p = P.parameters("vit_48_h32_noKD")
    assert (p.template_factor, p.template_size, p.search_factor, p.search_size) == (2.0, 128, 4.0, 256)
    assert p.checkpoint == "/ckpt/checkpoints/train/vit_dist/vit_48_h32_noKD/OstrackDist_ep0300.pth.tar"
    assert p.save_all_boxes is False and p.cfg.MODEL.HEAD.TYPE == "CENTER"


# ---------------------------------------------------------------- window / boxes
def test_hann_window_matches_reference_values():
    import torch
    from vittracker_amd.host_ops import hann2d
    h = np.load(os.path.join(GOLDEN_DIR, "ref_hann.npz"))
    for n in (8, 16, 20):
        np.testing.assert_array_equal(hann2d(torch.tensor([n, n])).numpy(), h[f"hann{n}"])


def test_clip_box_known_answers():
    from vittracker_amd.host_ops import clip_box
    g = np.load(os.path.join(GOLDEN_DIR, "ref_clip_box.npz"))
    for b, c in zip(g["boxes"].tolist(), g["clipped"].tolist()):
        assert clip_box(b, int(g["H"]), int(g["W"]), int(g["margin"])) == c


# ---------------------------------------------------------------- crop geometry (processing_utils.py:12-79)
def _brute_crop(im, bb, factor):
    """Independent per-pixel statement of the crop + zero pad (no resize)."""
    x, y, w, h = bb
    s = math.ceil(math.sqrt(w * h) * factor)
    x1 = round(x + 0.5 * w - s * 0.5)
    y1 = round(y + 0.5 * h - s * 0.5)
    H, W = im.shape[:2]
    # the reference's pad formula keeps one pixel less at the right/bottom edge (x2 - W + 1)
    out = np.zeros((s, s, 3), np.uint8)
    mask = np.ones((s, s), bool)
    for j in range(s):
        for i in range(s):
            yy, xx = y1 + j, x1 + i
            if 0 <= yy < H - (1 if y1 + s >= H else 0) and 0 <= xx < W - (1 if x1 + s >= W else 0):
                out[j, i] = im[yy, xx]
                mask[j, i] = False
    return out, mask


@pytest.mark.parametrize("bb", [[40, 30, 20, 24], [-5, -8, 30, 30], [100, 70, 40, 36], [0, 0, 8, 8], [60.5, 41.5, 11, 7]])
def test_sample_target_crop_and_pad(bb):
    from vittracker_amd.host_ops import sample_target
    rs = np.random.RandomState(0)
    im = rs.randint(0, 256, (96, 128, 3)).astype(np.uint8)
    crop, mask, factor = sample_target(im, bb, 2.0, output_sz=None)
    ref, rmask = _brute_crop(im, bb, 2.0)
    assert factor == 1.0
    np.testing.assert_array_equal(crop, ref)
    np.testing.assert_array_equal(mask, rmask)


def test_sample_target_geometry_matches_reference_fixture():
    """Crop side, banker's-rounded origin, pad formula (with the reference's `+ 1` quirk), slice and
    attention mask against fixtures produced by the reference's own sample_target
    (tests/golden/make_golden.py::make_crop_fixture; the `output_sz=None` return, processing_utils.py:76)."""
    from vittracker_amd.host_ops import sample_target
    g = np.load(os.path.join(GOLDEN_DIR, "ref_crop_geometry.npz"))
    H, W = g["image_hw"]
    im = np.random.RandomState(int(g["image_seed"])).randint(0, 256, (int(H), int(W), 3)).astype(np.uint8)
    for i in range(int(g["n"])):
        crop, mask, one = sample_target(im, g["boxes"][i].tolist(), float(g["factors"][i]), output_sz=None)
        np.testing.assert_array_equal(crop, g[f"crop_{i}"], err_msg=f"case {i} box {g['boxes'][i]}")
        np.testing.assert_array_equal(mask, g[f"mask_{i}"], err_msg=f"case {i}")
        assert one == 1.0
    with pytest.raises(Exception, match=str(g["too_small_message"]).rstrip(".")):
        sample_target(im, [5.0, 5.0, 0.0, 0.0], 4.0, output_sz=None)


def test_resize_port_equals_cv2_when_cv2_is_available():
    """The uint8 bilinear port (host_ops.resize_bilinear_u8, mirrored by vt_crop) against the real
    cv2.resize -- runs only on a box that has OpenCV (this image does not: the resize stays unpinned)."""
    cv2 = pytest.importorskip("cv2")
    from vittracker_amd.host_ops import resize_bilinear_u8
    rs = np.random.RandomState(5)
    for (h, w), T in (((37, 53), 64), ((200, 200), 128), ((97, 97), 256), ((300, 300), 128), ((64, 64), 64)):
        im = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        np.testing.assert_array_equal(resize_bilinear_u8(im, T, T), cv2.resize(im, (T, T)))


def test_sample_target_resize_properties():
    """cv2 is absent here, so the bilinear port is checked through properties cv.resize has:
    identity at equal size, constants stay constant, exact 2x upsample of a horizontal ramp stays
    monotone, resize_factor = output / crop."""
    from vittracker_amd.host_ops import resize_bilinear_u8, sample_target
    rs = np.random.RandomState(1)
    im = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    np.testing.assert_array_equal(resize_bilinear_u8(im, 64, 64), im)
    const = np.full((37, 53, 3), 113, np.uint8)
    assert (resize_bilinear_u8(const, 128, 128) == 113).all()
    ramp = np.tile(np.arange(64, dtype=np.uint8)[None, :, None] * 4, (8, 1, 3))
    up = resize_bilinear_u8(ramp, 8, 128).astype(int)
    assert (np.diff(up[0, :, 0]) >= 0).all() and up[0, 0, 0] == 0 and up[0, -1, 0] == 252
    crop, rf, mask = sample_target(im, [20, 20, 10, 10], 4.0, output_sz=128)
    assert crop.shape == (128, 128, 3) and crop.dtype == np.uint8 and mask.shape == (128, 128)
    assert rf == 128 / math.ceil(math.sqrt(100) * 4.0)
    with pytest.raises(Exception, match="Too small bounding box"):
        sample_target(im, [5, 5, 0, 0], 4.0, output_sz=128)


def test_map_box_back_matches_oracle():
    from vittracker_amd.tracker.vit_dist import Vit_dist
    class P: search_size = 256
    t = Vit_dist.__new__(Vit_dist)
    t.params, t.state = P, [100.0, 50.0, 40.0, 30.0]
    got = t.map_box_back([130.0, 120.0, 44.0, 28.0], 256 / 140)
    assert got == pytest.approx(onp.map_box_back(t.state, [130.0, 120.0, 44.0, 28.0], 256 / 140, 256))


# ---------------------------------------------------------------- fail loudly without the GPU / extension
def test_no_cpu_execution_path():
    import torch
    from vittracker_amd import config, native
    from vittracker_amd.model import build_ostrack_dist
    c = config.fresh_cfg()
    config.update_config_from_file(os.path.join(REPO, "experiments/vit_dist/vit_48_h32_noKD.yaml"), c)
    net = build_ostrack_dist(c)
    sd = net.state_dict()
    assert "blocks.2.mlp.fc2.bias" in sd and "box_head.conv5_size.bias" in sd
    r = net.load_state_dict({**sd, "convs.0.weight": torch.zeros(768, 48, 1)}, strict=False)
    assert r.unexpected_keys == ["convs.0.weight"] and not r.missing_keys       # training-only keys are ignored
    with pytest.raises(native.VtError, match="no CPU"):
        net.forward(torch.zeros(1, 3, 128, 128), torch.zeros(1, 3, 256, 256))
    with pytest.raises(native.VtError):
        net.to("cpu")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vittracker_amd/ may reference it."""
    import re
    for root, _, files in os.walk(os.path.join(REPO, "vittracker_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


# ---------------------------------------------------------------- sharding + gather (gloo, 2 ranks)
def test_shard_rule():
    from vittracker_amd.parallel import merge_rank_results, shard_sequences, shard_sizes
    assert shard_sequences(10, 1, 4) == [1, 5, 9]
    assert shard_sizes(10, 4) == [3, 3, 2, 2]
    per_rank = [[f"s{s}" for s in shard_sequences(7, r, 3)] for r in range(3)]
    assert merge_rank_results(per_rank, 7) == [f"s{i}" for i in range(7)]


_GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VT_REPO"])
from vittracker_amd.parallel import ResultGather, shard_sequences
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 7                                   # ragged: ranks get 4 and 3 sequences
mine = shard_sequences(N, rank, world)
g = ResultGather(N)
for step in range(5):                   # records encode (sequence, step) so any mix-up shows
    local = torch.tensor([[s, step, s * 10 + step, rank, 1.0] for s in mine], dtype=torch.float32)
    g.submit(step, local)
    if step >= 1:                       # collect one step late: the gather overlaps the next step
        got = g.collect(step - 1)
        want = torch.tensor([[s, step - 1, s * 10 + step - 1, s % world, 1.0] for s in range(N)])
        assert torch.equal(got, want), (rank, step, got)
got = g.collect(4)
assert got[:, 0].tolist() == list(range(N))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_result_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, VT_REPO=REPO, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


# ---------------------------------------------------------------- bench.py launcher (no GPU: gloo, compute skipped)
@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launches_ranks_plumbing_only(world):
    """`python bench.py --gpus N` with no torch.distributed.run wrapper: the parent spawns the ranks as a
    child, relays ONE JSON line and exits with the child's code (here on gloo with compute skipped).  N = 8 is the driver's
    SCALE configuration (BASELINE config 3)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--plumbing-only", "--steps", "7",
                        "--steps-per-graph", "3", "--warmup", "1", "--batch", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["config"]["global_batch"] == 8 * world and len(j["per_rank_frames_per_s"]) == world
    assert "gather_exposed_us_per_step" in j and j["scaling"] == "weak"
    assert j["steps"] == 7 and j["config"]["steps_per_graph"] == 3      # 2 launch units of 3 steps + 1 single step


def test_bench_refuses_diagnostic_switches():
    env = dict(os.environ, VT_SKIP_HEAD="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--plumbing-only", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refuses to run" in (r.stdout + r.stderr)


def test_build_ostrack_config_surface():
    from vittracker_amd import config
    from vittracker_amd.model_vitb import build_ostrack
    c = config.fresh_cfg()
    config.update_config_from_file(os.path.join(REPO, "experiments/ostrack/vitb_256.yaml"), c)
    net = build_ostrack(c, training=False)
    sd = net.state_dict()
    assert sd["backbone.blocks.11.mlp.fc2.weight"].shape == (768, 3072) and sd["box_head.conv1_ctr.0.weight"].shape == (256, 768, 3, 3)
    assert sd["backbone.pos_embed_z"].shape == (1, 64, 768) and net.box_head.feat_sz == 16
    c.MODEL.BACKBONE.TYPE = "vit_base_patch16_224_ce"
    with pytest.raises(NotImplementedError):
        build_ostrack(c, training=False)


def test_sharded_tracker_group_sizes():
    """ShardedBatchedTracker.shard_sizes: contiguous near-equal groups, none empty, order preserved (sequence b stays sequence b)."""
    from vittracker_amd.batched import ShardedBatchedTracker as S
    assert S.shard_sizes(256, 2) == [128, 128]
    assert S.shard_sizes(10, 3) == [4, 3, 3]
    assert S.shard_sizes(5, 1) == [5]
    assert S.shard_sizes(2, 4) == [1, 1]              # never more groups than sequences
    assert S.shard_sizes(1, 2) == [1]
    for b in range(1, 40):
        for n in range(1, 6):
            sz = S.shard_sizes(b, n)
            assert sum(sz) == b and min(sz) >= 1 and max(sz) - min(sz) <= 1 and sz == sorted(sz, reverse=True)

