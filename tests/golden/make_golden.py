#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own model code (imported, never copied).

Run in the build container only (needs /root/reference); the resulting ``*.npz`` are data
(inputs are regenerated from ``vittracker_amd.synth`` by seed, only expected outputs are stored).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
    python tests/golden/make_golden.py --time     # also prints the reference CPU timing loop

How the reference is imported: ``import lib.models`` is impossible here
(its ``__init__`` pulls cv2/timm internals), so the hot-path files are loaded by path under empty
package stubs.  Absent third-party modules are replaced as follows:

* ``easydict.EasyDict``       -> attribute dict (config container only; no arithmetic)
* ``torchvision.ops.boxes``   -> ``box_area`` (import-time only, lib/utils/box_ops.py:2)
* ``timm.models.layers``      -> ``Mlp`` (fc1 -> act -> fc2), ``DropPath`` (identity in eval),
                                 ``trunc_normal_`` / ``lecun_normal_`` (init only)
* ``timm.models.vision_transformer.Block`` -> the reference's OWN in-tree restatement,
  ``lib/models/layers/attn_blocks.py:117-133`` (``Block``) with
  ``lib/models/layers/attn.py:9-59`` (``Attention``).  At the arguments vit_dist uses
  (qkv_bias=True, no drop, nn.LayerNorm, nn.GELU) it is arithmetically the timm block
  (timm is unpinned in install.sh; SURVEY.md 8(c) 'Third-party arithmetic').

So every multiply in the golden outputs is executed by reference source files except the
three-line Mlp container.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import time
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("VT_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from vittracker_amd import synth  # noqa: E402


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []  # mark as package
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def import_reference():
    # --- third-party stand-ins (see module docstring)
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    ed = types.ModuleType("easydict"); ed.EasyDict = EasyDict; sys.modules["easydict"] = ed

    tv = _pkg("torchvision"); tv.__version__ = "0.0-stub"
    tvo = _pkg("torchvision.ops"); tvb = types.ModuleType("torchvision.ops.boxes")
    tvb.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sys.modules["torchvision.ops.boxes"] = tvb; tv.ops = tvo; tvo.boxes = tvb

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class DropPath(nn.Identity):
        def __init__(self, p=0.):
            super().__init__()

    _pkg("timm"); _pkg("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.Mlp, tl.DropPath = Mlp, DropPath
    tl.trunc_normal_ = lambda t, std=.02, **k: nn.init.trunc_normal_(t, std=std)
    tl.lecun_normal_ = lambda t: t
    sys.modules["timm.models.layers"] = tl

    # --- reference package skeleton + the hot-path files, loaded where they lie
    for p in ("lib", "lib.models", "lib.models.layers", "lib.models.vit_dist", "lib.utils",
              "lib.config", "lib.config.vit_dist", "lib.test", "lib.test.utils"):
        _pkg(p)
    _load("lib.models.layers.frozen_bn", "lib/models/layers/frozen_bn.py")
    _load("lib.models.layers.rpe", "lib/models/layers/rpe.py")
    _load("lib.models.layers.attn", "lib/models/layers/attn.py")
    ab = _load("lib.models.layers.attn_blocks", "lib/models/layers/attn_blocks.py")
    vt = types.ModuleType("timm.models.vision_transformer"); vt.Block = ab.Block
    sys.modules["timm.models.vision_transformer"] = vt
    _load("lib.models.layers.head", "lib/models/layers/head.py")
    box_ops = _load("lib.utils.box_ops", "lib/utils/box_ops.py")
    model = _load("lib.models.vit_dist.vit_dist", "lib/models/vit_dist/vit_dist.py")
    config = _load("lib.config.vit_dist.config", "lib/config/vit_dist/config.py")
    hann = _load("lib.test.utils.hann", "lib/test/utils/hann.py")
    return model, config, box_ops, hann


def import_reference_sample_target():
    """lib/train/data/processing_utils.py loaded where it lies, with `cv2` replaced by a two-function
    stand-in: ``copyMakeBorder(BORDER_CONSTANT)`` = constant zero ``np.pad`` (what OpenCV's constant
    border does with the default value), ``resize`` raising (never reached on the ``output_sz=None``
    return at :76).  Everything the fixtures pin -- crop side, banker's-rounded origin, the pad
    formula with its ``+ 1`` quirk, the slice, the attention mask -- is executed by reference source."""
    cv = types.ModuleType("cv2")
    cv.BORDER_CONSTANT = 0

    def copyMakeBorder(src, top, bottom, left, right, borderType, value=0):
        assert borderType == cv.BORDER_CONSTANT
        return np.pad(src, ((top, bottom), (left, right)) + ((0, 0),) * (src.ndim - 2), mode="constant")

    def resize(*a, **k):
        raise RuntimeError("cv2.resize is not available in this image: the resize step stays unpinned")

    cv.copyMakeBorder, cv.resize = copyMakeBorder, resize
    sys.modules["cv2"] = cv
    for p in ("lib", "lib.train", "lib.train.data"):
        if p not in sys.modules:
            _pkg(p)
    return _load("lib.train.data.processing_utils", "lib/train/data/processing_utils.py")


CROP_IMAGE_HW = (36, 48)
CROP_CASES = [  # (box [x, y, w, h], search_area_factor): inside, across each border and corner, fractional
    ([14, 10, 8, 6], 2.0), ([-3, -2, 9, 9], 2.0), ([40, 28, 10, 9], 2.0), ([0, 0, 4, 4], 4.0), ([20.5, 13.5, 5, 3], 2.0),
    ([44, 2, 6, 6], 2.0), ([2, 30, 7, 5], 2.0), ([3.25, 7.75, 2.5, 1.25], 4.0), ([10, 8, 30, 24], 1.5),
    ([23.5, 17.5, 4, 4], 2.0), ([24.5, 18.5, 4, 4], 2.0),            # x.5 origins: round-half-even both ways
    ([21, 15, 3, 3], 2.0), ([22, 16, 3, 3], 2.0), ([47, 35, 2, 2], 4.0), ([-6, 12, 5, 8], 3.0), ([12, -7, 8, 5], 3.0),
    ([30, 20, 17.9, 15.9], 2.0), ([16, 12, 16, 12], 1.0), ([5.5, 4.5, 9, 9], 1.0), ([0, 0, 48, 36], 1.0),
    ([11, 9, 1, 1], 1.0), ([11.5, 9.5, 1, 1], 2.0), ([35, 25, 12.5, 12.5], 2.0), ([1, 1, 0.6, 0.6], 4.0),
]


def make_crop_fixture(out_dir):
    pu = import_reference_sample_target()
    rs = np.random.RandomState(2024)
    im = rs.randint(0, 256, CROP_IMAGE_HW + (3,)).astype(np.uint8)
    res = {"image_seed": 2024, "image_hw": np.array(CROP_IMAGE_HW), "n": len(CROP_CASES),
           "boxes": np.array([b for b, _ in CROP_CASES], dtype=np.float64), "factors": np.array([f for _, f in CROP_CASES])}
    for i, (bb, f) in enumerate(CROP_CASES):
        crop, mask, one = pu.sample_target(im, list(map(float, bb)), f, output_sz=None)     # (sic) order of :76
        assert one == 1.0 and crop.shape[0] == crop.shape[1] == mask.shape[0]
        res[f"crop_{i}"], res[f"mask_{i}"] = crop, mask
    try:
        pu.sample_target(im, [5.0, 5.0, 0.0, 0.0], 4.0, output_sz=None)
        raise AssertionError("expected 'Too small bounding box.'")
    except Exception as e:  # noqa: BLE001
        res["too_small_message"] = str(e)
    np.savez_compressed(os.path.join(out_dir, "ref_crop_geometry.npz"), **res)
    print(f"ref_crop_geometry.npz: {len(CROP_CASES)} crops, sides {sorted({res[f'crop_{i}'].shape[0] for i in range(len(CROP_CASES))})}")


def build_reference_model(model_mod, config_mod, geom: str):
    """G256: the shipped YAML as is.  G128: DATA.SEARCH.SIZE=128 so the head gets feat_sz=8
    (lib/models/layers/head.py:356) and the hard-coded 64/256-token pos-embeds
    (lib/models/vit_dist/vit_dist.py:61-62) replaced by 16/64-token ones."""
    cfg = config_mod.cfg
    config_mod.update_config_from_file(os.path.join(REF, "experiments/vit_dist/vit_48_h32_noKD.yaml"))
    if geom == "G128":
        cfg.DATA.SEARCH.SIZE = 128
        cfg.DATA.TEMPLATE.SIZE = 64
    else:
        cfg.DATA.SEARCH.SIZE = 256
        cfg.DATA.TEMPLATE.SIZE = 128
    net = model_mod.build_ostrack_dist(cfg)  # depth=3, mode='eval' defaults (tracker vit_dist.py:24)
    if geom == "G128":
        C = cfg.MODEL.BACKBONE.CHANNELS
        net.pos_embed_z = nn.Parameter(torch.zeros(1, 16, C))
        net.pos_embed_x = nn.Parameter(torch.zeros(1, 64, C))
    return net.eval()


GEOMS = {"G256": (128, 256), "G128": (64, 128)}
CASES = [  # (geom, seed, B, with_acts)
    ("G256", 0, 4, False),
    ("G256", 1, 1, True),
    ("G128", 0, 4, False),
    ("G128", 2, 1, True),
    ("G128", 3, 8, False),
]


def run_case(model_mod, config_mod, hann_mod, geom, seed, B, with_acts):
    tz, tx = GEOMS[geom]
    net = build_reference_model(model_mod, config_mod, geom)
    sd = synth.synth_state_dict(seed, C=48, depth=3, head_ch=32, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    z, x = synth.synth_inputs(seed, B, tz, tx)

    acts = {}
    hooks = []
    if with_acts:
        def grab(name):
            def h(_m, _i, o):
                acts.setdefault(name, []).append(o.detach().clone().numpy())
            return h
        for i in (0, 2, 4, 6):
            # conv+BN output; the Hardswish modules (1,3,5) give the post-activation value
            hooks.append(net.patch_embed.net[i if i == 6 else i + 1].register_forward_hook(grab(f"stem{i // 2}")))
        for i, blk in enumerate(net.blocks):
            hooks.append(blk.register_forward_hook(grab(f"block{i}")))
        hooks.append(net.norm.register_forward_hook(grab("norm")))
        for t in ("ctr", "offset", "size"):
            for i in range(1, 5):
                hooks.append(getattr(net.box_head, f"conv{i}_{t}").register_forward_hook(grab(f"head_{t}{i}")))

    with torch.no_grad():
        out = net(torch.from_numpy(z), torch.from_numpy(x))
        F = net.box_head.feat_sz
        win = hann_mod.hann2d(torch.tensor([F, F]).long(), centered=True)       # tracker vit_dist.py:34
        response = win * out["score_map"]                                         # tracker vit_dist.py:104
        hbox = net.box_head.cal_bbox(response, out["size_map"], out["offset_map"])  # :105
        conf = out["score_map"].flatten(1).max(dim=1).values                      # :148 (per sample)
    for h in hooks:
        h.remove()

    res = {
        "geom": geom, "seed": seed, "B": B, "state_checksum": synth.state_checksum(sd),
        "score_map": out["score_map"].numpy(), "size_map": out["size_map"].numpy(),
        "offset_map": out["offset_map"].numpy(), "pred_boxes": out["pred_boxes"].numpy(),
        "hann_boxes": hbox.numpy(), "conf": conf.numpy(), "hann_window": win.numpy(),
    }
    if with_acts:
        # stem hooks fire twice: first for z (vit_dist.py:78), then x (:79)
        for i in range(4):
            res[f"act_stem{i}_z"], res[f"act_stem{i}_x"] = acts[f"stem{i}"][0], acts[f"stem{i}"][1]
        for k, v in acts.items():
            if not k.startswith("stem"):
                res["act_" + k] = v[0]
    return res, net, (z, x)


def time_reference(net, z, x, warm=500, n=1000):
    """The exact loop of tracking/profile_model_cpu.py:36-49 (bs=1, 500 warm-up + 1000 timed)."""
    zt, xt = torch.from_numpy(z[:1]), torch.from_numpy(x[:1])
    with torch.no_grad():
        for _ in range(warm):
            net(zt, xt)
        t0 = time.time()
        for _ in range(n):
            net(zt, xt)
        dt = (time.time() - t0) / n
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only-crop", action="store_true", help="regenerate ref_crop_geometry.npz only")
    args = ap.parse_args()
    if args.only_crop:
        make_crop_fixture(args.out)
        return
    torch.manual_seed(0)
    model_mod, config_mod, box_ops, hann_mod = import_reference()
    for geom, seed, B, with_acts in CASES:
        res, net, (z, x) = run_case(model_mod, config_mod, hann_mod, geom, seed, B, with_acts)
        name = f"ref_{geom}_s{seed}_b{B}.npz"
        np.savez_compressed(os.path.join(args.out, name), **res)
        sm = res["score_map"].reshape(B, -1)
        srt = np.sort(sm, axis=1)
        print(f"{name}: score range [{sm.min():.4f}, {sm.max():.4f}] top2 margin min {np.min(srt[:, -1] - srt[:, -2]):.2e}"
              f" pred_boxes[0]={res['pred_boxes'][0, 0]}")
        if args.time:
            dt = time_reference(net, z, x)
            print(f"  reference CPU loop ({torch.get_num_threads()} threads): {dt * 1e3:.2f} ms/frame = {1 / dt:.1f} fps")

    # tracker-tail known answers from reference functions that import cleanly
    rs = np.random.RandomState(7)
    boxes = rs.uniform(-50, 700, (64, 4)).tolist()
    clipped = [box_ops.clip_box(b, 480, 640, margin=10) for b in boxes]
    np.savez_compressed(os.path.join(args.out, "ref_clip_box.npz"), boxes=np.array(boxes), H=480, W=640,
                        margin=10, clipped=np.array(clipped))
    hw = {f"hann{n}": hann_mod.hann2d(torch.tensor([n, n]).long(), centered=True).numpy() for n in (8, 16, 20)}
    np.savez_compressed(os.path.join(args.out, "ref_hann.npz"), **hw)
    print("wrote clip_box / hann fixtures")
    make_crop_fixture(args.out)


if __name__ == "__main__":
    main()
