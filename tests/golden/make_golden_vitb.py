#!/usr/bin/env python3
"""Golden vectors for the ViT-Base OSTrack path (BASELINE config 4), produced by the REFERENCE's own code.

    python tests/golden/make_golden_vitb.py          # writes tests/golden/ref_vitb_*.npz   (build container only)

What runs: ``lib/models/ostrack/ostrack.py::build_ostrack(cfg, training=False)`` with
``MODEL.BACKBONE.TYPE = 'vit_base_patch16_224'``, ``DATA.SEARCH.SIZE = 256``, ``DATA.TEMPLATE.SIZE = 128``,
``MODEL.HEAD = CENTER / 256 channels`` -- i.e. the reference's ``OSTrack`` module, its ``VisionTransformer``
(``lib/models/ostrack/vit.py``), ``BaseBackbone.finetune_track / forward_features``
(``lib/models/ostrack/base_backbone.py``), ``PatchEmbed`` (``lib/models/layers/patch_embed.py``), ``combine_tokens``
(``lib/models/ostrack/utils.py``) and ``CenterPredictor`` (``lib/models/layers/head.py``), all loaded from
/root/reference where they lie.  Stand-ins for what this image lacks (none of them on the arithmetic path except Mlp):

* ``timm.data`` constants, ``timm.models.helpers`` (``build_model_with_cfg`` etc.: only referenced by the
  pretrained-weights loaders, never called here), ``timm.models.registry.register_model`` (identity decorator),
  ``timm.models.vision_transformer.resize_pos_embed`` (unused on this path)
* ``timm.models.layers``: ``Mlp`` (fc1 -> GELU -> fc2), ``DropPath`` (identity: eval mode), ``to_2tuple``,
  ``trunc_normal_`` / ``lecun_normal_`` (initialisers; every parameter is overwritten by the seeded state dict)
* the research add-ons ``ostrack.py`` imports at module level but that ``MODEL.PROCESS.* = 'None'`` never touches:
  ``vit_ce``, ``draw``, ``embedding``, ``preprocess``, ``clipvit`` -> empty modules with the imported names
* ``easydict`` / ``torchvision.ops.boxes`` as in make_golden.py

Weights / inputs are regenerated from seeds (``vittracker_amd.synth.synth_vitb_state_dict``); only outputs and a
row-subsampled set of activations are stored."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (EasyDict / torchvision / timm.models.layers stand-ins, _pkg, _load)

from vittracker_amd import synth  # noqa: E402

ACT_ROWS = sorted(set(list(range(0, 320, 16)) + [63, 64, 65, 319]))      # token rows kept per activation


def import_reference_ostrack():
    mg.import_reference()            # easydict, torchvision stub, timm.models.layers (Mlp, DropPath, inits), lib.* skeleton, head.py
    tl = sys.modules["timm.models.layers"]
    tl.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    td = types.ModuleType("timm.data")
    td.IMAGENET_DEFAULT_MEAN = td.IMAGENET_INCEPTION_MEAN = (0.485, 0.456, 0.406)
    td.IMAGENET_DEFAULT_STD = td.IMAGENET_INCEPTION_STD = (0.229, 0.224, 0.225)
    sys.modules["timm.data"] = td

    def _never(*a, **k):
        raise RuntimeError("pretrained-weights helper of timm: not reachable on the golden-vector path")
    th = types.ModuleType("timm.models.helpers")
    th.build_model_with_cfg = th.adapt_input_conv = _never
    th.named_apply = lambda fn, module, **k: module
    sys.modules["timm.models.helpers"] = th
    tr = types.ModuleType("timm.models.registry")
    tr.register_model = lambda f: f
    sys.modules["timm.models.registry"] = tr
    sys.modules["timm.models.vision_transformer"].resize_pos_embed = _never
    for p in ("lib.models.ostrack", "lib.config.ostrack"):
        mg._pkg(p)
    for name, attrs in (("vit_ce", ["vit_large_patch16_224_ce", "vit_base_patch16_224_ce"]),
                        ("draw", ["Draw", "Color", "DrawMask", "ExtraTemplateMask"]),
                        ("embedding", ["Embedding", "SearchEmbedding"]), ("preprocess", ["build_preprocess"]),
                        ("clipvit", ["clipvittracking_base_patch16"])):
        m = types.ModuleType("lib.models.ostrack." + name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules["lib.models.ostrack." + name] = m
    mg._load("lib.models.layers.patch_embed", "lib/models/layers/patch_embed.py")
    mg._load("lib.models.ostrack.utils", "lib/models/ostrack/utils.py")
    mg._load("lib.models.ostrack.base_backbone", "lib/models/ostrack/base_backbone.py")
    mg._load("lib.models.ostrack.vit", "lib/models/ostrack/vit.py")
    ostrack = mg._load("lib.models.ostrack.ostrack", "lib/models/ostrack/ostrack.py")
    config = mg._load("lib.config.ostrack.config", "lib/config/ostrack/config.py")
    hann = sys.modules["lib.test.utils.hann"]
    return ostrack, config, hann


def build_reference_vitb(ostrack, config):
    cfg = config.cfg
    cfg.MODEL.BACKBONE.TYPE = "vit_base_patch16_224"
    cfg.MODEL.HEAD.TYPE, cfg.MODEL.HEAD.NUM_CHANNELS = "CENTER", 256
    cfg.DATA.SEARCH.SIZE, cfg.DATA.TEMPLATE.SIZE = 256, 128
    cfg.TEST.SEARCH_SIZE, cfg.TEST.TEMPLATE_SIZE = 256, 128
    net = ostrack.build_ostrack(cfg, training=False)
    return net.eval()


def run_case(ostrack, config, hann_mod, seed, B, with_acts, common_mode=0.0):
    net = build_reference_vitb(ostrack, config)
    sd = synth.synth_vitb_state_dict(seed, common_mode=common_mode)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    z, x = synth.synth_inputs(seed, B, 128, 256)
    acts, hooks = {}, []
    if with_acts:
        def grab(name, inp=False):
            def h(_m, i, o):
                acts[name] = (i[0] if inp else o).detach().clone().numpy()
            return h
        hooks.append(net.backbone.blocks[0].register_forward_hook(grab("tokens", inp=True)))
        for i, blk in enumerate(net.backbone.blocks):
            hooks.append(blk.register_forward_hook(grab(f"block{i}")))
        hooks.append(net.backbone.norm.register_forward_hook(grab("norm")))
        for t in ("ctr", "offset", "size"):
            for i in range(1, 5):
                hooks.append(getattr(net.box_head, f"conv{i}_{t}").register_forward_hook(grab(f"head_{t}{i}")))
    with torch.no_grad():
        out = net(template=torch.from_numpy(z), search=torch.from_numpy(x))
        F = net.box_head.feat_sz
        win = hann_mod.hann2d(torch.tensor([F, F]).long(), centered=True)
        hbox = net.box_head.cal_bbox(win * out["score_map"], out["size_map"], out["offset_map"])
        conf = out["score_map"].flatten(1).max(dim=1).values
    for h in hooks:
        h.remove()
    top2 = lambda m: (lambda srt: srt[:, -1] - srt[:, -2])(np.sort(m.reshape(B, -1), axis=1))  # noqa: E731
    res = {"model": "vitb", "seed": seed, "B": B, "common_mode": float(common_mode), "margin_raw": top2(out["score_map"].numpy()),
           "margin_hann": top2((win * out["score_map"]).numpy()), "state_checksum": synth.state_checksum(sd), "act_rows": np.array(ACT_ROWS),
           "score_map": out["score_map"].numpy(), "size_map": out["size_map"].numpy(), "offset_map": out["offset_map"].numpy(),
           "pred_boxes": out["pred_boxes"].numpy(), "hann_boxes": hbox.numpy(), "conf": conf.numpy()}
    for k, v in acts.items():
        if k.startswith("head_"):
            res["act_" + k] = v[:1, ::8].astype(np.float32)          # sample 0, every 8th channel, full 16x16 map
        else:
            res["act_" + k] = v[:1, ACT_ROWS].astype(np.float32)     # sample 0, selected token rows, all 768 channels
    return res


def main():
    torch.manual_seed(0)
    ostrack, config, hann_mod = import_reference_ostrack()
    # bf16 kernels are compared with these fp32 outputs: a fixture is only useful for the bbox comparison when
    # its argmax margins sit well above bf16 noise, so seeds are searched for batches whose margins all exceed 0.03
    # (batch, activations, common-mode offset of the token rows in units of their sigma): the last two are the LayerNorm-fold hazard
    # cases (round 6) -- every token row rides on an offset of 2 / 6 sigma through all twelve blocks
    want = [(2, True, 0.0), (3, False, 0.0), (2, True, 2.0), (2, False, 6.0)]
    if "--only-common-mode" in sys.argv:
        want = [w for w in want if w[2] != 0.0]
    seed = 0
    while want:
        B, with_acts, cm = want[0]
        if cm != 0.0 and seed < 40:
            seed = 40          # their own seed range: the plain fixtures keep the seeds (and files) they have
        res = run_case(ostrack, config, hann_mod, seed, B, with_acts, cm)
        ok = min(res["margin_raw"].min(), res["margin_hann"].min()) > 0.03
        print(f"seed {seed} B {B}: margins raw {np.round(res['margin_raw'], 4)} hann {np.round(res['margin_hann'], 4)} -> {'keep' if ok else 'skip'}")
        seed += 1
        if not ok:
            continue
        want.pop(0)
        seed_used = seed - 1
        name = f"ref_vitb_s{seed_used}_b{B}.npz" if cm == 0.0 else f"ref_vitb_cm{int(cm)}_s{seed_used}_b{B}.npz"
        np.savez_compressed(os.path.join(HERE, name), **res)
        sm = np.sort(res["score_map"].reshape(B, -1), axis=1)
        print(f"{name}: score range [{sm.min():.4f}, {sm.max():.4f}], top-2 margins {np.round(sm[:, -1] - sm[:, -2], 4)}, "
              f"pred_boxes[0] {res['pred_boxes'][0, 0]}, |block11| max {np.abs(res.get('act_block11', np.zeros(1))).max():.2f}")


if __name__ == "__main__":
    main()
