"""Deterministic synthetic weights and crops for the vit_dist hot path.

There is no trained checkpoint in the reference tree (``.MISSING_LARGE_BLOBS``) and no network
here, so parity fixtures, tests and ``bench.py`` all draw weights and inputs from ONE frozen
stream generator (``numpy.random.RandomState(seed)``, draws in a fixed order, float64 then cast
to float32).  The same function runs in ``tests/golden/make_golden.py`` (which feeds the
reference model) and in the tests (which feed the oracle and the HIP path), so weights never
need to be committed; every fixture stores ``state_checksum`` to detect drift of this file.

Key names and shapes are the reference's ``ckpt['net']`` layout (SURVEY.md Appendix A;
``lib/models/vit_dist/vit_dist.py:10-75``, ``lib/models/layers/head.py:8-21,99-128``).
Scales are chosen so activations stay O(1) through the net, attention logits have unit-ish
spread (so softmax is neither flat nor one-hot) and BatchNorm running stats are non-trivial.
"""
from __future__ import annotations

import hashlib

import numpy as np

STEM_CH = lambda C: [3, C // 8, C // 4, C // 2, C]  # noqa: E731  (b16(): vit_dist.py:36-44)


def head_channels(C: int, W: int):
    """(cin, cout) of the four 3x3 head convs (lib/models/layers/head.py:106-109)."""
    return [(C, W), (W, W // 2), (W // 2, W // 4), (W // 4, W // 8)]


def synth_state_dict(seed: int = 0, C: int = 48, depth: int = 3, head_ch: int = 32,
                     len_z: int = 64, len_x: int = 256, mlp_ratio: int = 4) -> dict:
    rs = np.random.RandomState(seed)
    sd: dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rs.standard_normal(shape) * std).astype(np.float32)

    def uniform(shape, lo, hi):
        return rs.uniform(lo, hi, shape).astype(np.float32)

    def bn(prefix, n):
        sd[prefix + ".weight"] = uniform((n,), 0.8, 1.2)
        sd[prefix + ".bias"] = normal((n,), 0.1)
        sd[prefix + ".running_mean"] = normal((n,), 0.1)
        sd[prefix + ".running_var"] = uniform((n,), 0.5, 1.5)
        sd[prefix + ".num_batches_tracked"] = np.array(1234, dtype=np.int64)

    sd["pos_embed_z"] = normal((1, len_z, C), 0.1)
    sd["pos_embed_x"] = normal((1, len_x, C), 0.1)

    ch = STEM_CH(C)
    for i in range(4):
        cin, cout = ch[i], ch[i + 1]
        sd[f"patch_embed.net.{2 * i}.c.weight"] = normal((cout, cin, 3, 3), (2.0 / (cin * 9)) ** 0.5)
        bn(f"patch_embed.net.{2 * i}.bn", cout)

    H = C * mlp_ratio
    for b in range(depth):
        p = f"blocks.{b}."
        sd[p + "norm1.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
        sd[p + "norm1.bias"] = normal((C,), 0.05)
        sd[p + "attn.qkv.weight"] = normal((3 * C, C), 1.3 / C ** 0.5)
        sd[p + "attn.qkv.bias"] = normal((3 * C,), 0.1)
        sd[p + "attn.proj.weight"] = normal((C, C), 0.5 / C ** 0.5)
        sd[p + "attn.proj.bias"] = normal((C,), 0.05)
        sd[p + "norm2.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
        sd[p + "norm2.bias"] = normal((C,), 0.05)
        sd[p + "mlp.fc1.weight"] = normal((H, C), 1.0 / C ** 0.5)
        sd[p + "mlp.fc1.bias"] = normal((H,), 0.1)
        sd[p + "mlp.fc2.weight"] = normal((C, H), 0.5 / H ** 0.5)
        sd[p + "mlp.fc2.bias"] = normal((C,), 0.05)
    sd["norm.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
    sd["norm.bias"] = normal((C,), 0.05)

    for t in ("ctr", "offset", "size"):
        for i, (cin, cout) in enumerate(head_channels(C, head_ch)):
            sd[f"box_head.conv{i + 1}_{t}.0.weight"] = normal((cout, cin, 3, 3), (2.0 / (cin * 9)) ** 0.5)
            sd[f"box_head.conv{i + 1}_{t}.0.bias"] = normal((cout,), 0.1)
            bn(f"box_head.conv{i + 1}_{t}.1", cout)
        nout = 1 if t == "ctr" else 2
        # a hot last layer on the centre branch gives a peaked score map (argmax margins
        # well above fp32 noise); size/offset stay in their natural ranges
        std = {"ctr": 0.6, "offset": 0.3, "size": 0.6}[t]
        sd[f"box_head.conv5_{t}.weight"] = normal((nout, head_ch // 8, 1, 1), std)
        sd[f"box_head.conv5_{t}.bias"] = normal((nout,), 0.1)
    return sd


def synth_inputs(seed: int, B: int, template_size: int, search_size: int):
    """N(0,1) crops, as the reference's own profiler feeds (tracking/profile_model_cpu.py:104-105)."""
    rs = np.random.RandomState(1_000_003 + seed)
    z = rs.standard_normal((B, 3, template_size, template_size)).astype(np.float32)
    x = rs.standard_normal((B, 3, search_size, search_size)).astype(np.float32)
    return z, x


def synth_patches(seed: int, B: int, size: int) -> np.ndarray:
    """Seeded uint8 (B, size, size, 3) patches, the shape sample_target returns (lib/train/data/processing_utils.py:68-79): even
    frames are uniform noise over all 256 values (every byte value, the extremes included), odd ones a smooth field (a coarse
    random grid repeated 8 x 8) plus small noise and a black band at one border, as a crop that reaches over the frame has."""
    rs = np.random.RandomState(2_000_003 + seed)
    out = np.empty((B, size, size, 3), np.uint8)
    for b in range(B):
        if b % 2 == 0:
            out[b] = rs.randint(0, 256, (size, size, 3))
        else:
            coarse = rs.randint(0, 256, (size // 8, size // 8, 3)).astype(np.int64)
            img = np.repeat(np.repeat(coarse, 8, axis=0), 8, axis=1) + rs.randint(-6, 7, (size, size, 3))
            img = np.clip(img, 0, 255)
            img[:, : size // 5] = 0
            out[b] = img
    return out


def normalise_patches(patches: np.ndarray, reciprocal: bool = True) -> np.ndarray:
    """Preprocessor.process (lib/test/tracker/data_utils.py:11-17) in float32: HWC uint8 -> NCHW ((u / 255) - mean) / std.  torch on
    a GPU evaluates `tensor / 255.0` as a multiplication by float32(1 / 255) (reciprocal=True: what the reference's tracker runs and
    what vt_crop reproduces bit for bit); on the CPU it divides (reciprocal=False)."""
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    t = patches.astype(np.float32).transpose(0, 3, 1, 2)
    t = t * (np.float32(1.0) / np.float32(255.0)) if reciprocal else t / np.float32(255.0)
    return np.ascontiguousarray(((t - mean) / std).astype(np.float32))


def state_checksum(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------- ViT-Base (BASELINE config 4)
def synth_vitb_state_dict(seed: int = 0, C: int = 768, depth: int = 12, heads: int = 12, head_ch: int = 256,
                          len_z: int = 64, len_x: int = 256, patch: int = 16, mlp_ratio: int = 4, common_mode: float = 0.0) -> dict:
    """Seeded synthetic weights in the OSTrack ``ckpt['net']`` key layout (``backbone.*`` = VisionTransformer,
    lib/models/ostrack/vit.py:94-139 after ``finetune_track``, lib/models/ostrack/base_backbone.py:37-108;
    ``box_head.*`` = CenterPredictor(inplanes=768, channel=256), lib/models/layers/head.py:98-128).
    ``backbone.pos_embed`` / ``backbone.cls_token`` exist in the reference module but are unused by its forward."""
    rs = np.random.RandomState(50_000 + seed)
    sd: dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rs.standard_normal(shape) * std).astype(np.float32)

    def uniform(shape, lo, hi):
        return rs.uniform(lo, hi, shape).astype(np.float32)

    def bn(prefix, n):
        sd[prefix + ".weight"] = uniform((n,), 0.8, 1.2)
        sd[prefix + ".bias"] = normal((n,), 0.1)
        sd[prefix + ".running_mean"] = normal((n,), 0.1)
        sd[prefix + ".running_var"] = uniform((n,), 0.5, 1.5)
        sd[prefix + ".num_batches_tracked"] = np.array(1234, dtype=np.int64)

    b = "backbone."
    sd[b + "cls_token"] = np.zeros((1, 1, C), np.float32)
    sd[b + "pos_embed"] = np.zeros((1, 197, C), np.float32)
    # common_mode: the same offset on EVERY channel of every token row (in units of the rows' standard deviation, ~1 here): the residual
    # stream carries it through all twelve blocks, each LayerNorm has to remove it -- the case in which folding LayerNorm's mean
    # subtraction into the bf16 weights (vitb.hip: fold_layernorm) rounds the common mode with the row (tests/test_gpu_vitb.py)
    sd[b + "pos_embed_z"] = normal((1, len_z, C), 0.1) + np.float32(common_mode)
    sd[b + "pos_embed_x"] = normal((1, len_x, C), 0.1) + np.float32(common_mode)
    sd[b + "patch_embed.proj.weight"] = normal((C, 3, patch, patch), (1.0 / (3 * patch * patch)) ** 0.5)
    sd[b + "patch_embed.proj.bias"] = normal((C,), 0.1)
    H, hd = C * mlp_ratio, C // heads
    for i in range(depth):
        p = f"{b}blocks.{i}."
        sd[p + "norm1.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
        sd[p + "norm1.bias"] = normal((C,), 0.05)
        # per-head logits q.k / sqrt(hd) get a spread of ~1.3: softmax neither flat nor one-hot
        sd[p + "attn.qkv.weight"] = normal((3 * C, C), 1.15 / C ** 0.5)
        sd[p + "attn.qkv.bias"] = normal((3 * C,), 0.1)
        sd[p + "attn.proj.weight"] = normal((C, C), 0.5 / C ** 0.5)
        sd[p + "attn.proj.bias"] = normal((C,), 0.05)
        sd[p + "norm2.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
        sd[p + "norm2.bias"] = normal((C,), 0.05)
        sd[p + "mlp.fc1.weight"] = normal((H, C), 1.0 / C ** 0.5)
        sd[p + "mlp.fc1.bias"] = normal((H,), 0.1)
        sd[p + "mlp.fc2.weight"] = normal((C, H), 0.5 / H ** 0.5)
        sd[p + "mlp.fc2.bias"] = normal((C,), 0.05)
    sd[b + "norm.weight"] = (1.0 + rs.standard_normal(C) * 0.1).astype(np.float32)
    sd[b + "norm.bias"] = normal((C,), 0.05)
    for t in ("ctr", "offset", "size"):
        for i, (cin, cout) in enumerate(head_channels(C, head_ch)):
            sd[f"box_head.conv{i + 1}_{t}.0.weight"] = normal((cout, cin, 3, 3), (2.0 / (cin * 9)) ** 0.5)
            sd[f"box_head.conv{i + 1}_{t}.0.bias"] = normal((cout,), 0.1)
            bn(f"box_head.conv{i + 1}_{t}.1", cout)
        nout = 1 if t == "ctr" else 2
        std = {"ctr": 0.45, "offset": 0.15, "size": 0.3}[t]
        sd[f"box_head.conv5_{t}.weight"] = normal((nout, head_ch // 8, 1, 1), std)
        sd[f"box_head.conv5_{t}.bias"] = normal((nout,), 0.1)
    return sd
