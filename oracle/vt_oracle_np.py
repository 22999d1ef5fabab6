"""CPU oracle (numpy) for the vit_dist per-frame inference path.  TEST INFRASTRUCTURE ONLY.

This is a restatement, op by op, of the reference's hot path for checking the HIP kernels; it is
not part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product path (``vittracker_amd``) never does and fails
loudly when the HIP extension is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` hold outputs of the reference's own PyTorch model
(imported from /root/reference by ``tests/golden/make_golden.py``, with ``timm`` /
``torchvision`` / ``easydict`` absent from this image and replaced as that script documents);
``tests/test_oracle_golden.py`` checks every function below against them.

All citations are relative to the reference root (``/root/reference``).  Arithmetic runs in the
dtype of the inputs (float32 for parity; float64 gives a higher-precision "truth" used to put
the reference's own rounding error and the kernels' error on one scale).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

BN_EPS = 1e-5   # torch.nn.BatchNorm2d default, lib/models/vit_dist/vit_dist.py:16
LN_EPS = 1e-5   # nn.LayerNorm default, used by timm Block and lib/models/vit_dist/vit_dist.py:75


# ----------------------------------------------------------------------------- primitives
def conv3x3(x, w, b, stride):
    """3x3 conv, zero padding 1 (``torch.nn.Conv2d(.., 3, stride, 1)``).

    lib/models/vit_dist/vit_dist.py:14-15 (stride 2, no bias) and
    lib/models/layers/head.py:16-18 (stride 1, bias).
    x (B,Cin,H,W), w (Cout,Cin,3,3), b (Cout,) or None -> (B,Cout,Ho,Wo).
    """
    B, Cin, H, W = x.shape
    Ho = (H + 2 - 3) // stride + 1
    Wo = (W + 2 - 3) // stride + 1
    xp = np.zeros((B, Cin, H + 2, W + 2), dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, w.shape[0], Ho, Wo), dtype=x.dtype)
    for r in range(3):
        for s in range(3):
            patch = xp[:, :, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride]
            out += np.einsum("oc,bcpq->bopq", w[:, :, r, s], patch, optimize=True)
    if b is not None:
        out += b[None, :, None, None]
    return out


def batchnorm_eval(x, gamma, beta, mean, var, eps=BN_EPS):
    """BatchNorm2d in eval mode (running stats): vit_dist.py:16-19, head.py:19."""
    inv = gamma / np.sqrt(var + x.dtype.type(eps))
    return (x - mean[None, :, None, None]) * inv[None, :, None, None] + beta[None, :, None, None]


def hardswish(x):
    """nn.Hardswish: x * relu6(x + 3) / 6  (vit_dist.py:39,41,43,162)."""
    return x * np.clip(x + x.dtype.type(3), 0, 6) / x.dtype.type(6)


def gelu_erf(x):
    """nn.GELU() exact form used by timm Mlp (act_layer=nn.GELU): 0.5 x (1 + erf(x / sqrt 2))."""
    return (x * x.dtype.type(0.5) * (1 + _erf(x / x.dtype.type(math.sqrt(2.0))))).astype(x.dtype)


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """nn.LayerNorm over the last dim, biased variance, eps inside the sqrt."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def linear(x, w, b):
    """nn.Linear: y = x W^T + b, W is (out, in)."""
    return x @ w.T + b


def sigmoid_clamped(x):
    """``torch.clamp(x.sigmoid_(), min=1e-4, max=1-1e-4)``  (head.py:177-179)."""
    one = x.dtype.type(1)
    y = one / (one + np.exp(-x))
    return np.clip(y, x.dtype.type(1e-4), x.dtype.type(1 - 1e-4))


# ----------------------------------------------------------------------------- model pieces
def _cast(sd, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in sd.items()}


def stem(img, sd):
    """LevitPatchEmbedding.forward (vit_dist.py:47-54) = b16() (vit_dist.py:36-44).

    4 x (conv3x3 s2 p1 no-bias + BN), Hardswish after the first three, then
    ``flatten(2).transpose(1, 2)``: token index = y * W + x.
    Returns (tokens (B, HW, C), [per-layer NCHW activations]).
    """
    acts = []
    y = img
    for i in range(4):
        p = f"patch_embed.net.{2 * i}"
        y = conv3x3(y, sd[p + ".c.weight"], None, 2)
        y = batchnorm_eval(y, sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                           sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])
        if i < 3:
            y = hardswish(y)
        acts.append(y)
    B, C, H, W = y.shape
    return y.reshape(B, C, H * W).transpose(0, 2, 1).copy(), acts


def attention(x, sd, p, num_heads):
    """timm Attention.forward, restated in-tree at lib/models/layers/attn.py:33-59.

    qkv Linear -> reshape (B,N,3,h,C/h) -> permute(2,0,3,1,4); attn = (q @ k^T) * scale;
    softmax(dim=-1); (attn @ v).transpose(1,2).reshape(B,N,C); proj.
    """
    B, N, C = x.shape
    hd = C // num_heads
    qkv = linear(x, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(0, 1, 3, 2)) * x.dtype.type(hd ** -0.5)
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    a = e / e.sum(axis=-1, keepdims=True)
    o = (a @ v).transpose(0, 2, 1, 3).reshape(B, N, C)
    return linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])


def block(x, sd, i, num_heads):
    """timm Block.forward (pre-LN residual), restated at lib/models/layers/attn_blocks.py:130-133.

    x = x + attn(norm1(x)); x = x + mlp(norm2(x)); mlp = fc2(GELU(fc1(.))).
    """
    p = f"blocks.{i}."
    x = x + attention(layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]), sd, p, num_heads)
    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = gelu_erf(linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def head_tower(f, sd, t):
    """One CenterPredictor branch (head.py:182-200): 4 x [conv3x3+bias -> BN -> ReLU], conv1x1."""
    acts = []
    y = f
    for i in range(1, 5):
        p = f"box_head.conv{i}_{t}"
        y = conv3x3(y, sd[p + ".0.weight"], sd[p + ".0.bias"], 1)
        y = batchnorm_eval(y, sd[p + ".1.weight"], sd[p + ".1.bias"],
                           sd[p + ".1.running_mean"], sd[p + ".1.running_var"])
        y = np.maximum(y, 0)
        acts.append(y)
    w5 = sd[f"box_head.conv5_{t}.weight"][:, :, 0, 0]
    y = np.einsum("oc,bcpq->bopq", w5, y) + sd[f"box_head.conv5_{t}.bias"][None, :, None, None]
    return y, acts


def cal_bbox(score, size, offset, feat_sz):
    """CenterPredictor.cal_bbox (head.py:142-160).

    idx = argmax over the flattened score map (first maximum on ties, as torch.max on CPU);
    idx_y = idx // feat_sz, idx_x = idx % feat_sz; gather size/offset at idx;
    bbox = [(idx_x + off_x) / feat_sz, (idx_y + off_y) / feat_sz, w, h].
    Returns (bbox (B,4), max_score (B,), idx (B,)).
    """
    B = score.shape[0]
    flat = score.reshape(B, -1)
    idx = flat.argmax(axis=1)
    mx = flat[np.arange(B), idx]
    iy = idx // feat_sz
    ix = idx % feat_sz
    sz = size.reshape(B, 2, -1)[np.arange(B), :, idx]
    off = offset.reshape(B, 2, -1)[np.arange(B), :, idx]
    dt = score.dtype.type
    bbox = np.stack([(ix.astype(score.dtype) + off[:, 0]) / dt(feat_sz),
                     (iy.astype(score.dtype) + off[:, 1]) / dt(feat_sz),
                     sz[:, 0], sz[:, 1]], axis=1)
    return bbox.astype(score.dtype), mx, idx


def hann1d(sz):
    """lib/test/utils/hann.py:6-9 (centered=True): 0.5 (1 - cos(2 pi i / (sz + 1))), i = 1..sz.
    Evaluated in float32 like the reference's torch code."""
    i = np.arange(1, sz + 1, dtype=np.float32)
    return (np.float32(0.5) * (1 - np.cos(np.float32(2 * math.pi / (sz + 1)) * i))).astype(np.float32)


def hann2d(sz):
    """lib/test/utils/hann.py:14-16: outer product, shape (1,1,sz,sz)."""
    w = hann1d(sz)
    return (w.reshape(1, 1, -1, 1) * w.reshape(1, 1, 1, -1)).astype(np.float32)


def forward(sd, z, x, num_heads=1, depth=3, dtype=np.float32, want_acts=False):
    """OstrackDist.forward (vit_dist.py:77-100) + forward_head (vit_dist.py:122-153)
    + CenterPredictor.forward (head.py:130-140) + the tracker's device tail
    (lib/test/tracker/vit_dist.py:103-105: response = hann * score; cal_bbox(response, ..)).

    Returns the reference's dict (pred_boxes (B,1,4), score_map, size_map, offset_map) plus
    'hann_boxes' (B,4), 'conf' (B,) = max of the un-windowed score (vit_dist tracker :148),
    and with want_acts the per-stage activations.
    """
    sd = _cast(sd, dtype)
    z = z.astype(dtype)
    x = x.astype(dtype)
    acts = {}
    zt, za = stem(z, sd)                                   # vit_dist.py:78
    xt, xa = stem(x, sd)                                   # vit_dist.py:79
    acts["stem_z"], acts["stem_x"] = za, xa
    zt = zt + sd["pos_embed_z"]                            # vit_dist.py:81
    xt = xt + sd["pos_embed_x"]                            # vit_dist.py:82
    X = np.concatenate([zt, xt], axis=1)                   # vit_dist.py:84  template rows first
    acts["tokens"] = X
    for i in range(depth):                                 # vit_dist.py:88-89
        X = block(X, sd, i, num_heads)
        acts[f"block{i}"] = X
    X = layer_norm(X, sd["norm.weight"], sd["norm.bias"])  # vit_dist.py:94
    acts["norm"] = X
    len_x = xt.shape[1]
    F = int(round(math.sqrt(len_x)))
    B, L, C = X.shape
    # vit_dist.py:126-129: last feat_len_s tokens, (B,HW,C) -> (B,C,F,F); f[b,c,p,q] = X[b, Lz+p*F+q, c]
    f = X[:, -len_x:].transpose(0, 2, 1).reshape(B, C, F, F)
    ctr, a_ctr = head_tower(f, sd, "ctr")
    off, a_off = head_tower(f, sd, "offset")
    siz, a_siz = head_tower(f, sd, "size")
    acts["head_ctr"], acts["head_offset"], acts["head_size"] = a_ctr, a_off, a_siz
    score = sigmoid_clamped(ctr)                           # head.py:201
    size = sigmoid_clamped(siz)
    bbox, mx, idx = cal_bbox(score, size, off, F)          # head.py:136 (raw score)
    win = hann2d(F).astype(dtype)
    hbox, _, hidx = cal_bbox(win * score, size, off, F)    # tracker vit_dist.py:104-105
    out = {"pred_boxes": bbox.reshape(B, 1, 4), "score_map": score, "size_map": size,
           "offset_map": off, "hann_boxes": hbox, "conf": mx, "idx": idx, "hann_idx": hidx}
    if want_acts:
        out["acts"] = acts
    return out


def top2_margin(m):
    """Gap between the largest and second-largest value of each (B, ...) map: how far the
    argmax is from flipping ('Argmax discontinuity')."""
    flat = np.sort(m.reshape(m.shape[0], -1), axis=1)
    return flat[:, -1] - flat[:, -2]


# ----------------------------------------------------------------------------- tracker tail
def map_box_back(state, pred_box, resize_factor, search_size):
    """Vit_dist.map_box_back (lib/test/tracker/vit_dist.py:150-156)."""
    cx_prev, cy_prev = state[0] + 0.5 * state[2], state[1] + 0.5 * state[3]
    cx, cy, w, h = pred_box
    half_side = 0.5 * search_size / resize_factor
    cx_real = cx + (cx_prev - half_side)
    cy_real = cy + (cy_prev - half_side)
    return [cx_real - 0.5 * w, cy_real - 0.5 * h, w, h]


def clip_box(box, H, W, margin=0):
    """lib/utils/box_ops.py:97-106."""
    x1, y1, w, h = box
    x2, y2 = x1 + w, y1 + h
    x1 = min(max(0, x1), W - margin)
    x2 = min(max(margin, x2), W)
    y1 = min(max(0, y1), H - margin)
    y2 = min(max(margin, y2), H)
    w = max(margin, x2 - x1)
    h = max(margin, y2 - y1)
    return [x1, y1, w, h]


# ----------------------------------------------------------------------------- work counts
def macs_per_frame(template_size, search_size, C=48, depth=3, W=32, mlp_ratio=4):
    """Algorithmic multiply-accumulates of one forward; FLOP = 2 MAC."""
    ch = [3, C // 8, C // 4, C // 2, C]

    def stem_macs(T):
        tot, s = 0, T
        for i in range(4):
            s //= 2
            tot += s * s * ch[i] * ch[i + 1] * 9
        return tot

    Lz, Lx = (template_size // 16) ** 2, (search_size // 16) ** 2
    L = Lz + Lx
    lin = L * (C * 3 * C + C * C + 2 * C * C * mlp_ratio)
    att = 2 * L * L * C
    F2 = Lx
    tower = 9 * (C * W + W * (W // 2) + (W // 2) * (W // 4) + (W // 4) * (W // 8))
    head = F2 * (3 * tower + (W // 8) * 5)
    return {"stem": stem_macs(template_size) + stem_macs(search_size),
            "blocks": depth * (lin + att), "head": head,
            "total": stem_macs(template_size) + stem_macs(search_size) + depth * (lin + att) + head}
