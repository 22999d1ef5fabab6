"""CPU oracle (PyTorch, fp32) for the vit_dist path.  TEST INFRASTRUCTURE ONLY.

Same role and same restrictions as ``vt_oracle_np.py``: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.  It exists
because the reference's CPU path *is* PyTorch (``tracking/profile_model_cpu.py:99-106`` builds the
``nn.Module`` and times ``model(template, search)``), so the fair CPU baseline on the GPU box is
the same ATen op sequence (conv2d / batch_norm / linear / softmax / gelu ...) on that box's host
cores; the reference itself cannot travel there.

The module tree uses the reference's parameter names (``patch_embed.net.{0,2,4,6}.{c,bn}``,
``blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}``, ``norm``,
``box_head.conv{1..4}_{ctr,offset,size}.{0,1}``, ``box_head.conv5_*``) so a
reference ``ckpt['net']`` loads with ``load_state_dict(strict=False)``.

Parity status: PINNED via tests/test_oracle_golden.py against tests/golden/*.npz (outputs of the
reference's own model).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


class _ConvBN(nn.Sequential):
    """conv (no bias) + BatchNorm2d with child names 'c' / 'bn' (vit_dist.py:10-19)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("c", nn.Conv2d(cin, cout, 3, 2, 1, bias=False))
        self.add_module("bn", nn.BatchNorm2d(cout))


class _Stem(nn.Module):
    """b16() wrapped as ``patch_embed.net`` (vit_dist.py:36-54): indices 0,2,4,6 are Conv+BN,
    1,3,5 are Hardswish."""

    def __init__(self, C):
        super().__init__()
        ch = [3, C // 8, C // 4, C // 2, C]
        layers = []
        for i in range(4):
            layers.append(_ConvBN(ch[i], ch[i + 1]))
            if i < 3:
                layers.append(nn.Hardswish())
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x).flatten(2).transpose(1, 2)


class _Attn(nn.Module):
    """timm Attention at the arguments the reference uses (restated lib/models/layers/attn.py:33-59)."""

    def __init__(self, C, heads):
        super().__init__()
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.qkv = nn.Linear(C, 3 * C, bias=True)
        self.proj = nn.Linear(C, C)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, C, H):
        super().__init__()
        self.fc1 = nn.Linear(C, H)
        self.fc2 = nn.Linear(H, C)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))  # exact (erf) GELU


class _Block(nn.Module):
    """Pre-LN residual block (restated lib/models/layers/attn_blocks.py:117-133)."""

    def __init__(self, C, heads, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(C)
        self.attn = _Attn(C, heads)
        self.norm2 = nn.LayerNorm(C)
        self.mlp = _Mlp(C, C * mlp_ratio)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


def _cbr(cin, cout):
    # lib/models/layers/head.py:8-21 (freeze_bn=False)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _CenterHead(nn.Module):
    """CenterPredictor (lib/models/layers/head.py:98-201)."""

    def __init__(self, C, W, feat_sz):
        super().__init__()
        self.feat_sz = feat_sz
        for t, nout in (("ctr", 1), ("offset", 2), ("size", 2)):
            chans = [C, W, W // 2, W // 4, W // 8]
            for i in range(4):
                setattr(self, f"conv{i + 1}_{t}", _cbr(chans[i], chans[i + 1]))
            setattr(self, f"conv5_{t}", nn.Conv2d(W // 8, nout, 1))

    def _tower(self, x, t):
        for i in range(1, 5):
            x = getattr(self, f"conv{i}_{t}")(x)
        return getattr(self, f"conv5_{t}")(x)

    def cal_bbox(self, score, size, offset, return_score=False):
        mx, idx = torch.max(score.flatten(1), dim=1, keepdim=True)
        iy = idx // self.feat_sz
        ix = idx % self.feat_sz
        gi = idx.unsqueeze(1).expand(idx.shape[0], 2, 1)
        sz = size.flatten(2).gather(2, gi).squeeze(-1)
        off = offset.flatten(2).gather(2, gi).squeeze(-1)
        bbox = torch.cat([(ix.float() + off[:, :1]) / self.feat_sz,
                          (iy.float() + off[:, 1:]) / self.feat_sz, sz], dim=1)
        return (bbox, mx) if return_score else bbox

    def forward(self, x):
        clamp = lambda y: torch.clamp(y.sigmoid(), min=1e-4, max=1 - 1e-4)  # noqa: E731
        score = clamp(self._tower(x, "ctr"))
        size = clamp(self._tower(x, "size"))
        offset = self._tower(x, "offset")
        return score, self.cal_bbox(score, size, offset), size, offset


class OracleVitDist(nn.Module):
    """OstrackDist, eval graph only (lib/models/vit_dist/vit_dist.py:57-100,122-153)."""

    def __init__(self, C=48, heads=1, depth=3, head_ch=32, len_z=64, len_x=256):
        super().__init__()
        self.patch_embed = _Stem(C)
        self.pos_embed_z = nn.Parameter(torch.zeros(1, len_z, C))
        self.pos_embed_x = nn.Parameter(torch.zeros(1, len_x, C))
        self.blocks = nn.ModuleList([_Block(C, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(C)
        self.feat_sz = int(round(math.sqrt(len_x)))
        self.box_head = _CenterHead(C, head_ch, self.feat_sz)

    def forward(self, z, x):
        zt = self.patch_embed(z) + self.pos_embed_z
        xt = self.patch_embed(x) + self.pos_embed_x
        t = torch.cat((zt, xt), dim=1)
        for blk in self.blocks:
            t = blk(t)
        t = self.norm(t)
        B, _, C = t.shape
        f = t[:, -self.feat_sz ** 2:].transpose(1, 2).reshape(B, C, self.feat_sz, self.feat_sz)
        score, bbox, size, offset = self.box_head(f.contiguous())
        return {"pred_boxes": bbox.view(B, 1, 4), "score_map": score, "size_map": size,
                "offset_map": offset}


def build_from_state(sd_np: dict, heads=1, depth=3) -> OracleVitDist:
    """Instantiate at the geometry implied by a (numpy) state dict and load it."""
    C = sd_np["norm.weight"].shape[0]
    m = OracleVitDist(C=C, heads=heads, depth=depth,
                      head_ch=sd_np["box_head.conv1_ctr.0.weight"].shape[0],
                      len_z=sd_np["pos_embed_z"].shape[1], len_x=sd_np["pos_embed_x"].shape[1])
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()},
                                            strict=False)
    assert not missing, missing
    return m.eval()
