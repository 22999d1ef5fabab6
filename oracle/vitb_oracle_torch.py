"""CPU oracle (PyTorch, fp32) for the ViT-Base OSTrack path (BASELINE config 4).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this; the product
(``vittracker_amd``) never does.  Restates, op by op:

  patch embedding      lib/models/layers/patch_embed.py:20-32     Conv2d(3, 768, 16, stride 16) -> flatten(2).transpose(1,2)
  token assembly       lib/models/ostrack/base_backbone.py:110-137  x, z embedded; += pos_embed_x / pos_embed_z; cat((z, x)) ('direct')
  blocks               lib/models/ostrack/vit.py:39-91            pre-LN, 12 heads x 64, scale 64^-0.5, exact GELU, LayerNorm eps 1e-6 (:130)
  final norm           lib/models/ostrack/base_backbone.py:150
  head                 lib/models/ostrack/ostrack.py:122-151 + lib/models/layers/head.py:98-201 (CENTER, channel 256)

Module / parameter names follow the reference's ``ckpt['net']`` (``backbone.*``, ``box_head.*``).

Parity status: PINNED.  tests/golden/ref_vitb_*.npz hold outputs of the reference's own ``build_ostrack`` model
(imported from /root/reference by tests/golden/make_golden_vitb.py under the stubs that script documents);
tests/test_oracle_golden.py checks this module against them.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from .vt_oracle_torch import _CenterHead

LN_EPS = 1e-6   # partial(nn.LayerNorm, eps=1e-6), lib/models/ostrack/vit.py:130


class _Attn(nn.Module):
    def __init__(self, C, heads):
        super().__init__()
        self.heads, self.scale = heads, (C // heads) ** -0.5
        self.qkv = nn.Linear(C, 3 * C, bias=True)
        self.proj = nn.Linear(C, C)

    def forward(self, x):                                           # vit.py:51-66
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, C, H):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, H), nn.Linear(H, C)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, C, heads, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(C, eps=LN_EPS)
        self.attn = _Attn(C, heads)
        self.norm2 = nn.LayerNorm(C, eps=LN_EPS)
        self.mlp = _Mlp(C, C * mlp_ratio)

    def forward(self, x):                                           # vit.py:88-90
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, C, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, C, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class _Backbone(nn.Module):
    def __init__(self, C, heads, depth, len_z, len_x, patch):
        super().__init__()
        self.patch_embed = _PatchEmbed(C, patch)
        self.pos_embed_z = nn.Parameter(torch.zeros(1, len_z, C))
        self.pos_embed_x = nn.Parameter(torch.zeros(1, len_x, C))
        self.blocks = nn.Sequential(*[_Block(C, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(C, eps=LN_EPS)

    def forward(self, z, x, acts=None):
        x = self.patch_embed(x) + self.pos_embed_x                 # base_backbone.py:113-120
        z = self.patch_embed(z) + self.pos_embed_z
        t = torch.cat((z, x), dim=1)                                # combine_tokens(mode='direct'), utils.py:13-14
        if acts is not None:
            acts["tokens"] = t
        for i, blk in enumerate(self.blocks):
            t = blk(t)
            if acts is not None:
                acts[f"block{i}"] = t
        t = self.norm(t)
        if acts is not None:
            acts["norm"] = t
        return t


class OracleOSTrack(nn.Module):
    def __init__(self, C=768, heads=12, depth=12, head_ch=256, len_z=64, len_x=256, patch=16):
        super().__init__()
        self.backbone = _Backbone(C, heads, depth, len_z, len_x, patch)
        self.feat_sz = int(round(math.sqrt(len_x)))
        self.box_head = _CenterHead(C, head_ch, self.feat_sz)

    def forward(self, template, search, acts=None):
        t = self.backbone(template, search, acts)
        B, _, C = t.shape
        f = t[:, -self.feat_sz ** 2:].transpose(1, 2).reshape(B, C, self.feat_sz, self.feat_sz)   # ostrack.py:126-129
        score, bbox, size, offset = self.box_head(f.contiguous())
        return {"pred_boxes": bbox.view(B, 1, 4), "score_map": score, "size_map": size, "offset_map": offset}


def build_from_state(sd_np: dict, heads=12) -> OracleOSTrack:
    C = sd_np["backbone.norm.weight"].shape[0]
    depth = 1 + max(int(k.split(".")[2]) for k in sd_np if k.startswith("backbone.blocks."))
    m = OracleOSTrack(C=C, heads=heads, depth=depth, head_ch=sd_np["box_head.conv1_ctr.0.weight"].shape[0],
                      len_z=sd_np["backbone.pos_embed_z"].shape[1], len_x=sd_np["backbone.pos_embed_x"].shape[1],
                      patch=sd_np["backbone.patch_embed.proj.weight"].shape[-1])
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=False)
    assert not missing, missing
    return m.eval()


def macs_per_frame(template_size=128, search_size=256, C=768, depth=12, W=256, patch=16):
    """SURVEY.md 8(d) general formulas at ViT-Base."""
    lz, lx = (template_size // patch) ** 2, (search_size // patch) ** 2
    L = lz + lx
    return {"patch": L * 3 * patch * patch * C, "blocks": depth * (12 * C * C * L + 2 * L * L * C),
            "head": lx * (3 * 9 * (C * W + W * W // 2 + W * W // 8 + W * W // 32) + 5 * W // 8)}
