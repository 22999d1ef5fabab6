"""Model-level drop-in: ``build_ostrack_dist(cfg) -> OstrackDist`` with the reference's surface
(``lib/models/vit_dist/vit_dist.py:57-100,122-153,159-198``; ``lib/models/layers/head.py:98-160``):

    net = build_ostrack_dist(cfg)                       # depth=3, mode='eval'
    net.load_state_dict(ckpt['net'], strict=False)      # reference key layout
    net = net.cuda(); net.eval()
    out = net.forward(z=z, x=x)                         # {'pred_boxes','score_map','size_map','offset_map'}
    boxes = net.box_head.cal_bbox(score, size, offset)  # on device

All arithmetic happens in ``libvittrack_hip.so`` through :mod:`vittracker_amd.native`; this class
only keeps the host copy of the state dict and the torch tensors that own device memory.  There
is no CPU execution path: ``forward`` on CPU tensors raises.
"""
from __future__ import annotations

from collections import OrderedDict, namedtuple

import numpy as np

from . import native, synth
from .config import geometry

_Incompatible = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class CenterHead:
    """The part of ``CenterPredictor`` the tracker touches: ``feat_sz`` and ``cal_bbox``
    (lib/models/layers/head.py:101,142-160)."""

    def __init__(self, owner: "OstrackDist", feat_sz: int, stride: int):
        self._owner = owner
        self.feat_sz = feat_sz
        self.stride = stride
        self.img_sz = feat_sz * stride

    def cal_bbox(self, score_map_ctr, size_map, offset_map, return_score=False):
        bbox, mx = self._owner._native().cal_bbox(score_map_ctr.contiguous(), size_map.contiguous(),
                                                  offset_map.contiguous())
        return (bbox, mx.view(-1, 1)) if return_score else bbox


class OstrackDist:
    def __init__(self, cfg, depth=3, mode="eval", max_batch=1):
        if mode != "eval":
            raise NotImplementedError("only the inference graph (mode='eval') is implemented; the distillation "
                                      "taps of lib/models/vit_dist/vit_dist.py:69-73,103-119 are training-only")
        g = geometry(cfg)
        if g["head_type"] != "CENTER":
            raise ValueError("HEAD TYPE %s is not supported." % g["head_type"])  # head.py:361
        self.geom = g
        self.depth = depth
        self.mode = mode
        self.head_type = "CENTER"
        self.max_batch = max_batch
        self.feat_sz_s = g["feat_sz"]
        self.feat_len_s = g["len_x"]
        self.box_head = CenterHead(self, g["feat_sz"], g["stride"])
        # host copy of the parameters, reference names; deterministic init (the reference uses
        # torch's random init; nothing on the inference path depends on it)
        self._state = OrderedDict(synth.synth_state_dict(0, C=g["channels"], depth=depth, head_ch=g["head_channels"],
                                                         len_z=g["len_z"], len_x=g["len_x"]))
        self._nat = None
        self._device = None
        self.training = False

    # ---- nn.Module-like surface used by the reference's callers
    def state_dict(self):
        import torch
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._state.items())

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._state if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._state]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k in self._state:
            if k in state_dict:
                v = state_dict[k]
                v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                if tuple(v.shape) != tuple(self._state[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model "
                                       f"{tuple(self._state[k].shape)}")
                self._state[k] = v.astype(self._state[k].dtype, copy=True)
        if self._nat is not None:
            self._nat.load_state_dict(self._state)    # same buffers, new contents: captured graphs stay valid
        return _Incompatible(missing, unexpected)

    def cuda(self, device=None):
        import torch
        if device is not None:
            torch.cuda.set_device(device)
        self._device = torch.device("cuda", torch.cuda.current_device())
        self._native()
        return self

    def to(self, device):
        import torch
        d = torch.device(device)
        if d.type != "cuda":
            raise native.VtError("vittracker_amd has no CPU execution path (MI355X only)")
        return self.cuda(d.index)

    def eval(self):
        self.training = False
        return self

    def reserve(self, max_batch: int):
        """Re-size the native workspace (weights are re-uploaded).  Graphs captured from the old
        workspace are invalidated: their ``launch`` raises instead of replaying over freed buffers."""
        self.max_batch = max_batch
        if self._nat is not None:
            self._nat.close()       # invalidates every live native.Graph of this model
            self._nat = None
            self._native()
        return self

    def _native(self) -> native.Model:
        if self._nat is None:
            if self._device is None:
                raise native.VtError("call .cuda() first: vittracker_amd has no CPU execution path")
            g = self.geom
            self._nat = native.Model(g["template_size"], g["search_size"], g["channels"], g["heads"], self.depth,
                                     g["head_channels"], g["stride"], self.max_batch)
            self._nat.load_state_dict(self._state)
        return self._nat

    # ---- the hot path
    def forward(self, z, x):
        """OstrackDist.forward (vit_dist.py:77-100): z (B,3,Tz,Tz), x (B,3,Tx,Tx) fp32 on the GPU."""
        if not (getattr(z, "is_cuda", False) and getattr(x, "is_cuda", False)):
            raise native.VtError("forward() needs CUDA(HIP) tensors: there is no CPU fallback")
        B = x.shape[0]
        if z.shape[0] != B:
            raise ValueError(f"batch mismatch: z {tuple(z.shape)} vs x {tuple(x.shape)}")
        if B > self.max_batch:
            nat = self._nat
            if nat is not None and nat.live_graphs():
                raise native.VtError(
                    f"batch {B} exceeds max_batch={self.max_batch} and {nat.live_graphs()} captured graph(s) still use "
                    f"this model's workspace: call reserve({B}) explicitly and re-capture them")
            self.reserve(B)
        o = self._native().forward(z.float().contiguous(), x.float().contiguous())
        return {"pred_boxes": o.pred_boxes.view(B, 1, 4), "score_map": o.score_map, "size_map": o.size_map,
                "offset_map": o.offset_map,
                # extras the tracker uses to avoid a second decode + sync (lib/test/tracker/vit_dist.py:103-109,148)
                "hann_boxes": o.hann_boxes, "conf": o.conf}

    __call__ = forward


def build_ostrack_dist(cfg, depth=3, mode="eval", max_batch=1):
    """lib/models/vit_dist/vit_dist.py:159-164.  ``cfg.MODEL.PRETRAIN_FILE`` is only read when
    mode != 'eval' in the reference (:165), i.e. never on this path."""
    return OstrackDist(cfg, depth=depth, mode=mode, max_batch=max_batch)
