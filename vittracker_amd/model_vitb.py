"""Model-level drop-in for the ViT-Base OSTrack path (BASELINE config 4): ``build_ostrack(cfg, training=False) -> OSTrack``
with the surface the reference's callers use (``lib/models/ostrack/ostrack.py:22-151,164-286``):

    net = build_ostrack(cfg, training=False)            # MODEL.BACKBONE.TYPE = vit_base_patch16_224, HEAD = CENTER / 256
    net.load_state_dict(ckpt['net'], strict=False)      # keys backbone.* / box_head.*
    net = net.cuda(); net.eval()
    out = net.forward(template=z, search=x)             # {'pred_boxes','score_map','size_map','offset_map'}
    net.box_head.cal_bbox(score, size, offset)

Arithmetic runs in ``libvittrack_hip.so`` (bf16 MFMA contractions, f32 accumulate / LayerNorm / softmax / residual stream);
no CPU execution path.  Only what the CENTER-head inference graph needs is accepted: the candidate-elimination / prompt /
draw variants of the reference (``vit_ce.py``, ``MODEL.PROCESS.*``) raise NotImplementedError."""
from __future__ import annotations

from collections import OrderedDict

from . import native, synth
from .config import geometry
from .model import CenterHead, OstrackDist


class OSTrack(OstrackDist):
    def __init__(self, cfg, depth=12, max_batch=1):
        g = geometry(cfg)
        if g["head_type"] != "CENTER":
            raise ValueError("HEAD TYPE %s is not supported." % g["head_type"])
        if str(cfg.MODEL.BACKBONE.TYPE) != "vit_base_patch16_224":
            raise NotImplementedError(f"backbone {cfg.MODEL.BACKBONE.TYPE!r}: only vit_base_patch16_224 is implemented")
        if str(cfg.MODEL.BACKBONE.CAT_MODE) != "direct":
            raise NotImplementedError("only CAT_MODE 'direct' (template tokens first) is implemented")
        self.geom, self.depth, self.mode, self.head_type, self.max_batch = g, depth, "eval", "CENTER", max_batch
        self.feat_sz_s, self.feat_len_s = g["feat_sz"], g["len_x"]
        self.box_head = CenterHead(self, g["feat_sz"], g["stride"])
        self._state = OrderedDict(synth.synth_vitb_state_dict(0, C=g["channels"], depth=depth, heads=g["heads"],
                                                              head_ch=g["head_channels"], len_z=g["len_z"], len_x=g["len_x"]))
        self._nat, self._device, self.training = None, None, False

    def forward(self, template, search, **unused):
        """OSTrack.forward (ostrack.py:47-120) for the plain backbone: no CE mask, no template / search pre-processing."""
        extra = {k: v for k, v in unused.items() if v is not None and v is not False}
        if extra:
            raise NotImplementedError(f"OSTrack.forward arguments {sorted(extra)} belong to variants that are not implemented")
        return OstrackDist.forward(self, template, search)

    __call__ = forward


def build_ostrack(cfg, training=False, max_batch=1):
    if training:
        raise NotImplementedError("inference graph only")
    return OSTrack(cfg, max_batch=max_batch)
