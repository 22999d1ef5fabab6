"""Lock-step batched tracking of B independent sequences on one GPU, state resident on the device.

The reference runs one tracker object per sequence and per process
(``lib/test/evaluation/tracker.py:90-152``, ``running.py:105-112``): per frame a host crop
(cv2), an H2D copy, ~130 kernel launches and a ``.tolist()`` sync.  Here B sequences advance
together: one H2D copy of the raw uint8 frames, then ``crop -> hipGraph(forward) -> state update``
on the stream with no host synchronisation; boxes are read back whenever the caller wants them
(every frame, or once at the end of the sequence).

Semantics per sequence are those of ``Vit_dist.initialize / track``
(``lib/test/tracker/vit_dist.py:53-148``); the crop and the state update run in
``vt_crop`` / ``vt_update_state`` (include/vittrack.h), the network in the captured graph.
"""
from __future__ import annotations

import numpy as np

from .config import geometry
from .host_ops import hann2d
from .model import build_ostrack_dist
from .native import VtError


def check_params_geometry(params, nat):
    """The tracker crops at params.{template,search}_size (TEST.*_SIZE) while the model is built from
    DATA.*.SIZE: a YAML that sets only one of them raises a shape error at the pos-embed add in the
    reference (lib/models/vit_dist/vit_dist.py:81-82); here the mismatch is refused up front."""
    if (params.template_size, params.search_size) != (nat.template_size, nat.search_size):
        raise VtError(f"tracker crop sizes (TEST.TEMPLATE_SIZE={params.template_size}, TEST.SEARCH_SIZE="
                      f"{params.search_size}) differ from the model geometry (DATA.TEMPLATE.SIZE={nat.template_size}, "
                      f"DATA.SEARCH.SIZE={nat.search_size})")


class BatchedVitTracker:
    def __init__(self, params, batch: int):
        import torch
        self.params = params
        self.cfg = params.cfg
        self.B = batch
        g = geometry(self.cfg)
        self.net = build_ostrack_dist(self.cfg, max_batch=batch)
        ckpt = getattr(params, "checkpoint", None)
        import os
        if ckpt and os.path.isfile(ckpt):
            self.net.load_state_dict(torch.load(ckpt, map_location="cpu")["net"], strict=False)
        elif not getattr(params, "allow_synthetic_weights", False):
            raise FileNotFoundError(f"checkpoint {ckpt!r} not found")
        self.net.cuda().eval()
        self.nat = self.net._native()
        check_params_geometry(params, self.nat)
        F = params.search_size // self.cfg.MODEL.BACKBONE.STRIDE
        self.nat.set_window(hann2d(torch.tensor([F, F]).long()).numpy())
        self.mean, self.std = list(self.cfg.DATA.MEAN), list(self.cfg.DATA.STD)
        dev = "cuda"
        self.z = torch.zeros(batch, 3, params.template_size, params.template_size, device=dev)
        self.x = torch.zeros(batch, 3, params.search_size, params.search_size, device=dev)
        self.states = torch.zeros(batch, 4, dtype=torch.float64, device=dev)
        self.rf = torch.zeros(batch, dtype=torch.float64, device=dev)
        self.graph, self.out = self.nat.capture(self.z, self.x)
        self.frames = None
        self._pinned = None
        self._h2d_done = None
        self._slot = 0
        self.hw = None
        self.frame_id = 0

    def _upload(self, frames):
        """Host frames go through two pinned staging buffers, each guarded by an event recorded after
        its H2D copy: with track(sync=False) the host may run ahead of the device, and a staging
        buffer is only rewritten once the copy that last read it has finished.  The device-side frame
        buffer is also double-buffered: the crop kernel of step f may still be reading it when the
        copy of step f+1 is queued on the same stream -- stream order covers that, the two slots
        simply keep a host thread that uploads from a side stream safe too."""
        import torch
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            t = frames
            if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3 or t.shape[0] != self.B or not t.is_contiguous():
                raise ValueError(f"frames must be a contiguous (B={self.B}, H, W, 3) uint8 tensor")
        else:
            a = np.ascontiguousarray(np.stack(frames) if not isinstance(frames, np.ndarray) else frames)
            if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3 or a.shape[0] != self.B:
                raise ValueError(f"frames must be (B={self.B}, H, W, 3) uint8")
            if self.frames is None or tuple(self.frames[0].shape) != a.shape:
                torch.cuda.current_stream().synchronize()      # nothing may still read the old buffers
                self.frames = [torch.empty(a.shape, dtype=torch.uint8, device="cuda") for _ in range(2)]
                self._pinned = [torch.empty(a.shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
                self._h2d_done = [None, None]
            k = self._slot
            self._slot ^= 1
            if self._h2d_done[k] is not None:
                self._h2d_done[k].synchronize()                # the copy that last read this staging buffer
            self._pinned[k].copy_(torch.from_numpy(a))
            self.frames[k].copy_(self._pinned[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._h2d_done[k] = ev
            t = self.frames[k]
        self.hw = (int(t.shape[1]), int(t.shape[2]))
        return t

    def initialize(self, frames, init_boxes):
        """frames: (B,H,W,3) uint8 (numpy, list of arrays or CUDA tensor); init_boxes: (B,4) [x,y,w,h]."""
        import torch
        fr = self._upload(frames)
        boxes = np.asarray(init_boxes, dtype=np.float64)
        if boxes.shape != (self.B, 4):
            raise ValueError(f"init_boxes must be (B={self.B}, 4) [x, y, w, h]")
        for f in (self.params.template_factor, self.params.search_factor):
            side = np.ceil(np.sqrt(boxes[:, 2] * boxes[:, 3]) * f)
            if not np.all(side >= 1):         # also catches NaN / negative sizes
                raise Exception("Too small bounding box.")   # processing_utils.py:33-34
        self.states.copy_(torch.as_tensor(boxes))
        self.nat.crop(fr, self.states, self.params.template_factor, self.params.template_size, self.mean, self.std,
                      out=self.z, resize_factor=self.rf)
        self.frame_id = 0

    def track(self, frames, sync: bool = True):
        """Advance every sequence by one frame.  Returns {'target_bbox': (B,4) float64, 'confidence': (B,)}
        as CPU tensors when sync=True, else the device tensors (valid until the next call)."""
        fr = self._upload(frames)
        H, W = self.hw
        self.frame_id += 1
        self.nat.crop(fr, self.states, self.params.search_factor, self.params.search_size, self.mean, self.std,
                      out=self.x, resize_factor=self.rf)
        self.graph.launch()
        self.nat.update_state(self.out.hann_boxes, self.rf, self.states, self.params.search_size, H, W, margin=10)
        if sync:
            return {"target_bbox": self.states.cpu(), "confidence": self.out.conf.cpu()}
        return {"target_bbox": self.states, "confidence": self.out.conf}
