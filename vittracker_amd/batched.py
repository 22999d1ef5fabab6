"""Lock-step batched tracking of B independent sequences on one GPU, state resident on the device.

The reference runs one tracker object per sequence and per process
(``lib/test/evaluation/tracker.py:90-152``, ``running.py:105-112``): per frame a host crop
(cv2), an H2D copy, ~130 kernel launches and a ``.tolist()`` sync.  Here B sequences advance
together: one H2D copy of the raw uint8 frames, then ``crop -> forward -> state update`` as one captured
graph on the stream with no host synchronisation; boxes are read back whenever the caller wants them
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
    ZERO_COPY_MAX_BYTES = 2 << 20      # host frames up to this size (all sequences together) are read in place from pinned memory

    def __init__(self, params, batch: int, form_batch: int = 0):
        """form_batch: this tracker steps `batch` sequences of a larger group of `form_batch` (a shard of ShardedBatchedTracker, a rank
        of run_dataset_batched): its kernels take the forms the whole group would run (native.Model.set_form_batch), so that the
        group's results do not depend on how it is sharded."""
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
        if form_batch:
            self.nat.set_form_batch(form_batch)
        check_params_geometry(params, self.nat)
        F = params.search_size // self.cfg.MODEL.BACKBONE.STRIDE
        self.nat.set_window(hann2d(torch.tensor([F, F]).long()).numpy())
        self.mean, self.std = list(self.cfg.DATA.MEAN), list(self.cfg.DATA.STD)
        dev = "cuda"
        self.z = torch.zeros(batch, 3, params.template_size, params.template_size, device=dev)
        self.x = torch.zeros(batch, 3, params.search_size, params.search_size, device=dev)
        self.states = torch.zeros(batch, 4, dtype=torch.float64, device=dev)
        self.rf = torch.zeros(batch, dtype=torch.float64, device=dev)
        from .native import Outputs
        self.out = Outputs(batch, F, dev)
        self.graph = None            # captured at the first initialize(): the forward reads the cached template (z = None)
        self._chunk_graphs = {}      # (frame buffer address, n, H, W) -> whole-step graph of n frames (track_chunk)
        self._held = None            # hold_states(): open loop, every step searches around the boxes it was switched on with
        self._chunk_buf = None
        self.frames = None
        self._fast_shape = None
        self._fast = [None, None]    # per pinned frame slot: (numpy view of the slot, its whole-step graph, numpy view of its pinned record)
        self._slot = 0
        self.hw = None
        self.frame_id = 0

    def _upload(self, frames):
        """Host frames are copied straight from the caller's array into one of two device frame buffers with ONE blocking copy
        (the runtime stages pageable memory itself: 53 GB/s for a 236 MB batch, ~20 us for one 230 KB frame; an explicit
        pinned staging buffer filled by the CPU and DMA-ed from measured 6.7 GB/s and 2 ms -- tools/upload_probe.py).  The
        copy returns when the caller's array has been read, so there is no staging buffer a later call could overwrite; it is
        queued behind the previous step on the stream, whose kernels may still be reading the OTHER device buffer."""
        import torch
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            t = frames
            if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3 or t.shape[0] != self.B or not t.is_contiguous():
                raise ValueError(f"frames must be a contiguous (B={self.B}, H, W, 3) uint8 tensor")
        else:
            a = np.ascontiguousarray(np.stack(frames) if not isinstance(frames, np.ndarray) else frames)
            if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3 or a.shape[0] != self.B:
                raise ValueError(f"frames must be (B={self.B}, H, W, 3) uint8")
            zero_copy = a.nbytes <= self.ZERO_COPY_MAX_BYTES
            if self.frames is None or tuple(self.frames[0].shape) != a.shape or self.frames[0].is_cuda == zero_copy:
                torch.cuda.current_stream().synchronize()      # nothing may still read the old buffers
                self._fast = [None, None]
                if zero_copy:
                    # A few small frames (the plugin's one-sequence step): no upload at all.  The CPU copies the frame into one of
                    # two PINNED host slots (230 KB: ~3 us) and the crop kernel reads the pixels it needs over the bus (+1 us on the
                    # kernel) -- against ~19 us for the blocking copy from pageable memory (tools/zero_copy_probe.py).
                    self.frames = [torch.empty(a.shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
                    self._frames_np = [f.numpy() for f in self.frames]
                    self._slot_done = [None, None]
                else:
                    self.frames = [torch.empty(a.shape, dtype=torch.uint8, device="cuda") for _ in range(2)]
            k = self._slot
            self._slot ^= 1
            if zero_copy:
                if self._slot_done[k] is not None:
                    self._slot_done[k].synchronize()           # the step that last read this slot (two calls ago) has finished
                np.copyto(self._frames_np[k], a)
                self._cur_slot = k
            else:
                self.frames[k].copy_(torch.from_numpy(a))
                self._cur_slot = None
            t = self.frames[k]
        self.hw = (int(t.shape[1]), int(t.shape[2]))
        return t

    def _mark_slot(self, synced=False):
        """After the launch that reads the current pinned frame slot: an event the next writer of that slot waits for (none when
        the caller synchronises on this step anyway)."""
        import torch
        k = getattr(self, "_cur_slot", None)
        if k is None:
            return
        if synced:
            self._slot_done[k] = None
            return
        if self._slot_done[k] is None:
            self._slot_done[k] = torch.cuda.Event()
        self._slot_done[k].record()

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
        self._mark_slot()
        # The template never changes after this (lib/test/tracker/vit_dist.py:57-60): its patch embedding and block 0's
        # LayerNorm-1 + qkv rows are computed once here (vt_set_template, bit-identical to recomputing them every frame);
        # graphs captured with z = None read that cache, so they stay valid across re-initialisation.
        self.nat.set_template(self.z)
        if self.graph is None:
            self.graph, _ = self.nat.capture(None, self.x, self.out)
        self.frame_id = 0

    def hold_states(self, on: bool = True):
        """Benchmark / study aid: with on=True every later step searches around the boxes the sequences have NOW (open loop,
        vt_set_open_loop: the step's boxes are in the records and the returned confidence, `states` stay) instead of around the previous
        step's result.  On synthetic noise frames with random weights a free-running tracker's boxes drift to the clip limits within a
        few frames, and the crop then reads windows no real sequence has (tracking/track_batch_demo.py --hold-boxes)."""
        self._held = bool(on) or None
        self.nat.set_open_loop(bool(on))
        self._chunk_graphs.clear()      # the flag is an argument of the captured kernels
        self._fast = [None, None]

    def track_record(self, frames):
        """track(frames, sync=True) for callers that want the raw per-sequence records: a (B,5) float64 numpy array
        [x, y, w, h, confidence] that the caller owns.
        The plugin tracker's per-frame call: for host frames that are read in place (see _upload) a repeat call is one CPU copy,
        one graph launch, one stream synchronisation and nothing else."""
        import torch
        a = frames
        if type(a) is np.ndarray and self.frames is not None and a.shape == self._fast_shape and a.dtype == np.uint8 and a.flags.c_contiguous:
            k = self._slot
            ent = self._fast[k]
            if ent is not None:
                if self.graph is None:
                    raise VtError("track before initialize")
                self._slot = k ^ 1
                if self._slot_done[k] is not None:            # a step queued by track(sync=False) / initialize() may still read this slot
                    self._slot_done[k].synchronize()
                    self._slot_done[k] = None
                np.copyto(ent[0], a)
                self._cur_slot = k
                ent[1].replay()
                torch.cuda.current_stream().synchronize()
                self.frame_id += 1
                return ent[2][0].copy()                       # the caller owns it (40 bytes per sequence), as on the slow path
        out = self.track(frames, sync=True)
        return torch.cat([out["target_bbox"], out["confidence"].double().view(-1, 1)], dim=1).numpy()

    def track(self, frames, sync: bool = True):
        """Advance every sequence by one frame.  Returns {'target_bbox': (B,4) float64, 'confidence': (B,)}
        as CPU tensors when sync=True, else the device tensors (valid until the next call)."""
        if self.graph is None:
            raise VtError("track before initialize")
        fr = self._upload(frames)
        H, W = self.hw
        self.frame_id += 1
        if self.frames is not None and any(fr.data_ptr() == f.data_ptr() for f in self.frames):
            # host frames land in one of two fixed device slots: the whole step (crop -> forward -> state update) is one
            # captured graph per slot -- one launch instead of three, no launch gaps inside the step
            g, rec, host, _ = self._chunk_graph(fr.unsqueeze(0), to_host=sync)
            g.replay()
            self._mark_slot(synced=sync)
            k = getattr(self, "_cur_slot", None)
            if sync and k is not None and host is None and self._fast[k] is None:      # pinned frame slot + pinned record: the fast path of track_record
                self._fast[k] = (self._frames_np[k], g, rec.numpy())
                self._fast_shape = tuple(self._frames_np[k].shape)
            if sync:
                r = self._records(rec, host)
                return {"target_bbox": r[0, :, :4], "confidence": r[0, :, 4].float()}
            return {"target_bbox": self.states, "confidence": self.out.conf}
        # a caller-owned device tensor (a new address every call would mean a new capture every call): eager launches, as ONE library
        # call (vt_track_step: crop -> network on the cached template with the state update on the head's decoding lane)
        self.nat.track_step(fr, self.states, self.params.search_factor, self.mean, self.std, self.x, self.rf, self.out, margin=10)
        if sync:
            return {"target_bbox": self.states.cpu(), "confidence": self.out.conf.cpu()}
        return {"target_bbox": self.states, "confidence": self.out.conf}

    # ---- n frames per launch ------------------------------------------------------------------------------------
    def _chunk_graph(self, buf, to_host=False):
        """The whole tracker step -- crop -> forward -> map back / clip / state update -> record -- for the n frames of `buf`
        (n,B,H,W,3), captured once per frame buffer into ONE graph: no host work and one launch gap per n frames."""
        import torch
        n, _, H, W, _ = buf.shape
        to_host = bool(to_host) and self.B * n <= 64
        key = (buf.data_ptr(), n, H, W, to_host)
        hit = self._chunk_graphs.get(key)
        if hit is not None:
            return hit
        if len(self._chunk_graphs) >= 4:                      # a caller cycling through many buffers: keep the table small
            self._chunk_graphs.pop(next(iter(self._chunk_graphs)))
        # Per-frame result records [x, y, w, h, confidence] (float64).  vt_update_state_record writes them where the caller reads
        # them: small batches straight into PINNED host memory (device-mapped; 40 bytes per sequence over the bus, no copy kernel
        # and no device -> host copy after the step -- the plugin's one-sequence step), large batches into device memory (read
        # back on demand).
        # (to_host: the caller synchronises after every launch -- track(sync=True) of a few sequences)
        rec = torch.empty(n, self.B, 5, dtype=torch.float64).pin_memory() if to_host else torch.empty(n, self.B, 5, dtype=torch.float64, device="cuda")
        host = None if to_host else torch.empty(n, self.B, 5, dtype=torch.float64).pin_memory()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            cs = torch.cuda.current_stream()
            for i in range(n):
                # vt_track_step = vt_crop -> vt_forward on the cached template -> vt_update_state_record in one library call
                self.nat.track_step(buf[i], self.states, self.params.search_factor, self.mean, self.std, self.x, self.rf, self.out,
                                    record=rec[i], margin=10, stream=cs)
        torch.cuda.current_stream().wait_stream(side)
        self._chunk_graphs[key] = (g, rec, host, buf)   # buf: the graph's kernels read it, keep it alive
        return self._chunk_graphs[key]

    def track_chunk(self, frames, sync: bool = True):
        """Advance every sequence by n = len(frames) frames with ONE graph launch: frames (n,B,H,W,3) uint8, a CUDA tensor (used in
        place: the graph is captured on its address and reused whenever the same buffer comes back) or host data (uploaded into
        an internal n-deep device buffer).  For callers that have the next frames at hand -- offline evaluation reads whole
        sequences (lib/test/evaluation/running.py:56-102).  Frame by frame the same kernels run in the same order as in
        track(): results are bit-identical.  Returns {'target_bbox': (n,B,4) float64, 'confidence': (n,B)}."""
        import torch
        if self.graph is None:
            raise VtError("track_chunk before initialize")
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            buf = frames
            if buf.dtype != torch.uint8 or buf.dim() != 5 or buf.shape[1] != self.B or buf.shape[4] != 3 or not buf.is_contiguous():
                raise ValueError(f"frames must be a contiguous (n, B={self.B}, H, W, 3) uint8 tensor")
        else:
            a = np.ascontiguousarray(np.stack(frames) if not isinstance(frames, np.ndarray) else frames)
            if a.dtype != np.uint8 or a.ndim != 5 or a.shape[1] != self.B or a.shape[4] != 3:
                raise ValueError(f"frames must be (n, B={self.B}, H, W, 3) uint8")
            if self._chunk_buf is None or tuple(self._chunk_buf.shape) != a.shape:
                torch.cuda.current_stream().synchronize()
                self._chunk_buf = torch.empty(a.shape, dtype=torch.uint8, device="cuda")
            self._chunk_buf.copy_(torch.from_numpy(a))        # one blocking copy from the caller's array (see _upload)
            buf = self._chunk_buf
        n = int(buf.shape[0])
        self.hw = (int(buf.shape[2]), int(buf.shape[3]))
        g, rec, host, _ = self._chunk_graph(buf, to_host=sync)
        g.replay()
        self.frame_id += n
        if sync:
            r = self._records(rec, host)
            return {"target_bbox": r[:, :, :4], "confidence": r[:, :, 4].float()}
        return {"target_bbox": rec[:, :, :4], "confidence": rec[:, :, 4]}

    @staticmethod
    def _records(rec, host):
        """The records of the launch that was just queued, as a CPU tensor the caller owns: one stream synchronisation; records
        that live in pinned host memory were written by the kernel itself, device records take one async copy first."""
        import torch
        if host is not None:
            host.copy_(rec, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return (rec if host is None else host).clone()


class ShardedBatchedTracker:
    """B sequences as `shards` independent groups, each a BatchedVitTracker of its own -- its own model workspaces, states, frame
    buffers and graphs -- stepped under its own HIP stream.  Every kernel of the step is one workgroup per frame with most of a
    CU's LDS, so a single stream leaves the chip idle at each kernel's ramp and tail and between two graph launches; with two
    groups in flight the next graph's first kernel takes over the CUs the previous graph's last kernel is leaving (DESIGN.md 4.5:
    +5 % end to end at B = 2 x 256, +10-18 % on the network alone).  Same interface and the same results per sequence as one
    BatchedVitTracker of B sequences (initialize / track / track_chunk; host or device frames), bit for bit: every shard's model
    selects its kernel forms by the WHOLE group's size (vt_set_form_batch; the forms of a stage differ by fp32 rounding, and the
    library picks them by batch size), and within one form a sequence's kernels do not depend on which other sequences share
    its batch (tests/test_gpu_harness.py: B = 256 in two shards at both geometries)."""

    def __init__(self, params, batch: int, shards: int = 2):
        import torch
        self.sizes = self.shard_sizes(batch, shards)
        self.offsets = [sum(self.sizes[:k]) for k in range(len(self.sizes))]
        self.B = int(batch)
        self.params = params
        self.trackers = [BatchedVitTracker(params, n, form_batch=self.B) for n in self.sizes]
        self.streams = [torch.cuda.Stream() for _ in self.sizes]
        self.frame_id = 0

    @staticmethod
    def shard_sizes(batch: int, shards: int):
        """Contiguous groups of near-equal size, the first `batch % shards` one sequence larger; never an empty group."""
        shards = max(1, min(int(shards), int(batch)))
        base, extra = divmod(int(batch), shards)
        return [base + (1 if k < extra else 0) for k in range(shards)]

    def _slices(self):
        return [slice(o, o + n) for o, n in zip(self.offsets, self.sizes)]

    def _each(self, fn):
        """fn(tracker, slice) under the shard's stream, for every shard, without waiting in between."""
        import torch
        cur = torch.cuda.current_stream()
        out = []
        for t, st, sl in zip(self.trackers, self.streams, self._slices()):
            st.wait_stream(cur)                      # what the caller queued (e.g. the frames' producer) comes first
            with torch.cuda.stream(st):
                out.append(fn(t, sl))
        return out

    def _join(self):
        import torch
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)

    def initialize(self, frames, init_boxes):
        boxes = np.asarray(init_boxes, dtype=np.float64)
        self._each(lambda t, sl: t.initialize(frames[sl], boxes[sl]))
        self._join()
        self.frame_id = 0

    def track(self, frames, sync: bool = True):
        """One frame for every sequence: frames (B,H,W,3) uint8, host or device.  sync=True: {'target_bbox': (B,4) float64,
        'confidence': (B,)} on the host; sync=False: the same keys as LISTS of per-shard device tensors (valid until the next call)."""
        import torch
        res = self._each(lambda t, sl: t.track(frames[sl], sync=False))
        self.frame_id += 1
        if not sync:
            self._join()
            return {"target_bbox": [r["target_bbox"] for r in res], "confidence": [r["confidence"] for r in res]}
        host = []
        for r, st in zip(res, self.streams):
            with torch.cuda.stream(st):
                host.append((r["target_bbox"].to("cpu", non_blocking=True), r["confidence"].to("cpu", non_blocking=True)))
        for st in self.streams:
            st.synchronize()
        return {"target_bbox": torch.cat([h[0] for h in host]), "confidence": torch.cat([h[1] for h in host]).float()}

    def track_chunk(self, frames, sync: bool = True):
        """n frames per launch and shard: frames (n,B,H,W,3) uint8 host data, or a list of per-shard contiguous CUDA tensors
        (n,B_k,H,W,3) -- a slice of one device tensor along its batch axis is not contiguous, and the graphs are captured on
        buffer addresses.  Returns what BatchedVitTracker.track_chunk returns, batch axes concatenated (sync=True) or per shard."""
        import torch
        if isinstance(frames, (list, tuple)) and len(frames) == len(self.trackers) and all(isinstance(f, torch.Tensor) for f in frames):
            parts = list(frames)
        else:
            a = np.stack(frames) if not isinstance(frames, np.ndarray) else frames
            parts = [np.ascontiguousarray(a[:, sl]) for sl in self._slices()]
        idx = {id(t): k for k, t in enumerate(self.trackers)}
        res = self._each(lambda t, sl: t.track_chunk(parts[idx[id(t)]], sync=False))
        self.frame_id += int(parts[0].shape[0])
        if not sync:
            self._join()
            return {"target_bbox": [r["target_bbox"] for r in res], "confidence": [r["confidence"] for r in res]}
        host = []
        for r, st in zip(res, self.streams):
            with torch.cuda.stream(st):
                host.append((r["target_bbox"].to("cpu", non_blocking=True), r["confidence"].to("cpu", non_blocking=True)))
        for st in self.streams:
            st.synchronize()
        return {"target_bbox": torch.cat([h[0] for h in host], dim=1), "confidence": torch.cat([h[1] for h in host], dim=1).float()}
