"""ctypes binding of ``libvittrack_hip.so`` (C ABI declared in ``include/vittrack.h``).

PyTorch is plumbing here: it owns device memory and streams; every kernel launch goes through
the C ABI with raw device pointers.  There is NO fallback: if the library is missing or a call
fails, an exception is raised (``VtError``).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: VITTRACK_LIB=<path> selects another build of the fp32 library (A/B tools: tools/ab_stages.py, tools/block_stamps.py)
LIB_PATH = os.environ.get("VITTRACK_LIB") or os.path.join(_HERE, "csrc", "libvittrack_hip.so")
#: the same sources built with every vit_48 contraction on f16 MFMA (BASELINE config 5; make -C csrc all)
LIB_PATH_F16 = os.path.join(_HERE, "csrc", "libvittrack_hip_f16.so")
PRECISIONS = ("f32", "f16")

#: every symbol include/vittrack.h declares (tests check the library exports all of them)
SYMBOLS = [
    "vt_last_error", "vt_version", "vt_create", "vt_destroy", "vt_load_weights", "vt_set_window",
    "vt_forward", "vt_stem", "vt_blocks", "vt_head", "vt_cal_bbox", "vt_graph_capture",
    "vt_graph_launch", "vt_graph_destroy", "vt_query", "vt_selftest_mfma", "vt_probe_clock", "vt_debug_stamps", "vt_crop", "vt_update_state",
    "vt_set_template", "vt_graph_capture_steps", "vt_update_state_record", "vt_track_step", "vt_set_form_batch",
    "vt_crop_u8", "vt_set_normalization", "vt_forward_u8", "vt_stem_u8", "vt_patch_u8_supported", "vt_crop_form", "vt_set_open_loop",
]


class VtError(RuntimeError):
    pass


class VtConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("template_size", "search_size", "channels", "heads", "depth",
                                         "head_channels", "stride", "max_batch")]


class VtTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class VtOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")]


_lib = None          # the fp32 library (kept as a module attribute: __graft_entry__.build() resets it)
_libs = {}


def lib(precision: str = "f32"):
    """Load a library once.  Raises VtError (never falls back) when it has not been built."""
    global _lib
    if precision not in PRECISIONS:
        raise VtError(f"unknown precision {precision!r}: 'f32' (default) or 'f16'")
    if precision == "f32" and _lib is not None:
        return _lib
    if precision != "f32" and precision in _libs:
        return _libs[precision]
    path = LIB_PATH if precision == "f32" else LIB_PATH_F16
    if not os.path.exists(path):
        raise VtError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C vittracker_amd/csrc all` (there is no CPU fallback)")
    # PyTorch-ROCm ships its own copy of the HIP runtime.  It has to be loaded before this library pulls in the
    # system libamdhip64: in the opposite order torch.cuda.is_available() turns False for the rest of the process
    # (seen when a process created a Model before it had ever imported torch).
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    L.vt_last_error.restype = C.c_char_p
    L.vt_version.restype = C.c_char_p
    L.vt_create.argtypes = [C.POINTER(VtConfig), C.POINTER(vp)]
    L.vt_destroy.argtypes = [vp]
    L.vt_destroy.restype = None
    L.vt_load_weights.argtypes = [vp, C.POINTER(VtTensor), i32]
    L.vt_set_window.argtypes = [vp, vp]
    L.vt_forward.argtypes = [vp, vp, vp, i32, vp, C.POINTER(VtOutputs)]
    L.vt_stem.argtypes = [vp, vp, vp, i32, vp, vp]
    L.vt_blocks.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.vt_head.argtypes = [vp, vp, i32, vp, C.POINTER(VtOutputs)]
    L.vt_cal_bbox.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.vt_graph_capture.argtypes = [vp, vp, vp, i32, C.POINTER(VtOutputs), C.POINTER(vp)]
    L.vt_graph_capture_steps.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(VtOutputs), C.POINTER(vp)]
    L.vt_graph_launch.argtypes = [vp, vp]
    L.vt_graph_destroy.argtypes = [vp]
    L.vt_graph_destroy.restype = None
    L.vt_query.argtypes = [vp] + [C.POINTER(i32)] * 4
    L.vt_selftest_mfma.argtypes = [vp]
    L.vt_probe_clock.argtypes = [i32, i32] + [C.POINTER(C.c_double)] * 3
    L.vt_debug_stamps.argtypes = [vp, i32, vp]
    L.vt_crop.argtypes = [vp, vp, i32, i32, vp, C.c_double, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp, vp, vp]
    L.vt_update_state.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.vt_update_state_record.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.vt_set_template.argtypes = [vp, vp, i32, vp]
    L.vt_set_form_batch.argtypes = [vp, i32]
    L.vt_track_step.argtypes = [vp, vp, i32, i32, vp, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp, vp, vp, vp, i32, vp]
    L.vt_crop_u8.argtypes = [vp, vp, i32, i32, vp, C.c_double, i32, i32, vp, vp, vp]
    L.vt_set_normalization.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.vt_forward_u8.argtypes = [vp, vp, vp, i32, vp, C.POINTER(VtOutputs)]
    L.vt_stem_u8.argtypes = [vp, vp, i32, vp, vp]
    L.vt_patch_u8_supported.argtypes = [vp, i32]
    L.vt_crop_form.argtypes = []
    L.vt_set_open_loop.argtypes = [vp, i32]
    if precision == "f32":
        _lib = L
    else:
        _libs[precision] = L
    return L


def _check(rc: int, what: str, L=None):
    if rc != 0:
        raise VtError(f"{what} failed ({rc}): {(L or lib()).vt_last_error().decode()}")


def _ptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise VtError("expected a contiguous float32 tensor on the GPU, got "
                      f"{type(t).__name__} {getattr(t, 'dtype', '')} {getattr(t, 'device', '')}")
    return C.c_void_p(t.data_ptr())


def _stream(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def crop_form() -> int:
    """Which crop kernel form the current device runs: 1 = 8-byte unaligned windows (crop_fast_kernel / crop_kernel<false>), 2 = the
    byte-load fallback the device self test selects on any mismatch."""
    return int(lib().vt_crop_form())


def probe_clock(iters=20000, waves_per_simd=1):
    """(shader MHz under dense f32 MFMA, SIMD cycles per MFMA, wall us) -- development probe."""
    v = [C.c_double() for _ in range(3)]
    _check(lib().vt_probe_clock(iters, waves_per_simd, *[C.byref(x) for x in v]), "vt_probe_clock")
    return tuple(x.value for x in v)


def selftest_mfma():
    _check(lib().vt_selftest_mfma(_stream(None)), "vt_selftest_mfma")


class Outputs:
    """Torch-owned device buffers for one forward of batch B (the dict the reference returns)."""

    def __init__(self, B, F, device):
        import torch
        self.score_map = torch.empty(B, 1, F, F, device=device)
        self.size_map = torch.empty(B, 2, F, F, device=device)
        self.offset_map = torch.empty(B, 2, F, F, device=device)
        self.pred_boxes = torch.empty(B, 4, device=device)
        self.hann_boxes = torch.empty(B, 4, device=device)
        self.conf = torch.empty(B, device=device)

    def struct(self):
        return VtOutputs(*[C.c_void_p(getattr(self, n).data_ptr()) for n, _ in VtOutputs._fields_])


class Graph:
    """A captured step.  It holds its Model (the graph's kernels read the model's weight and workspace
    buffers) and the tensors it was captured on; when the model is closed or re-sized the graph is
    invalidated and ``launch`` raises instead of replaying kernels over freed memory."""

    def __init__(self, handle, keep, model=None):
        self._h = handle
        self._keep = keep   # tensors the captured kernels read / write
        self._model = model
        self._L = model._L if model is not None else lib()
        self._valid = True

    def launch(self, stream=None):
        if not self._valid:
            raise VtError("this graph was captured from a model that has since been closed or re-sized "
                          "(its weight / workspace buffers are gone): capture it again")
        _check(self._L.vt_graph_launch(self._h, _stream(stream)), "vt_graph_launch", self._L)

    def _invalidate(self):
        self._valid = False

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_L", None) is not None:
            try:
                self._L.vt_graph_destroy(self._h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass
            self._h = None


class Model:
    """Owns one ``vt_model`` (weights + workspace on the current device)."""

    def __init__(self, template_size, search_size, channels=48, heads=1, depth=3, head_channels=32, stride=16,
                 max_batch=1, precision="f32"):
        """precision: 'f32' (exact fp32 contractions, the parity path) or 'f16' (vit_48 contractions on f16 MFMA,
        BASELINE config 5).  The ViT-Base path (channels=768) always contracts in bf16."""
        self.precision = precision
        self._L = lib(precision)
        self.cfg = VtConfig(template_size, search_size, channels, heads, depth, head_channels, stride, max_batch)
        h = C.c_void_p()
        _check(self._L.vt_create(C.byref(self.cfg), C.byref(h)), "vt_create", self._L)
        self._h = h
        q = [C.c_int32() for _ in range(4)]
        _check(self._L.vt_query(self._h, *[C.byref(v) for v in q]), "vt_query", self._L)
        self.len_z, self.len_x, self.feat_sz, self.channels = [v.value for v in q]
        self.L = self.len_z + self.len_x
        self.max_batch = max_batch
        self.template_size, self.search_size = template_size, search_size
        self._graphs = weakref.WeakSet()

    def live_graphs(self) -> int:
        return sum(1 for g in self._graphs if g._valid)

    # ---- argument checks (raw device pointers cross the C ABI: a wrong shape would read out of bounds)
    def _check_crops(self, z, x):
        B = z.shape[0]
        tz, tx = self.template_size, self.search_size
        if tuple(z.shape) != (B, 3, tz, tz) or tuple(x.shape) != (B, 3, tx, tx):
            raise VtError(f"expected z (B,3,{tz},{tz}) and x (B,3,{tx},{tx}) with one batch size, got "
                          f"{tuple(z.shape)} and {tuple(x.shape)}")
        self._check_batch(B)
        return B

    def _check_batch(self, B):
        if not (1 <= B <= self.max_batch):
            raise VtError(f"batch {B} outside [1, max_batch={self.max_batch}]")

    def _check_out(self, out, B):
        F = self.feat_sz
        want = {"score_map": (B, 1, F, F), "size_map": (B, 2, F, F), "offset_map": (B, 2, F, F), "pred_boxes": (B, 4),
                "hann_boxes": (B, 4), "conf": (B,)}
        for k, shp in want.items():
            if tuple(getattr(out, k).shape) != shp:
                raise VtError(f"output buffer {k} has shape {tuple(getattr(out, k).shape)}, want {shp}")

    def close(self):
        for g in list(getattr(self, "_graphs", ())):
            g._invalidate()
        if getattr(self, "_h", None) and getattr(self, "_L", None) is not None:
            try:
                self._L.vt_destroy(self._h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass
            self._h = None

    def __del__(self):
        self.close()

    def load_state_dict(self, sd: dict):
        """sd: name -> numpy array / torch tensor, reference ckpt['net'] layout (strict=False)."""
        keep, arr = [], (VtTensor * len(sd))()
        n = 0
        for k, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            if v.dtype.kind != "f":
                continue   # num_batches_tracked etc.
            a = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(a)
            arr[n] = VtTensor(k.encode(), a.ctypes.data, a.size)
            n += 1
        _check(self._L.vt_load_weights(self._h, arr, n), "vt_load_weights", self._L)

    def set_form_batch(self, n: int):
        """Choose the kernel forms as for a batch of n sequences (vt_set_form_batch): a shard of a group of n runs the forms the whole
        group would run, so a sequence's results do not depend on how the group is sharded.  0 = by each call's own batch."""
        _check(self._L.vt_set_form_batch(self._h, int(n)), "vt_set_form_batch", self._L)
        self.form_batch = int(n)

    def set_window(self, win):
        a = np.ascontiguousarray(np.asarray(win, dtype=np.float32).reshape(-1))
        if a.size != self.feat_sz ** 2:
            raise VtError(f"window must have {self.feat_sz ** 2} elements")
        _check(self._L.vt_set_window(self._h, a.ctypes.data), "vt_set_window", self._L)

    # ---- whole step
    def set_template(self, z, stream=None):
        """Exact template cache (vt_set_template): afterwards ``forward(None, x)`` / ``capture(None, x)`` skip the
        template's patch embedding and block 0's LayerNorm-1 + qkv of its rows."""
        B, tz = z.shape[0], self.template_size
        if tuple(z.shape) != (B, 3, tz, tz):
            raise VtError(f"expected z (B,3,{tz},{tz}), got {tuple(z.shape)}")
        self._check_batch(B)
        _check(self._L.vt_set_template(self._h, _ptr(z), B, _stream(stream)), "vt_set_template", self._L)
        self._tmpl_B = B

    def _check_x_only(self, x):
        B, tx = x.shape[0], self.search_size
        if tuple(x.shape) != (B, 3, tx, tx):
            raise VtError(f"expected x (B,3,{tx},{tx}), got {tuple(x.shape)}")
        self._check_batch(B)
        if getattr(self, "_tmpl_B", 0) < B:
            raise VtError(f"forward with z=None needs set_template() for at least {B} frames first")
        return B

    def forward(self, z, x, out: Outputs | None = None, stream=None) -> Outputs:
        B = self._check_crops(z, x) if z is not None else self._check_x_only(x)
        out = out or Outputs(B, self.feat_sz, x.device)
        self._check_out(out, B)
        st = out.struct()
        _check(self._L.vt_forward(self._h, _ptr(z), _ptr(x), B, _stream(stream), C.byref(st)), "vt_forward", self._L)
        return out

    def capture(self, z, x, out: Outputs | None = None) -> tuple[Graph, Outputs]:
        B = self._check_crops(z, x) if z is not None else self._check_x_only(x)
        out = out or Outputs(B, self.feat_sz, x.device)
        self._check_out(out, B)
        st = out.struct()
        g = C.c_void_p()
        _check(self._L.vt_graph_capture(self._h, _ptr(z), _ptr(x), B, C.byref(st), C.byref(g)), "vt_graph_capture", self._L)
        gr = Graph(g, (z, x, out), self)
        self._graphs.add(gr)
        return gr, out

    def capture_steps(self, zs, xs, outs=None) -> tuple[Graph, list]:
        """``len(xs)`` consecutive steps in ONE graph: step i is ``forward(zs[i], xs[i], outs[i])``.  ``zs`` may be None (cached
        template) or hold None entries; steps run in order and may share tensors.  One launch then advances ``len(xs)`` frames --
        the gap the runtime leaves between two graph launches (~7 us) is paid once per graph instead of once per step."""
        n = len(xs)
        if n < 1:
            raise VtError("capture_steps needs at least one step")
        zs = list(zs) if zs is not None else [None] * n
        if len(zs) != n or (outs is not None and len(outs) != n):
            raise VtError("capture_steps: zs, xs and outs must have the same length")
        B = None
        for z, x in zip(zs, xs):
            b = self._check_crops(z, x) if z is not None else self._check_x_only(x)
            if B is not None and b != B:
                raise VtError("capture_steps: every step must have the same batch size")
            B = b
        outs = list(outs) if outs is not None else [Outputs(B, self.feat_sz, xs[0].device) for _ in range(n)]
        for o in outs:
            self._check_out(o, B)
        zp = (C.c_void_p * n)(*[_ptr(z) for z in zs])
        xp = (C.c_void_p * n)(*[_ptr(x) for x in xs])
        sts = (VtOutputs * n)(*[o.struct() for o in outs])
        g = C.c_void_p()
        _check(self._L.vt_graph_capture_steps(self._h, n, zp, xp, B, sts, C.byref(g)), "vt_graph_capture_steps", self._L)
        gr = Graph(g, (zs, list(xs), outs), self)
        self._graphs.add(gr)
        return gr, outs

    # ---- stages
    def stem(self, z, x, stream=None):
        import torch
        B = self._check_crops(z, x)
        tok = torch.empty(B, self.L, self.channels, device=z.device)
        _check(self._L.vt_stem(self._h, _ptr(z), _ptr(x), B, _stream(stream), _ptr(tok)), "vt_stem", self._L)
        return tok

    def blocks(self, tokens, nblocks=-1, want_resid=False, stream=None, feat=None):
        import torch
        B = tokens.shape[0]
        self._check_batch(B)
        if tuple(tokens.shape) != (B, self.L, self.channels):
            raise VtError(f"tokens must be (B,{self.L},{self.channels}), got {tuple(tokens.shape)}")
        if feat is None:
            feat = torch.empty(B, self.len_x, self.channels, device=tokens.device)
        elif tuple(feat.shape) != (B, self.len_x, self.channels):
            raise VtError(f"feat must be (B,{self.len_x},{self.channels}), got {tuple(feat.shape)}")
        resid = torch.empty_like(tokens) if want_resid else None
        _check(self._L.vt_blocks(self._h, _ptr(tokens), B, nblocks, _stream(stream), _ptr(feat), _ptr(resid)), "vt_blocks", self._L)
        return (feat, resid) if want_resid else feat

    def head(self, feat, out: Outputs | None = None, stream=None) -> Outputs:
        B = feat.shape[0]
        self._check_batch(B)
        if tuple(feat.shape) != (B, self.len_x, self.channels):
            raise VtError(f"feat must be (B,{self.len_x},{self.channels}), got {tuple(feat.shape)}")
        out = out or Outputs(B, self.feat_sz, feat.device)
        self._check_out(out, B)
        st = out.struct()
        _check(self._L.vt_head(self._h, _ptr(feat), B, _stream(stream), C.byref(st)), "vt_head", self._L)
        return out

    # ---- pre / post steps of track() on the device
    def crop(self, frames, states, factor, out_size, mean, std, out=None, resize_factor=None, stream=None):
        """frames (B,H,W,3) uint8 cuda -- or PINNED host memory, which the kernel reads over the bus (a few sequences: no upload) --,
        states (B,4) float64 cuda -> (crops (B,3,T,T) fp32, resize_factor (B) fp64)."""
        import torch
        if not ((frames.is_cuda or frames.is_pinned()) and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4
                and frames.shape[3] == 3):
            raise VtError("frames must be a contiguous (B,H,W,3) uint8 tensor on the GPU (or in pinned host memory)")
        if not (states.is_cuda and states.dtype == torch.float64 and states.is_contiguous()):
            raise VtError("states must be a contiguous (B,4) float64 tensor on the GPU")
        B, H, W, _ = frames.shape
        if tuple(states.shape) != (B, 4):
            raise VtError(f"states must be ({B},4) for {B} frames, got {tuple(states.shape)}")
        if out is None:
            out = torch.empty(B, 3, out_size, out_size, device=states.device)
        elif tuple(out.shape) != (B, 3, out_size, out_size):
            raise VtError(f"crop output must be ({B},3,{out_size},{out_size}), got {tuple(out.shape)}")
        if resize_factor is None:
            resize_factor = torch.empty(B, dtype=torch.float64, device=states.device)
        elif tuple(resize_factor.shape) != (B,) or resize_factor.dtype != torch.float64 or not resize_factor.is_cuda:
            raise VtError(f"resize_factor must be a ({B},) float64 tensor on the GPU")
        m3 = (C.c_float * 3)(*[float(v) for v in mean])
        s3 = (C.c_float * 3)(*[float(v) for v in std])
        _check(self._L.vt_crop(self._h, C.c_void_p(frames.data_ptr()), H, W, C.c_void_p(states.data_ptr()), float(factor),
                             out_size, m3, s3, B, _stream(stream), _ptr(out), C.c_void_p(resize_factor.data_ptr())),
               "vt_crop", self._L)
        return out, resize_factor

    # ---- the uint8 patch path (round 6): sample_target's output goes to the stem as it is
    def crop_u8(self, frames, states, factor, out_size, out=None, resize_factor=None, stream=None):
        """sample_target alone: frames (B,H,W,3) uint8 (GPU or pinned), states (B,4) float64 cuda -> (patch (B,T,T,3) uint8 --
        the array the reference's sample_target returns --, resize_factor (B) fp64)."""
        import torch
        if not ((frames.is_cuda or frames.is_pinned()) and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4
                and frames.shape[3] == 3):
            raise VtError("frames must be a contiguous (B,H,W,3) uint8 tensor on the GPU (or in pinned host memory)")
        if not (states.is_cuda and states.dtype == torch.float64 and states.is_contiguous()):
            raise VtError("states must be a contiguous (B,4) float64 tensor on the GPU")
        B, H, W, _ = frames.shape
        if tuple(states.shape) != (B, 4):
            raise VtError(f"states must be ({B},4) for {B} frames, got {tuple(states.shape)}")
        if out is None:
            out = torch.empty(B, out_size, out_size, 3, dtype=torch.uint8, device=states.device)
        elif tuple(out.shape) != (B, out_size, out_size, 3) or out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous():
            raise VtError(f"patch output must be a contiguous ({B},{out_size},{out_size},3) uint8 tensor on the GPU")
        if resize_factor is None:
            resize_factor = torch.empty(B, dtype=torch.float64, device=states.device)
        elif tuple(resize_factor.shape) != (B,) or resize_factor.dtype != torch.float64 or not resize_factor.is_cuda:
            raise VtError(f"resize_factor must be a ({B},) float64 tensor on the GPU")
        _check(self._L.vt_crop_u8(self._h, C.c_void_p(frames.data_ptr()), H, W, C.c_void_p(states.data_ptr()), float(factor), out_size, B,
                                _stream(stream), C.c_void_p(out.data_ptr()), C.c_void_p(resize_factor.data_ptr())), "vt_crop_u8", self._L)
        return out, resize_factor

    def set_normalization(self, mean, std):
        """Preprocessor's mean / std for the uint8 entry points (folded into the stem's first layer; default: ImageNet)."""
        m3 = (C.c_float * 3)(*[float(v) for v in mean])
        s3 = (C.c_float * 3)(*[float(v) for v in std])
        _check(self._L.vt_set_normalization(self._h, m3, s3), "vt_set_normalization", self._L)

    def set_open_loop(self, on: bool = True):
        """vt_set_open_loop: track_step (and graphs captured from now on) leave `states` untouched; the step's box is in `record`."""
        _check(self._L.vt_set_open_loop(self._h, 1 if on else 0), "vt_set_open_loop", self._L)

    def patch_u8_supported(self, B: int) -> bool:
        return bool(self._L.vt_patch_u8_supported(self._h, int(B)))

    def _check_patch(self, xp):
        import torch
        B, tx = xp.shape[0], self.search_size
        if tuple(xp.shape) != (B, tx, tx, 3) or xp.dtype != torch.uint8 or not xp.is_cuda or not xp.is_contiguous():
            raise VtError(f"expected a contiguous uint8 patch (B,{tx},{tx},3) on the GPU, got {tuple(xp.shape)} {xp.dtype}")
        self._check_batch(B)
        return B

    def forward_u8(self, z, x_patch, out: Outputs | None = None, stream=None) -> Outputs:
        """Preprocessor.process + forward on the uint8 search patch of crop_u8; z: fp32 template crop or None (cached template)."""
        B = self._check_patch(x_patch)
        if z is None:
            if getattr(self, "_tmpl_B", 0) < B:
                raise VtError(f"forward_u8 with z=None needs set_template() for at least {B} frames first")
        elif tuple(z.shape) != (B, 3, self.template_size, self.template_size):
            raise VtError(f"expected z ({B},3,{self.template_size},{self.template_size}), got {tuple(z.shape)}")
        out = out or Outputs(B, self.feat_sz, x_patch.device)
        self._check_out(out, B)
        st = out.struct()
        _check(self._L.vt_forward_u8(self._h, _ptr(z), C.c_void_p(x_patch.data_ptr()), B, _stream(stream), C.byref(st)), "vt_forward_u8", self._L)
        return out

    def stem_u8(self, x_patch, tokens, stream=None):
        """Search rows of the token matrix from a uint8 patch (rows [len_z, L) of `tokens` (B,L,C) are written)."""
        B = self._check_patch(x_patch)
        if tuple(tokens.shape) != (B, self.len_z + self.len_x, self.channels):
            raise VtError(f"tokens must be ({B},{self.len_z + self.len_x},{self.channels}), got {tuple(tokens.shape)}")
        _check(self._L.vt_stem_u8(self._h, C.c_void_p(x_patch.data_ptr()), B, _stream(stream), _ptr(tokens)), "vt_stem_u8", self._L)
        return tokens

    def update_state(self, hann_boxes, resize_factor, states, search_size, H, W, margin=10, stream=None):
        B = states.shape[0]
        if (tuple(states.shape) != (B, 4) or tuple(hann_boxes.shape) != (B, 4) or tuple(resize_factor.shape) != (B,)
                or states.dtype != resize_factor.dtype or not states.is_cuda or not resize_factor.is_cuda):
            raise VtError(f"update_state wants hann_boxes ({B},4) fp32, resize_factor ({B},) fp64 and states ({B},4) fp64 "
                          f"on the GPU")
        _check(self._L.vt_update_state(self._h, _ptr(hann_boxes), C.c_void_p(resize_factor.data_ptr()), search_size, H, W,
                                     margin, B, _stream(stream), C.c_void_p(states.data_ptr())), "vt_update_state", self._L)
        return states

    def update_state_record(self, hann_boxes, conf, resize_factor, states, record, search_size, H, W, margin=10, stream=None):
        """update_state + the (B,5) float64 record [x, y, w, h, confidence] of the new state into `record`: a CUDA tensor or a
        PINNED host tensor (device-mapped: the kernel writes it over the bus, no copy afterwards)."""
        import torch
        B = states.shape[0]
        if (tuple(states.shape) != (B, 4) or tuple(hann_boxes.shape) != (B, 4) or tuple(resize_factor.shape) != (B,)
                or states.dtype != torch.float64 or resize_factor.dtype != torch.float64 or not states.is_cuda or not resize_factor.is_cuda):
            raise VtError(f"update_state_record wants hann_boxes ({B},4) fp32, resize_factor ({B},) fp64 and states ({B},4) fp64 on the GPU")
        if (tuple(record.shape) != (B, 5) or record.dtype != torch.float64 or not record.is_contiguous()
                or not (record.is_cuda or record.is_pinned())):
            raise VtError(f"record must be a contiguous ({B},5) float64 tensor on the GPU or in pinned host memory")
        _check(self._L.vt_update_state_record(self._h, _ptr(hann_boxes), _ptr(conf), C.c_void_p(resize_factor.data_ptr()), search_size,
                                            H, W, margin, B, _stream(stream), C.c_void_p(states.data_ptr()),
                                            C.c_void_p(record.data_ptr())), "vt_update_state_record", self._L)
        return record

    def track_step(self, frames, states, factor, mean, std, x, resize_factor, out: Outputs, record=None, margin=10, stream=None):
        """crop -> network on the cached template -> map back / clip / state update (-> record) as ONE library call (vt_track_step):
        the same kernels as crop() + forward(None, x) + update_state_record(), with the tail fused into the decode kernel on the
        small-batch path.  frames (B,H,W,3) uint8 on the GPU or pinned; x: the (B,3,S,S) crop workspace; states (B,4) / resize_factor
        (B,) float64 on the GPU; record: optional (B,5) float64, GPU or pinned."""
        import torch
        if not ((frames.is_cuda or frames.is_pinned()) and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 4
                and frames.shape[3] == 3):
            raise VtError("frames must be a contiguous (B,H,W,3) uint8 tensor on the GPU (or in pinned host memory)")
        B, H, W, _ = frames.shape
        if self._check_x_only(x) != B:
            raise VtError(f"crop workspace is for {x.shape[0]} frames, got {B}")
        if (tuple(states.shape) != (B, 4) or states.dtype != torch.float64 or not states.is_cuda or not states.is_contiguous()
                or tuple(resize_factor.shape) != (B,) or resize_factor.dtype != torch.float64 or not resize_factor.is_cuda):
            raise VtError(f"track_step wants states ({B},4) and resize_factor ({B},) float64 on the GPU")
        if record is not None and (tuple(record.shape) != (B, 5) or record.dtype != torch.float64 or not record.is_contiguous()
                                   or not (record.is_cuda or record.is_pinned())):
            raise VtError(f"record must be a contiguous ({B},5) float64 tensor on the GPU or in pinned host memory")
        self._check_out(out, B)
        st = out.struct()
        m3 = (C.c_float * 3)(*[float(v) for v in mean])
        s3 = (C.c_float * 3)(*[float(v) for v in std])
        _check(self._L.vt_track_step(self._h, C.c_void_p(frames.data_ptr()), H, W, C.c_void_p(states.data_ptr()), float(factor), m3, s3, B,
                                   _stream(stream), _ptr(x), C.c_void_p(resize_factor.data_ptr()), C.byref(st), margin,
                                   C.c_void_p(record.data_ptr()) if record is not None else None), "vt_track_step", self._L)
        return out

    def cal_bbox(self, score, size, offset, stream=None):
        import torch
        B, F = score.shape[0], self.feat_sz
        if tuple(score.shape) != (B, 1, F, F) or tuple(size.shape) != (B, 2, F, F) or tuple(offset.shape) != (B, 2, F, F):
            raise VtError(f"cal_bbox wants (B,1,{F},{F}), (B,2,{F},{F}), (B,2,{F},{F}) maps, got {tuple(score.shape)}, "
                          f"{tuple(size.shape)}, {tuple(offset.shape)}")
        bbox = torch.empty(B, 4, device=score.device)
        mx = torch.empty(B, device=score.device)
        _check(self._L.vt_cal_bbox(self._h, _ptr(score), _ptr(size), _ptr(offset), B, _stream(stream), _ptr(bbox), _ptr(mx)),
               "vt_cal_bbox", self._L)
        return bbox, mx
