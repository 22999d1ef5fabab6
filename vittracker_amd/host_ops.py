"""Host-side helpers of the tracker plugin (everything around the device step of ``track()``).

Each function names the reference code it stands in for.  These are the small, per-frame scalar
pieces; the device step lives behind :mod:`vittracker_amd.native`.
"""
from __future__ import annotations

import math

import numpy as np

try:  # the reference crops with OpenCV; use it when the box has it, else the numpy port below
    import cv2 as _cv
except Exception:  # noqa: BLE001
    _cv = None


# ------------------------------------------------------------------ lib/test/utils/hann.py:6-16
def hann1d(sz: int, centered: bool = True):
    import torch
    if not centered:
        raise NotImplementedError("only the centered window is used (lib/test/tracker/vit_dist.py:34)")
    return 0.5 * (1 - torch.cos((2 * math.pi / (sz + 1)) * torch.arange(1, sz + 1).float()))


def hann2d(sz, centered: bool = True):
    """(1,1,H,W) cosine window; same torch ops as the reference, so bit-identical on CPU."""
    h, w = int(sz[0]), int(sz[1])
    return hann1d(h, centered).reshape(1, 1, -1, 1) * hann1d(w, centered).reshape(1, 1, 1, -1)


# ------------------------------------------------------------------ lib/utils/box_ops.py:97-106
def clip_box(box: list, H, W, margin=0):
    x1, y1, w, h = box
    x2, y2 = x1 + w, y1 + h
    x1 = min(max(0, x1), W - margin)
    x2 = min(max(margin, x2), W)
    y1 = min(max(0, y1), H - margin)
    y2 = min(max(margin, y2), H)
    w = max(margin, x2 - x1)
    h = max(margin, y2 - y1)
    return [x1, y1, w, h]


# ------------------------------------------------- cv.resize(INTER_LINEAR) for uint8, numpy port
_COEF_BITS = 11
_ONE = 1 << _COEF_BITS


def _linear_coeffs(src: int, dst: int):
    """Source indices and 11-bit fixed-point weights of OpenCV's bilinear resize (pixel centres
    aligned: fx = (dx + 0.5) * scale - 0.5; clamped at the borders; weights = round(w * 2048))."""
    scale = src / dst
    fx = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    fx = fx.astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fr = fx - sx.astype(np.float32)
    lo = sx < 0
    sx[lo], fr[lo] = 0, 0.0
    hi = sx >= src - 1
    sx[hi], fr[hi] = src - 1, 0.0
    a1 = np.rint(fr * _ONE).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - fr) * _ONE).astype(np.int64)
    sx1 = np.minimum(sx + 1, src - 1)
    return sx, sx1, a0, a1


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Port of ``cv.resize(img, (out_w, out_h))`` (default INTER_LINEAR) for uint8 HxWxC input,
    following OpenCV's fixed-point scheme: horizontal pass in int32 with 11-bit weights, vertical
    pass ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.
    PARITY UNPINNED: cv2 is absent from the build image, so this has not been compared with the
    real library here; when cv2 is importable it is used instead."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    x0, x1, ax0, ax1 = _linear_coeffs(W, out_w)
    y0, y1, by0, by1 = _linear_coeffs(H, out_h)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]      # (H, out_w, C)
    s0, s1 = rows[y0], rows[y1]                                                         # (out_h, out_w, C)
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _resize_f64(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv.resize on a float64 HxW array (bilinear, float arithmetic) -- the attention-mask path."""
    H, W = a.shape
    def coef(src, dst):
        f = (np.arange(dst) + 0.5) * (src / dst) - 0.5
        s = np.floor(f).astype(np.int64)
        fr = f - s
        lo = s < 0
        s[lo], fr[lo] = 0, 0.0
        hi = s >= src - 1
        s[hi], fr[hi] = src - 1, 0.0
        return s, np.minimum(s + 1, src - 1), fr
    x0, x1, fx = coef(W, out_w)
    y0, y1, fy = coef(H, out_h)
    r = a[:, x0] * (1 - fx)[None] + a[:, x1] * fx[None]
    return r[y0] * (1 - fy)[:, None] + r[y1] * fy[:, None]


# ------------------------------------------- lib/train/data/processing_utils.py:12-79 (mask=None)
def sample_target(im, target_bb, search_area_factor, output_sz=None):
    """Square crop of side ceil(sqrt(w*h) * factor) centred on the box, zero padded outside the
    image, resized to output_sz.  Returns (crop uint8 HxWx3, resize_factor, att_mask bool)."""
    x, y, w, h = target_bb.tolist() if not isinstance(target_bb, list) else target_bb
    crop_sz = math.ceil(math.sqrt(w * h) * search_area_factor)
    if crop_sz < 1:
        raise Exception("Too small bounding box.")
    x1 = round(x + 0.5 * w - crop_sz * 0.5)   # Python banker's rounding, as in the reference
    x2 = x1 + crop_sz
    y1 = round(y + 0.5 * h - crop_sz * 0.5)
    y2 = y1 + crop_sz
    x1_pad = max(0, -x1)
    x2_pad = max(x2 - im.shape[1] + 1, 0)
    y1_pad = max(0, -y1)
    y2_pad = max(y2 - im.shape[0] + 1, 0)
    im_crop = im[y1 + y1_pad:y2 - y2_pad, x1 + x1_pad:x2 - x2_pad, :]
    # cv.copyMakeBorder(..., BORDER_CONSTANT) -> zeros
    im_crop_padded = np.pad(im_crop, ((y1_pad, y2_pad), (x1_pad, x2_pad), (0, 0)), mode="constant")
    H, W = im_crop_padded.shape[:2]
    att_mask = np.ones((H, W))
    end_x = -x2_pad if x2_pad != 0 else None
    end_y = -y2_pad if y2_pad != 0 else None
    att_mask[y1_pad:end_y, x1_pad:end_x] = 0
    if output_sz is None:
        return im_crop_padded, att_mask.astype(np.bool_), 1.0   # (sic) reference order, :76
    resize_factor = output_sz / crop_sz
    if _cv is not None:
        im_r = _cv.resize(im_crop_padded, (output_sz, output_sz))
        mask_r = _cv.resize(att_mask, (output_sz, output_sz)).astype(np.bool_)
    else:
        im_r = resize_bilinear_u8(np.ascontiguousarray(im_crop_padded), output_sz, output_sz)
        mask_r = _resize_f64(att_mask, output_sz, output_sz).astype(np.bool_)
    return im_r, resize_factor, mask_r


class NestedTensor:
    """lib/utils/misc.py:284-304 (only what the tracker touches)."""

    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def decompose(self):
        return self.tensors, self.mask


class Preprocessor:
    """lib/test/tracker/data_utils.py:6-17: uint8 HWC -> float NCHW / 255, ImageNet mean / std,
    on the GPU.  The H2D copy moves the uint8 crop (3 bytes / pixel), normalisation runs on the
    device."""

    def __init__(self):
        import torch
        self.mean = torch.tensor([0.485, 0.456, 0.406]).view((1, 3, 1, 1)).cuda()
        self.std = torch.tensor([0.229, 0.224, 0.225]).view((1, 3, 1, 1)).cuda()

    def process(self, img_arr: np.ndarray, amask_arr: np.ndarray):
        import torch
        img_tensor = torch.tensor(img_arr).cuda().float().permute((2, 0, 1)).unsqueeze(dim=0)
        img_tensor_norm = ((img_tensor / 255.0) - self.mean) / self.std
        amask_tensor = torch.from_numpy(amask_arr).to(torch.bool).cuda().unsqueeze(dim=0)
        return NestedTensor(img_tensor_norm.contiguous(), amask_tensor)
