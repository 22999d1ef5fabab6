"""Sequences the harness runs on (the part of ``lib/test/evaluation/data.py:21-169`` the tracker loop
touches: ``name``, ``frames``, ``dataset``, ``ground_truth_rect``, ``object_ids``, ``init_info()``,
``frame_info()``).  A frame is either a path (``.npy`` HxWx3 uint8, or any format PIL / cv2 reads) or an
in-memory HxWx3 uint8 array -- synthetic sequences never touch the disk.  Dataset parsers of the reference
(OTB, LaSOT, GOT-10k ... ``lib/test/evaluation/*dataset.py``) are benchmark I/O and out of scope; the
``folder`` dataset below reads the common ``<seq>/img/*.{jpg,png,npy}`` + ``groundtruth.txt`` layout."""
from __future__ import annotations

import glob
import os
import re

import numpy as np


class Sequence:
    def __init__(self, name, frames, dataset, ground_truth_rect, object_ids=None):
        self.name = name
        self.frames = list(frames)
        self.dataset = dataset
        self.ground_truth_rect = None if ground_truth_rect is None else np.asarray(ground_truth_rect, dtype=np.float64)
        self.object_ids = object_ids          # single-object tracking: always None here
        self.multiobj_mode = False

    def init_info(self) -> dict:
        return self.frame_info(0)

    def frame_info(self, frame_num: int) -> dict:
        """Only frame 0 carries initialisation data (data.py:114-124): {'init_bbox': [x, y, w, h]}."""
        if frame_num == 0 and self.ground_truth_rect is not None:
            return {"init_bbox": [float(v) for v in self.ground_truth_rect[0]]}
        return {}

    def __len__(self):
        return len(self.frames)

    def __repr__(self):
        return f"Sequence {self.name}, length={len(self.frames)} frames"


class SequenceList(list):
    """list of Sequence, indexable by name, position, or a list of positions (data.py:150-169)."""

    def __getitem__(self, item):
        if isinstance(item, str):
            for s in self:
                if s.name == item:
                    return s
            raise IndexError("Sequence name not in the dataset.")
        if isinstance(item, (tuple, list)):
            return SequenceList(list.__getitem__(self, i) for i in item)
        r = list.__getitem__(self, item)
        return SequenceList(r) if isinstance(item, slice) else r

    def __add__(self, other):
        return SequenceList(list.__add__(self, other))


def read_image(frame) -> np.ndarray:
    """HxWx3 uint8 RGB (``Tracker._read_image``, evaluation/tracker.py:282-285: cv.imread + BGR->RGB)."""
    if isinstance(frame, np.ndarray):
        return frame
    if frame.endswith(".npy"):
        return np.load(frame)
    try:
        import cv2
        return cv2.cvtColor(cv2.imread(frame), cv2.COLOR_BGR2RGB)
    except ImportError:
        from PIL import Image
        return np.asarray(Image.open(frame).convert("RGB"))


def synthetic_sequence(name: str, n_frames: int, H: int = 240, W: int = 320, seed: int = 0, dataset: str = "synthetic"):
    """A textured target drifting over a fixed random background; ground truth known exactly."""
    rs = np.random.RandomState(seed)
    bg = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    tw, th = 30 + rs.randint(0, 30), 24 + rs.randint(0, 24)
    patch = rs.randint(0, 256, (th, tw, 3)).astype(np.uint8)
    x0, y0 = rs.randint(20, W - tw - 80), rs.randint(20, H - th - 60)
    vx, vy = rs.uniform(0.5, 3.0), rs.uniform(0.3, 2.0)
    frames, gt = [], []
    for t in range(n_frames):
        x, y = int(round(x0 + vx * t)), int(round(y0 + vy * t))
        x, y = min(x, W - tw), min(y, H - th)
        f = bg.copy()
        f[y:y + th, x:x + tw] = patch
        frames.append(f)
        gt.append([x, y, tw, th])
    return Sequence(name, frames, dataset, np.array(gt, dtype=np.float64))


def synthetic_dataset(n_sequences: int = 8, n_frames: int = 20, H: int = 240, W: int = 320, ragged: bool = True):
    """Deterministic synthetic benchmark: sequence s has n_frames + (s % 3) * 2 frames when ragged."""
    return SequenceList(synthetic_sequence(f"synth_{s:03d}", n_frames + ((s % 3) * 2 if ragged else 0), H, W, seed=100 + s)
                        for s in range(n_sequences))


def folder_dataset(root: str, dataset: str = "folder"):
    """<root>/<seq>/(img/)*.{jpg,jpeg,png,bmp,npy} sorted by name + <root>/<seq>/groundtruth*.txt with one
    `x,y,w,h` (comma / tab / space separated) row per frame, at least the first."""
    seqs = []
    for d in sorted(os.listdir(root)):
        p = os.path.join(root, d)
        if not os.path.isdir(p):
            continue
        img_dir = os.path.join(p, "img") if os.path.isdir(os.path.join(p, "img")) else p
        frames = sorted(f for ext in ("jpg", "jpeg", "png", "bmp", "npy") for f in glob.glob(os.path.join(img_dir, "*." + ext)))
        gts = sorted(glob.glob(os.path.join(p, "groundtruth*.txt")))
        if not frames or not gts:
            continue
        rows = [[float(v) for v in re.split(r"[,\s]+", ln.strip())] for ln in open(gts[0]) if ln.strip()]
        seqs.append(Sequence(d, frames, dataset, np.array(rows, dtype=np.float64)))
    return SequenceList(seqs)


def get_dataset(*names):
    """``get_dataset('synthetic')``, ``get_dataset('synthetic:16x50')`` (16 sequences x 50 frames),
    ``get_dataset('folder:/path')`` -- the registry role of lib/test/evaluation/datasets.py:43-48."""
    out = SequenceList()
    for n in names:
        if n.startswith("synthetic"):
            m = re.fullmatch(r"synthetic(?::(\d+)x(\d+))?", n)
            if not m:
                raise ValueError(f"bad synthetic dataset spec {n!r} (want synthetic or synthetic:<sequences>x<frames>)")
            out = out + (synthetic_dataset(int(m.group(1)), int(m.group(2))) if m.group(1) else synthetic_dataset())
        elif n.startswith("folder:"):
            out = out + folder_dataset(n[len("folder:"):])
        else:
            raise ValueError(f"unknown dataset {n!r}: this build ships 'synthetic[:NxT]' and 'folder:<path>' "
                             f"(the reference's benchmark parsers are out of scope)")
    return out
