"""``Tracker``: the wrapper the CLI scripts drive (``lib/test/evaluation/tracker.py:27-290``):
plugin discovery, per-sequence loop with per-frame wall times, headless ``run_video``."""
from __future__ import annotations

import importlib
import os
import time
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .data import read_image
from .environment import env_settings
from .results import format_boxes


def trackerlist(name, parameter_name, dataset_name, run_ids=None, display_name=None):
    run_ids = [run_ids] if run_ids is None or isinstance(run_ids, int) else run_ids
    return [Tracker(name, parameter_name, dataset_name, r, display_name) for r in run_ids]


class Tracker:
    def __init__(self, name: str, parameter_name: str, dataset_name: str, run_id: int = None, display_name: str = None):
        assert run_id is None or isinstance(run_id, int)
        self.name, self.parameter_name, self.dataset_name = name, parameter_name, dataset_name
        self.run_id, self.display_name = run_id, display_name
        env = env_settings()
        # <results_path>/<name>/<parameter_name>[_<run_id:03d>]   (evaluation/tracker.py:44-50)
        leaf = parameter_name if run_id is None else "%s_%03d" % (parameter_name, run_id)
        self.results_dir = "{}/{}/{}".format(env.results_path, name, leaf)
        # plugin discovery: module <name> with get_tracker_class() (evaluation/tracker.py:52-60)
        try:
            self.tracker_class = importlib.import_module(f"vittracker_amd.tracker.{name}").get_tracker_class()
        except ModuleNotFoundError:
            self.tracker_class = None

    def get_parameters(self):
        return importlib.import_module(f"vittracker_amd.parameter.{self.name}").parameters(self.parameter_name)

    def create_tracker(self, params):
        if self.tracker_class is None:
            raise ValueError(f"no tracker plugin named {self.name!r}")
        return self.tracker_class(params, self.dataset_name)

    # ------------------------------------------------------------------ one sequence, one tracker object
    def run_sequence(self, seq, debug=None, params=None):
        params = params or self.get_parameters()
        params.debug = getattr(params, "debug", 0) if debug is None else debug
        return self._track_sequence(self.create_tracker(params), seq, seq.init_info())

    def _track_sequence(self, tracker, seq, init_info):
        """Per-frame outputs as lists: target_bbox[i] = box of frame i (frame 0: the init box), time[i] = wall
        seconds of initialize / track for frame i (evaluation/tracker.py:90-152)."""
        output = {"target_bbox": [], "time": []}
        if tracker.params.save_all_boxes:
            output["all_boxes"], output["all_scores"] = [], []

        def store(tracker_out, defaults):
            for key in output:
                val = tracker_out.get(key, defaults.get(key))
                if key in tracker_out or val is not None:
                    output[key].append(val)

        image = read_image(seq.frames[0])
        t0 = time.time()
        out = tracker.initialize(image, init_info) or {}
        prev = OrderedDict(out)
        store(out, {"target_bbox": init_info.get("init_bbox"), "time": time.time() - t0,
                    "all_boxes": out.get("all_boxes"), "all_scores": out.get("all_scores")})
        for frame_num, frame in enumerate(seq.frames[1:], start=1):
            image = read_image(frame)
            t0 = time.time()
            info = seq.frame_info(frame_num)
            info["previous_output"] = prev
            if seq.ground_truth_rect is not None and len(seq.ground_truth_rect) > 1:
                info["gt_bbox"] = seq.ground_truth_rect[frame_num]
            out = tracker.track(image, info)
            prev = OrderedDict(out)
            store(out, {"time": time.time() - t0})
        for key in ("target_bbox", "all_boxes", "all_scores"):
            if key in output and len(output[key]) <= 1:
                output.pop(key)
        return output

    # ------------------------------------------------------------------ a video
    def run_video(self, videofilepath, optional_box=None, debug=None, visdom_info=None, save_results=False, params=None):
        """Headless counterpart of ``run_video`` (evaluation/tracker.py:154-273): frames from a ``.npy`` array
        (T,H,W,3) or a directory of images (or, with OpenCV present, any file cv.VideoCapture opens), the init
        box from ``optional_box`` [x, y, w, h]; ``track(frame)`` is called WITHOUT info, as the reference does;
        boxes are truncated to int per frame; ``video_<stem>.txt`` is written when save_results is set.
        The display window, ROI selection and key handling of the reference are UI and not reproduced."""
        params = params or self.get_parameters()
        params.debug = getattr(params, "debug", 0) if debug is None else debug
        params.tracker_name, params.param_name = self.name, self.parameter_name
        tracker = self.create_tracker(params)
        if optional_box is None:
            raise ValueError("headless run_video needs optional_box = [x, y, w, h] (no ROI selection window)")
        assert isinstance(optional_box, (list, tuple)) and len(optional_box) == 4, "valid box's format is [x,y,w,h]"
        frames = iter(_video_frames(videofilepath))
        first = next(frames, None)
        if first is None:
            raise SystemExit("Read frame from {} failed.".format(videofilepath))
        tracker.initialize(first, {"init_bbox": list(optional_box)})
        output_boxes = [list(optional_box)]
        for frame in frames:
            out = tracker.track(frame)
            output_boxes.append([int(s) for s in out["target_bbox"]])
        if save_results:
            os.makedirs(self.results_dir, exist_ok=True)
            path = os.path.join(self.results_dir, "video_{}.txt".format(Path(videofilepath).stem))
            with open(path, "w") as f:
                f.write(format_boxes(output_boxes))
        return output_boxes


def _video_frames(path):
    if os.path.isdir(path):
        for f in sorted(os.listdir(path)):
            if f.lower().endswith((".jpg", ".jpeg", ".png", ".bmp", ".npy")):
                yield read_image(os.path.join(path, f))
    elif str(path).endswith(".npy"):
        for fr in np.load(path):
            yield np.ascontiguousarray(fr)
    else:
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError(f"{path}: container formats need OpenCV (absent here); pass a .npy (T,H,W,3) array or a "
                               f"directory of images") from e
        cap = cv2.VideoCapture(path)
        while True:
            ok, fr = cap.read()
            if not ok or fr is None:
                break
            yield fr            # raw BGR, as the reference feeds it (evaluation/tracker.py:190,219-228)
        cap.release()
