#!/usr/bin/env python3
"""Counterpart of the reference's ``tracking/video_demo.py`` (``:14-42``), headless.

    python tracking/video_demo.py vit_dist vit_48_h32_noKD clip.npy --optional_box 100 80 50 40 --save_results

``videofile``: a ``.npy`` (T,H,W,3) uint8 array, a directory of images, or (with OpenCV installed) any video file.
The reference's window / ROI selection / key loop is UI and not reproduced; ``--optional_box`` is therefore required."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def run_video(tracker_name, tracker_param, videofile, optional_box=None, debug=None, save_results=False,
              synthetic_weights=False):
    from vittracker_amd.evaluation import Tracker
    tracker = Tracker(tracker_name, tracker_param, "video")
    params = tracker.get_parameters()
    params.allow_synthetic_weights = synthetic_weights
    return tracker.run_video(videofilepath=videofile, optional_box=optional_box, debug=debug, save_results=save_results,
                             params=params)


def main():
    p = argparse.ArgumentParser(description="Run the tracker on a video.")
    p.add_argument("tracker_name", type=str)
    p.add_argument("tracker_param", type=str)
    p.add_argument("videofile", type=str)
    p.add_argument("--optional_box", type=float, default=None, nargs="+", help="optional_box with format x y w h.")
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--save_results", dest="save_results", action="store_true")
    p.add_argument("--synthetic_weights", action="store_true")
    a = p.parse_args()
    boxes = run_video(a.tracker_name, a.tracker_param, a.videofile, a.optional_box, a.debug, a.save_results, a.synthetic_weights)
    print("tracked %d frames; last box %s" % (len(boxes), boxes[-1]))


if __name__ == "__main__":
    main()
