"""Result files of the harness (``_save_tracker_output``, ``lib/test/evaluation/running.py:14-102``).

Formats, byte for byte what ``np.savetxt`` writes there:
  ``<seq>.txt``            one row per frame, the box truncated toward zero to int (``astype(int)``),
                           tab separated, ``%d``
  ``<seq>_time.txt``       one ``%f`` per frame
  ``<seq>_all_boxes.txt``  like ``<seq>.txt``; ``<seq>_all_scores.txt``: ``%.2f``
For the datasets 'trackingnet' and 'got10k' the files go one directory deeper (``<results>/<dataset>/``)."""
from __future__ import annotations

import os

import numpy as np


def format_boxes(rows) -> str:
    a = np.asarray(rows, dtype=np.float64)
    a = a.reshape(len(a), -1).astype(np.int64)      # np.array(data).astype(int): truncation toward zero
    return "".join("\t".join("%d" % v for v in r) + "\n" for r in a)


def format_floats(rows, fmt="%f") -> str:
    a = np.asarray(rows, dtype=np.float64)
    a = a.reshape(len(a), -1)
    return "".join("\t".join(fmt % v for v in r) + "\n" for r in a)


def base_results_path(results_dir: str, seq) -> str:
    if seq.dataset in ("trackingnet", "got10k"):
        return os.path.join(results_dir, seq.dataset, seq.name)
    return os.path.join(results_dir, seq.name)


def results_exist(results_dir: str, seq) -> bool:
    return os.path.isfile(base_results_path(results_dir, seq) + ".txt")


def save_tracker_output(seq, results_dir: str, output: dict):
    base = base_results_path(results_dir, seq)
    os.makedirs(os.path.dirname(base), exist_ok=True)
    writers = {"target_bbox": (".txt", format_boxes), "all_boxes": ("_all_boxes.txt", format_boxes),
               "all_scores": ("_all_scores.txt", lambda d: format_floats(d, "%.2f")), "time": ("_time.txt", format_floats)}
    written = []
    for key, data in output.items():
        if key not in writers or data is None or len(data) == 0:
            continue
        suffix, fmt = writers[key]
        with open(base + suffix, "w") as f:
            f.write(fmt(data))
        written.append(base + suffix)
    return written
