"""Dataset runners (``lib/test/evaluation/running.py:105-187``).

Three modes:
  sequential   one tracker object per sequence, in this process             (threads = 0 in the reference)
  parallel     a spawn-ed process pool, worker w bound to GPU ``w % num_gpus`` (threads > 0 in the reference)
  batched      MI355X-native: B sequences per GPU advance in lock-step through ``BatchedVitTracker`` (device crop ->
               hipGraph -> device state update, no per-frame host sync); with several GPUs, sequence s runs on rank
               ``s % world`` -- the same rule as the reference's worker -> GPU map.
All three write the reference's result files (results.py)."""
from __future__ import annotations

import multiprocessing
import sys
import time
from datetime import timedelta
from itertools import product

import numpy as np

from .data import read_image
from .results import results_exist, save_tracker_output


def run_sequence(seq, tracker, debug=False, num_gpu=8):
    try:   # worker -> GPU (running.py:105-112)
        import torch
        name = multiprocessing.current_process().name
        worker_id = int(name[name.find("-") + 1:]) - 1
        torch.cuda.set_device(worker_id % num_gpu)
    except Exception:  # noqa: BLE001  (main process: name has no index)
        pass
    if results_exist(tracker.results_dir, seq) and not debug:
        print("FPS: {}".format(-1))
        return None
    print("Tracker: {} {} {} ,  Sequence: {}".format(tracker.name, tracker.parameter_name, tracker.run_id, seq.name))
    if debug:
        output = tracker.run_sequence(seq, debug=debug)
    else:
        try:
            output = tracker.run_sequence(seq, debug=debug)
        except Exception as e:  # noqa: BLE001  (the reference prints and skips the sequence, running.py:138-142)
            print(e)
            return None
    sys.stdout.flush()
    print("FPS: {}".format(len(output["time"]) / sum(output["time"])))
    if not debug:
        save_tracker_output(seq, tracker.results_dir, output)
    return output


def run_dataset(dataset, trackers, debug=False, threads=0, num_gpus=8):
    multiprocessing.set_start_method("spawn", force=True)
    print("Evaluating {:4d} trackers on {:5d} sequences".format(len(trackers), len(dataset)))
    t0 = time.time()
    if threads == 0:
        for seq in dataset:
            for tr in trackers:
                run_sequence(seq, tr, debug=debug)
    else:
        work = [(seq, tr, debug, num_gpus) for seq, tr in product(dataset, trackers)]
        with multiprocessing.Pool(processes=threads) as pool:
            pool.starmap(run_sequence, work)
    print("Done, total time: {}".format(str(timedelta(seconds=(time.time() - t0)))))


# ----------------------------------------------------------------------------------- lock-step batches
def _groups(dataset, batch):
    """Sequences that can share one (B,H,W,3) frame tensor: same frame size, at most `batch` per group."""
    by_hw = {}
    for s in dataset:
        by_hw.setdefault(read_image(s.frames[0]).shape[:2], []).append(s)
    for hw, seqs in by_hw.items():
        for i in range(0, len(seqs), batch):
            yield hw, seqs[i:i + batch]


def run_dataset_batched(dataset, tracker, batch=256, rank=0, world=1, params=None, make_batched=None, frames_per_launch=1, shards=1):
    """Lock-step batched run of `tracker` (an evaluation.Tracker) over `dataset`.  Sequence s belongs to rank
    s % world.  Ragged lengths: a finished sequence keeps receiving its last frame (its extra outputs are dropped).
    Per-frame time written for a sequence = wall time of the lock-step step / live sequences in that step (amortised:
    there is no per-sequence call to time).  frames_per_launch > 1: that many frames are read ahead and tracked by one
    graph launch (BatchedVitTracker.track_chunk; same boxes, the files come out identical).  shards > 1: a group's sequences
    are stepped as that many independent sub-groups on their own streams (ShardedBatchedTracker; same boxes, identical files).
    Returns {seq.name: output dict} for this rank's sequences."""
    from ..parallel import shard_sequences
    mine = [dataset[i] for i in shard_sequences(len(dataset), rank, world)]
    todo = [s for s in mine if not results_exist(tracker.results_dir, s)]
    params = params or tracker.get_parameters()
    params.debug = 0
    if make_batched is None:
        from ..batched import BatchedVitTracker, ShardedBatchedTracker
        make_batched = BatchedVitTracker if int(shards) <= 1 else (lambda p, B: ShardedBatchedTracker(p, B, int(shards)))
    # A sequence whose init box is too small to crop makes `sample_target` raise in the reference (processing_utils.py:33-34),
    # and `run_sequence` prints the error and skips THAT sequence (running.py:138-142): screened out here, per sequence, before
    # the lock-step groups are formed, so one bad box does not cost its whole group.
    import math
    ok = []
    for s in todo:
        try:
            x, y, w, h = [float(v) for v in s.init_info()["init_bbox"]]
            for f in (params.template_factor, params.search_factor):
                if not math.ceil(math.sqrt(w * h) * f) >= 1:
                    raise Exception("Too small bounding box.")
            ok.append(s)
        except Exception as e:  # noqa: BLE001
            print("Tracker: {} {} {} ,  Sequence: {}".format(tracker.name, tracker.parameter_name, tracker.run_id, s.name))
            print(e)
    outputs = {}
    pipelines = {}       # batch size -> BatchedVitTracker: weights, workspaces and captured graphs are built once per size
    for (H, W), seqs in _groups(ok, batch):
        try:
            outputs.update(_run_group(seqs, tracker, params, make_batched, pipelines, frames_per_launch))
        except Exception as e:  # noqa: BLE001  (a failing group is reported and skipped; finished groups are already on disk)
            print("Tracker: {} {} {} ,  Sequences: {}".format(tracker.name, tracker.parameter_name, tracker.run_id, [s.name for s in seqs]))
            print(e)
            pipelines.clear()        # a pipeline that raised mid-run is not reused
    return outputs


def _run_group(seqs, tracker, params, make_batched, pipelines, frames_per_launch):
    """One lock-step group (same frame size, <= batch sequences); results are written when the group is done."""
    B = len(seqs)
    bt = pipelines.get(B)
    if bt is None:
        bt = pipelines[B] = make_batched(params, B)
    T = max(len(s) for s in seqs)
    frame = lambda s, t: read_image(s.frames[min(t, len(s) - 1)])  # noqa: E731
    out = {s.name: {"target_bbox": [list(s.init_info()["init_bbox"])], "time": []} for s in seqs}
    t0 = time.time()
    bt.initialize(np.stack([frame(s, 0) for s in seqs]), [s.init_info()["init_bbox"] for s in seqs])
    dt = (time.time() - t0) / B
    for s in seqs:
        out[s.name]["time"].append(dt)
    n = max(1, int(frames_per_launch))
    t = 1
    while t < T:
        k = n if (n > 1 and t + n <= T and hasattr(bt, "track_chunk")) else 1      # whole chunks, then frame by frame
        t0 = time.time()
        if k == 1:
            boxes = bt.track(np.stack([frame(s, t) for s in seqs]))["target_bbox"].numpy()[None]     # sync=True: boxes on the host
        else:
            boxes = bt.track_chunk(np.stack([np.stack([frame(s, t + j) for s in seqs]) for j in range(k)]))["target_bbox"].numpy()
        wall = (time.time() - t0) / k
        for j in range(k):
            live = sum(1 for s in seqs if t + j < len(s))
            for b, s in enumerate(seqs):
                if t + j < len(s):
                    out[s.name]["target_bbox"].append(boxes[j, b].tolist())
                    out[s.name]["time"].append(wall / live)
        t += k
    for s in seqs:
        save_tracker_output(s, tracker.results_dir, out[s.name])
        print("Tracker: {} {} {} ,  Sequence: {}  FPS: {}".format(tracker.name, tracker.parameter_name, tracker.run_id, s.name,
                                                                 len(out[s.name]["time"]) / sum(out[s.name]["time"])))
    return out
