#!/usr/bin/env python3
"""Counterpart of the reference's ``tracking/test.py`` (``:14-56``): run a tracker on a dataset.

    python tracking/test.py vit_dist vit_48_h32_noKD --dataset_name synthetic:16x50 --threads 0 --num_gpus 1
    python tracking/test.py vit_dist vit_48_h32_noKD --dataset_name synthetic:512x100 --batch 256      # lock-step batches

Same positional arguments and options as the reference, plus ``--batch B`` (> 0 selects the MI355X-native lock-step
batched runner) and ``--synthetic_weights`` (no trained checkpoint exists in the reference tree).  Result files:
``<save_dir>/test/tracking_results/<tracker>/<param>[_<runid>]/<seq>.txt`` and ``<seq>_time.txt``."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def run_tracker(tracker_name, tracker_param, run_id=None, dataset_name="synthetic", sequence=None, debug=0, threads=0,
                num_gpus=8, batch=0, synthetic_weights=False, frames_per_launch=1, shards=1):
    from vittracker_amd.evaluation import Tracker, get_dataset
    from vittracker_amd.evaluation.running import run_dataset, run_dataset_batched
    dataset = get_dataset(dataset_name)
    if sequence is not None:
        dataset = type(dataset)([dataset[sequence]])
    tracker = Tracker(tracker_name, tracker_param, dataset_name, run_id)
    if synthetic_weights:
        _get = tracker.get_parameters

        def get_parameters():
            p = _get()
            p.allow_synthetic_weights = True
            return p
        tracker.get_parameters = get_parameters
    if batch > 0:
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            import torch
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        run_dataset_batched(dataset, tracker, batch=batch, rank=rank, world=world, frames_per_launch=frames_per_launch, shards=shards)
    else:
        run_dataset(dataset, [tracker], debug, threads, num_gpus=num_gpus)


def main():
    p = argparse.ArgumentParser(description="Run tracker on sequence or dataset.")
    p.add_argument("tracker_name", type=str, help="Name of tracking method.")
    p.add_argument("tracker_param", type=str, help="Name of config file.")
    p.add_argument("--runid", type=int, default=None, help="The run id.")
    p.add_argument("--dataset_name", type=str, default="synthetic", help="synthetic[:NxT] or folder:<path>")
    p.add_argument("--sequence", type=str, default=None, help="Sequence number or name.")
    p.add_argument("--debug", type=int, default=0, help="Debug level.")
    p.add_argument("--threads", type=int, default=0, help="Number of worker processes (0 = sequential).")
    p.add_argument("--num_gpus", type=int, default=8)
    p.add_argument("--batch", type=int, default=0, help="> 0: lock-step batches of this many sequences per GPU")
    p.add_argument("--shards", type=int, default=1,
                   help="with --batch: step a group's sequences as this many independent sub-groups on their own HIP streams (ShardedBatchedTracker)")
    p.add_argument("--frames_per_launch", type=int, default=1,
                   help="with --batch: frames read ahead and tracked per graph launch (BatchedVitTracker.track_chunk)")
    p.add_argument("--synthetic_weights", action="store_true", help="run on the seeded synthetic weights")
    a = p.parse_args()
    try:
        seq = int(a.sequence)
    except (TypeError, ValueError):
        seq = a.sequence
    run_tracker(a.tracker_name, a.tracker_param, a.runid, a.dataset_name, seq, a.debug, a.threads, a.num_gpus, a.batch,
                a.synthetic_weights, a.frames_per_launch, a.shards)


if __name__ == "__main__":
    main()
