#!/usr/bin/env python3
"""End-to-end lock-step tracking of B synthetic sequences (frames uploaded from the host every step):
the real-pipeline number next to bench.py's device-step number.

    python tracking/track_batch_demo.py --batch 256 --frames 240 --size 480 640
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_48_h32_noKD")
    ap.add_argument("--geom", default="", choices=["", "G128", "G256"],
                    help="shorthand for --config: G128 = vit_48_h32_g128 (128 / 64 px, BASELINE's metric), G256 = vit_48_h32_noKD (the shipped YAML)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--size", type=int, nargs=2, default=[480, 640])
    ap.add_argument("--hold-boxes", action="store_true",
                    help="open loop (vt_set_open_loop): every step searches around the sequences' INITIAL boxes, the result goes to the record only: on "
                         "noise frames with random weights free-running boxes drift to the clip limits within a few frames -- 10 px wide at G128, the whole "
                         "frame at G256 -- and the crop then reads a 40 px or a 2200 px window; held boxes keep the 120-360 px windows of a tracker "
                         "that follows a target")
    ap.add_argument("--one-stream", action="store_true", help="stop before the two-shard phase (profiling: per-kernel times of ONE step in flight)")
    ap.add_argument("--device-frames-only", action="store_true",
                    help="skip the host-frames phase (profiling: while the step waits 4 ms for each upload the GPU idles at low clocks, and that phase's "
                         "kernels -- a fifth of a short run's -- read 1.3-1.5 x their busy-chip duration in a kernel trace)")
    a = ap.parse_args()
    if a.geom:
        a.config = {"G128": "vit_48_h32_g128", "G256": "vit_48_h32_noKD"}[a.geom]
    import torch
    os.environ.setdefault("VITTRACK_PRJ_DIR", ROOT)
    from vittracker_amd.batched import BatchedVitTracker
    from vittracker_amd.parameter import vit_dist as P
    p = P.parameters(a.config)
    p.allow_synthetic_weights = True
    H, W = a.size
    rs = np.random.RandomState(0)
    B = a.batch
    frames = rs.randint(0, 256, (2, B, H, W, 3)).astype(np.uint8)      # two alternating frame sets
    boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
    bt = BatchedVitTracker(p, B)
    bt.initialize(frames[0], boxes)
    if a.hold_boxes:
        bt.hold_states(True)
    if not a.device_frames_only:
        for f in range(3):
            bt.track(frames[f & 1], sync=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for f in range(a.frames):
            out = bt.track(frames[f & 1], sync=False)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"{B} sequences x {a.frames} frames of {H}x{W}: {B * a.frames / dt:.0f} frames/s end to end "
              f"(host frames -> H2D -> crop -> graph -> state update), {dt / a.frames * 1e3:.2f} ms per step; "
              f"H2D {B * H * W * 3 / 1e6:.0f} MB per step")
    dev = torch.from_numpy(frames).cuda()
    for f in range(3):       # caller-owned device frames take the eager path: warm it (a graph's first launch includes its upload)
        bt.track(dev[f & 1], sync=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for f in range(a.frames):
        out = bt.track(dev[f & 1], sync=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"frames already on the device: {B * a.frames / dt:.0f} frames/s, {dt / a.frames * 1e3:.3f} ms per step")
    print("last boxes[0]:", out["target_bbox"][0].tolist())
    bb = out["target_bbox"].cpu().numpy()
    side = np.sqrt(bb[:, 2] * bb[:, 3])
    print(f"box sides sqrt(w h) after {bt.frame_id} frames: min {side.min():.1f} median {np.median(side):.1f} max {side.max():.1f} px "
          f"(the crop reads a square of search_factor x that side)")
    for n in (2, 4):      # n frames per graph launch (crop -> forward -> state update, n times, one graph)
        chunk = dev[[i & 1 for i in range(n)]].contiguous()
        for _ in range(3):       # the first replays of a fresh graph include its upload
            bt.track_chunk(chunk, sync=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for f in range(a.frames // n):
            out = bt.track_chunk(chunk, sync=False)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"track_chunk, {n} frames per launch, frames on the device: {B * (a.frames // n) * n / dt:.0f} frames/s, "
              f"{dt / ((a.frames // n) * n) * 1e3:.3f} ms per frame")
    # Two shards of B sequences, each a BatchedVitTracker of its own (own model workspaces, states and graphs) on its own stream,
    # stepped alternately: the trackers run on the CURRENT stream, so two stream contexts are all it takes.  Kernels of the two
    # shards overlap at their tails and nothing is left of the gap between graph launches.
    if a.one_stream:
        return
    bt2 = BatchedVitTracker(p, B)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    trackers = [bt, bt2]
    with torch.cuda.stream(streams[1]):
        bt2.initialize(frames[0], boxes)
    torch.cuda.synchronize()
    n = 4
    chunk = dev[[i & 1 for i in range(n)]].contiguous()
    for _ in range(3):
        for t, st in zip(trackers, streams):
            with torch.cuda.stream(st):
                t.track_chunk(chunk, sync=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for f in range(a.frames // n):
        for t, st in zip(trackers, streams):
            with torch.cuda.stream(st):
                out = t.track_chunk(chunk, sync=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"two shards of {B} sequences on two streams, track_chunk {n} frames per launch: {2 * B * (a.frames // n) * n / dt:.0f} frames/s, "
          f"{dt / (2 * (a.frames // n) * n) * 1e3:.3f} ms per frame of one shard")


if __name__ == "__main__":
    main()
