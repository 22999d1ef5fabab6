#!/usr/bin/env python3
"""Phase stamps of crop_band_kernel (uint8 form) from a -DVT_CROPF_DBG=16 build:
    VITTRACK_LIB=build_variants/cropstamp.so VT_CROP_BYTES=0 python tools/crop_stamps.py [T] [band]
Every workgroup writes s_memtime at: start, geometry done, tables + barrier done, the first two items' loads issued, ... arrived, end (over the first 48 bytes
of its band).  Printed relative to the launch's earliest start, as percentiles over the workgroups, in stamp ticks and us (the counter's
rate is measured against a HIP-event-timed launch)."""
import os
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vittracker_amd import native, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ipt = int(os.environ.get("VT_CROP_BAND", "4"))
ipt = 4 if ipt >= 4 else 2
B, H, W = 256, 480, 640
rs = np.random.RandomState(0)
frames = torch.from_numpy(rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)).cuda()
boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
st = torch.tensor(boxes, dtype=torch.float64).cuda()
m = native.Model(64, 128, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=16, len_x=64))
out = torch.empty(B, T, T, 3, dtype=torch.uint8, device="cuda")
rf = torch.empty(B, dtype=torch.float64, device="cuda")
for _ in range(5):
    m.crop_u8(frames, st, 4.0, T, out=out, resize_factor=rf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
m.crop_u8(frames, st, 4.0, T, out=out, resize_factor=rf)
e1.record()
torch.cuda.synchronize()
ev_us = e0.elapsed_time(e1) * 1000
rows_per_band = ipt * 256 // (T // 4)
raw = out.cpu().numpy().reshape(B, T // rows_per_band, rows_per_band * T * 3)[:, :, :56].copy().view(np.uint64).reshape(-1, 7).astype(np.int64)
t0 = raw[:, 0].min()
hwid = raw[:, 6].copy()
raw = raw[:, :6]
rel = raw - t0
span = rel[:, 5].max()
print(f"T={T} band={ipt}: {raw.shape[0]} workgroups; launch by HIP events {ev_us:.1f} us (eager, incl. launch); stamp span {span} ticks")
names = ["start", "geometry", "tables+barrier", "loads issued", "loads here", "end"]
for i, n in enumerate(names):
    v = rel[:, i]
    print(f"  {n:16s} min {v.min():8d}  p10 {int(np.percentile(v, 10)):8d}  median {int(np.median(v)):8d}  p90 {int(np.percentile(v, 90)):8d}  max {v.max():8d}")
d = np.diff(rel, axis=1)
for i in range(5):
    print(f"  {names[i]:>16s} -> {names[i + 1]:16s} median {int(np.median(d[:, i])):7d}  p90 {int(np.percentile(d[:, i], 90)):7d}")

# the counters of different XCDs are not synchronised: cluster the workgroups by counter base (gaps > 1e6 ticks) and look inside each
order = np.argsort(raw[:, 0])
starts = raw[order, 0]
cuts = np.where(np.diff(starts) > 1_000_000)[0] + 1
for ci, idx in enumerate(np.split(order, cuts)):
    r = raw[idx] - raw[idx, 0].min()
    print(f"  cluster {ci}: {len(idx):5d} workgroups; start p50 {int(np.median(r[:, 0])):7d} p90 {int(np.percentile(r[:, 0], 90)):7d} max {r[:, 0].max():7d}; "
          f"end min {r[:, 5].min():7d} p50 {int(np.median(r[:, 5])):7d} max {r[:, 5].max():7d}")

# HW_ID: [3:0] wave, [5:4] SIMD, [11:8] CU, [12] SH, [15:13] SE (gfx9); XCC_ID [3:0]
cu = ((hwid >> 32) & 0xf) * 4096 + ((hwid >> 13) & 7) * 32 + ((hwid >> 12) & 1) * 16 + ((hwid >> 8) & 0xf)
ids, counts = np.unique(cu, return_counts=True)
print(f"  workgroups ran on {len(ids)} distinct (XCC, SE, SH, CU); workgroups per CU: min {counts.min()} median {int(np.median(counts))} max {counts.max()}; histogram {dict(zip(*np.unique(counts, return_counts=True)))}")
# timeline of the busiest CU and of a median one (same counter): starts and ends
for label, c in (("busiest", ids[np.argmax(counts)]), ("median", ids[np.argsort(counts)[len(ids) // 2]])):
    r = raw[cu == c]
    r = r - r[:, 0].min()
    o = np.argsort(r[:, 0])
    print(f"  {label} CU {c:#x}: " + "  ".join(f"[{r[i, 0]}..{r[i, 5]}]" for i in o[:12]))
