#!/usr/bin/env python3
"""bench.py -- track() device-step throughput of the vit_dist hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (stem -> 3 transformer blocks -> box head -> both bbox
decodes) over one batch of 256 synthetic frames per GPU, replayed as a hipGraph, inputs already
resident in HBM.  Workload = BASELINE.json configs[1]: vit_48_h32, 128 px search / 64 px template
("G128"), batch 256.  With N > 1 every rank steps its own independent shard of sequences
(weak scaling) and the per-step (B,5) results are all-gathered over RCCL, overlapped with the
next step.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline      dominant kernel (the transformer-block kernel): algorithmic FLOP per launch /
                its average duration (HIP events on the launch stream), vs. the 157.3 TFLOP/s
                fp32 MFMA/VALU peak of MI355X_MICROARCH.md.
  cpu_baseline  the torch fp32 restatement of the reference module graph (oracle/, kind "port")
                timed on this box's host cores on a bounded sample.
  stages_us, also   per-stage durations, the G256 (shipped-YAML geometry) throughput and a G128 batch sweep (1/16/64).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "track() frames/sec per GPU, vit_48_h32, 128px search / 64px template"
PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: vector == f32-MFMA peak
GEOMS = {"G128": (64, 128), "G256": (128, 256)}


def macs_per_frame(tz, tx, C=48, depth=3, W=32):
    """Algorithmic MACs of one forward (equals the hook count on the reference)."""
    ch = [3, C // 8, C // 4, C // 2, C]

    def stem(T):
        tot, s = 0, T
        for i in range(4):
            s //= 2
            tot += s * s * ch[i] * ch[i + 1] * 9
        return tot
    lz, lx = (tz // 16) ** 2, (tx // 16) ** 2
    L = lz + lx
    blocks = depth * (L * 12 * C * C + 2 * L * L * C)
    head = lx * (3 * 9 * (C * W + W * W // 2 + W * W // 8 + W * W // 32) + 5 * W // 8)
    return {"stem": stem(tz) + stem(tx), "blocks": blocks, "head": head}


class Runner:
    def __init__(self, geom, B, seed=0):
        import torch
        from vittracker_amd import native, synth
        self.torch, self.native = torch, native
        self.tz, self.tx = GEOMS[geom]
        self.B = B
        lz, lx = (self.tz // 16) ** 2, (self.tx // 16) ** 2
        self.model = native.Model(self.tz, self.tx, max_batch=B)
        self.model.load_state_dict(synth.synth_state_dict(0, len_z=lz, len_x=lx))
        z, x = synth.synth_inputs(seed, B, self.tz, self.tx)
        self.z, self.x = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
        self.out = native.Outputs(B, self.model.feat_sz, "cuda")
        self.graph, _ = self.model.capture(self.z, self.x, self.out)
        self.stream = torch.cuda.Stream()

    def time_us(self, fn, iters, warm=5):
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            for _ in range(warm):
                fn()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    def stage_times(self, iters):
        m, nat, s = self.model, self.native, self.stream
        tok = m.stem(self.z, self.x)
        feat = m.blocks(tok)
        self.torch.cuda.synchronize()
        L = nat.lib()
        t_stem = self.time_us(lambda: nat._check(L.vt_stem(m._h, nat._ptr(self.z), nat._ptr(self.x), self.B, nat._stream(s),
                                                           nat._ptr(tok)), "vt_stem"), iters)
        t_blocks = self.time_us(lambda: m.blocks(tok, stream=s, feat=feat), iters)
        t_head = self.time_us(lambda: m.head(feat, self.out, stream=s), iters)
        return {"stem": t_stem, "blocks": t_blocks, "head": t_head}


def pmc_traffic(geom, B):
    """HBM bytes per launch of the block kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, in bytes,
    measured at this geometry and batch).  None when no matching measurement is committed."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = t.get(f"{geom}_B{B}", {}).get("vtb::blocks_kernel")
        return None if e is None else int(e["hbm_bytes_per_launch"])
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(budget_s=20.0):
    """Reference-equivalent CPU path: the torch restatement of the module graph (oracle/), timed
    like tracking/profile_model_cpu.py:36-49 (bs=1 loop) and at bs=256, bounded to ~budget_s."""
    import torch
    from oracle import vt_oracle_torch as ot
    from vittracker_amd import synth
    tz, tx = GEOMS["G128"]
    sd = synth.synth_state_dict(0, len_z=16, len_x=64)
    m = ot.build_from_state(sd)
    res = {}
    with torch.no_grad():
        for bs, warm in ((1, 30), (256, 1)):
            z, x = synth.synth_inputs(0, bs, tz, tx)
            zt, xt = torch.from_numpy(z), torch.from_numpy(x)
            for _ in range(warm):
                m(zt, xt)
            n, t0 = 0, time.time()
            while time.time() - t0 < budget_s / 2 and n < (1000 if bs == 1 else 40):
                m(zt, xt)
                n += 1
            dt = (time.time() - t0) / n
            res[bs] = (bs / dt, n)
    best = max(res, key=lambda k: res[k][0])
    return {"value": round(res[best][0], 1), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"torch fp32 module restatement (oracle/vt_oracle_torch.py), G128, best of bs=1 x{res[1][1]} "
                      f"({res[1][0]:.0f} fps) and bs=256 x{res[256][1]} batches ({res[256][0]:.0f} fps), "
                      f"os.cpu_count()={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--geom", default="G128", choices=list(GEOMS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip per-stage timings and the G256 line")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    B = a.batch
    r = Runner(a.geom, B, seed=rank)   # each rank: its own shard of independent sequences
    s = r.stream
    # per-step result record (B,5) = hann box + confidence, double-buffered for the async gather
    res = [torch.empty(B, 5, device="cuda") for _ in range(2)]
    gathered = [torch.empty(world * B, 5, device="cuda") for _ in range(2)] if world > 1 else None
    pending = [None, None]

    def step(i):
        with torch.cuda.stream(s):
            r.graph.launch(s)
            if world > 1:
                k = i & 1
                if pending[k] is not None:
                    pending[k].wait()          # slot free again (gather of step i-2 done)
                res[k][:, :4].copy_(r.out.hann_boxes)
                res[k][:, 4].copy_(r.out.conf)
                pending[k] = dist.all_gather_into_tensor(gathered[k], res[k], async_op=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    if world > 1:
        with torch.cuda.stream(s):
            for p in pending:
                if p is not None:
                    p.wait()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = world * B * a.steps / elapsed
        macs = macs_per_frame(*GEOMS[a.geom])
        line = {
            "metric": METRIC, "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"vit_48_h32 {a.geom} ({GEOMS[a.geom][1]}px search / {GEOMS[a.geom][0]}px template), "
                                   f"batch {B} per GPU, hipGraph replay, N(0,1) crops, seeded synthetic weights",
                       "batch_per_gpu": B, "global_batch": world * B, "geometry": a.geom,
                       "parallelism": f"{world} independent sequence shards" + (", RCCL all_gather of (B,5) results" if world > 1 else "")},
            "frac_fp32_peak_whole_step": round(value / world * 2 * sum(macs.values()) / 1e12 / PEAK_FP32_TFLOPS, 4),
        }
        if not a.no_extra:
            st = r.stage_times(max(20, a.steps // 4))
            flop = 2 * macs["blocks"] * B
            ach = flop / (st["blocks"] * 1e-6) / 1e12
            line["roofline"] = {"kernel": "vtb::blocks_kernel", "bound": "mfma", "achieved": round(ach, 2),
                                "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                                "traffic": pmc_traffic(a.geom, B), "flop_per_launch": flop,
                                "avg_launch_us": round(st["blocks"], 2)}
            try:   # the clock the fraction was measured at (dense f32-MFMA probe, 1 wave / SIMD)
                mhz, cpm, _ = r.native.probe_clock(20000, 1)
                line["roofline"]["probe_clock_mhz"] = round(mhz)
                line["roofline"]["probe_cycles_per_mfma"] = round(cpm, 2)
            except Exception:  # noqa: BLE001
                pass
            line["stages_us"] = {k: round(v, 2) for k, v in st.items()}
            line["stages_frac_fp32_peak"] = {k: round(2 * macs[k] * B / (st[k] * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4) for k in st}
            if a.geom == "G128" and world == 1:
                r2 = Runner("G256", B)
                t256 = r2.time_us(lambda: r2.graph.launch(r2.stream), max(20, a.steps // 4))
                m256 = macs_per_frame(*GEOMS["G256"])
                line["also"] = {"G256_frames_per_s": round(B / t256 * 1e6, 1),
                                "G256_frac_fp32_peak": round(B / t256 * 1e6 * 2 * sum(m256.values()) / 1e12 / PEAK_FP32_TFLOPS, 4),
                                "G256_note": "shipped YAML geometry 256/128, 320 tokens, batch %d" % B}
                del r2
                sweep = {}     # smaller batches (one workgroup per frame: latency-bound below 256 frames)
                for bs in (1, 16, 64):
                    rb = Runner(a.geom, bs)
                    t = rb.time_us(lambda: rb.graph.launch(rb.stream), max(50, a.steps // 2))
                    sweep[str(bs)] = {"us_per_step": round(t, 2), "frames_per_s": round(bs / t * 1e6, 1)}
                    del rb
                line["also"]["G128_batch_sweep"] = sweep
        if not a.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
