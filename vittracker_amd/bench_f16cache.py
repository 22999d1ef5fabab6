"""bench.py --config vit48_f16cache: BASELINE config 5 -- vit_48_h32 with f16 contractions and the exact template cache
over a 1000-frame synthetic sequence on 1 x MI355X.

B = 256 independent sequences advance in lock-step for 1000 frames: the template crops are fixed (cached once with
vt_set_template), the search crops are fresh every frame -- a ring of 8 distinct input batches (8 x 50 MB at G128, more
than the 256 MB Infinity Cache) captured as one graph of 8 consecutive frames, replayed back to back.  `value` counts
frames (sequences x frames) per second over the 1000-frame run."""
from __future__ import annotations

import json
import os
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PEAK_F16_TFLOPS = 2500.0
METRIC = "track() frames/sec per GPU, vit_48_h32, 128px search / 64px template"
GEOMS = {"G128": (64, 128), "G256": (128, 256)}
RING = 8


class Seq:
    def __init__(self, geom, B, precision, cached, ring=RING):
        import torch
        from vittracker_amd import native, synth
        self.torch = torch
        tz, tx = GEOMS[geom]
        self.m = native.Model(tz, tx, max_batch=B, precision=precision)
        self.m.load_state_dict(synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
        z, _ = synth.synth_inputs(0, B, tz, tx)
        self.z = torch.from_numpy(z).cuda()
        g = torch.Generator(device="cuda").manual_seed(1234)
        self.xs = [torch.randn(B, 3, tx, tx, device="cuda", generator=g) for _ in range(ring)]
        self.out = native.Outputs(B, self.m.feat_sz, "cuda")
        if cached:
            self.m.set_template(self.z)
        # one graph per ring slot, and the whole ring as ONE graph of `ring` consecutive frames (vt_graph_capture_steps): the gap
        # the runtime leaves between two graph launches is then paid once per `ring` frames
        self.graphs = [self.m.capture(None if cached else self.z, x, self.out)[0] for x in self.xs]
        self.ring_graph = self.m.capture_steps(None if cached else [self.z] * ring, self.xs, [self.out] * ring)[0]
        self.s = torch.cuda.Stream()
        self.B = B

    def run(self, frames):
        """EXACTLY `frames` frames: whole rings through the ring graph, the remainder one slot at a time."""
        torch = self.torch
        n = len(self.graphs)
        with torch.cuda.stream(self.s):
            for _ in range(frames // n):
                self.ring_graph.launch(self.s)
            for f in range(frames % n):
                self.graphs[f].launch(self.s)

    def timed(self, frames, warm):
        self.run(warm)
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run(frames)
        self.torch.cuda.synchronize()
        return time.perf_counter() - t0

    def check_cache_exact(self):
        """At the TIMED batch (the large-batch cached kernel forms): the cached graph of ring slot 0 == the uncached eager
        step on the same crops, bit for bit."""
        torch = self.torch
        ref = self.m.forward(self.z, self.xs[0])
        ref = {k: getattr(ref, k).clone() for k in ("score_map", "size_map", "offset_map", "hann_boxes", "conf")}
        self.graphs[0].launch(self.s)
        self.s.synchronize()
        bad = [k for k, v in ref.items() if not torch.equal(getattr(self.out, k), v)]
        if bad:
            raise SystemExit(f"bench.py --config vit48_f16cache: cached step differs from the uncached step at B={self.B}: {bad}")
        return True

    def close(self):
        self.graphs = self.ring_graph = None
        self.m.close()


def check_against_golden(geom):
    """The f16 + cached configuration on a reference fixture's inputs (tolerances of tests/test_gpu_f16cache.py)."""
    import torch
    from vittracker_amd import native, synth
    tz, tx = GEOMS[geom]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"ref_{geom}_s0_b4.npz"))
    m = native.Model(tz, tx, max_batch=4, precision="f16")
    m.load_state_dict(synth.synth_state_dict(0, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2))
    z, x = synth.synth_inputs(0, 4, tz, tx)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    m.set_template(zd)
    graph, out = m.capture(None, xd)
    graph.launch()
    torch.cuda.synchronize()
    errs = {k: float(np.abs(getattr(out, k).cpu().numpy() - g[k]).max()) for k in ("score_map", "size_map", "offset_map")}
    errs["hann_boxes"] = float(np.abs(out.hann_boxes.cpu().numpy() - g["hann_boxes"]).max())
    if not (max(errs["score_map"], errs["size_map"], errs["offset_map"]) < 6e-3 and errs["hann_boxes"] < 2e-3):
        raise SystemExit(f"bench.py --config vit48_f16cache disagrees with the reference fixture: {errs}")
    graph = None
    m.close()
    return {"fixture": f"ref_{geom}_s0_b4.npz", "frames": 4, "max_abs_err": {k: float(f"{v:.2e}") for k, v in errs.items()},
            "tolerance": {"maps": 6e-3, "boxes": 2e-3}}


def run(a):
    print(json.dumps(measure(a)), flush=True)


def measure(a, variants=("f16_uncached", "f32_cached", "f32_uncached")):
    """One timed 1000-frame run; returns the JSON line as a dict (bench.py's default run quotes it under `also`)."""
    import torch
    import bench as B0
    if a.gpus != 1:
        raise SystemExit("--config vit48_f16cache is a single-GPU configuration (BASELINE config 5)")
    torch.cuda.set_device(0)
    B, frames = a.batch, 1000
    geom = a.geom
    checked = check_against_golden(geom)
    s = Seq(geom, B, "f16", cached=True)
    checked["cache_exact_at_timed_batch"] = s.check_cache_exact()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.35:
        s.run(40); torch.cuda.synchronize()
    elapsed = s.timed(frames, a.warmup)
    # bench.py --streams 2 (the default): a second shard of B sequences with its own model workspaces, ring graph and stream; the two
    # shards' ring graphs are launched alternately, `frames` frames of each shard's sequences
    nshard, single = 1, None
    if int(getattr(a, "streams", 1)) > 1 and frames % RING == 0:
        s2 = Seq(geom, B, "f16", cached=True)
        s2.check_cache_exact()

        def both(nfr):
            for _ in range(nfr // RING):
                s.ring_graph.launch(s.s)
                s2.ring_graph.launch(s2.s)
        both(max(RING, (a.warmup // RING) * RING))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        both(frames)
        torch.cuda.synchronize()
        single = B * frames / elapsed
        elapsed = (time.perf_counter() - t0) / 2          # per `frames` frames of ONE shard
        nshard = 2
        s2.close()
    s.close()
    macs = B0.macs_per_frame(*GEOMS[geom])
    value = B * frames / elapsed
    line = {"metric": METRIC, "value": round(value, 1), "unit": "frames/s", "n_gpus": 1, "steps": frames, "warmup": a.warmup,
            "ms_per_step": round(elapsed / frames * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"vit_48_h32 {geom}, f16 contractions (f32 accumulate / LayerNorm / softmax / residual), exact template "
                                   f"cache (template tokens + block-0 q,k,v of the template rows), {B} sequences x {frames} frames, fixed "
                                   f"template, fresh search crop per frame (ring of {RING} distinct batches = {RING} consecutive frames per hipGraph launch, back to back)",
                       "batch_per_gpu": B, "global_batch": B, "geometry": geom, "frames_per_sequence": frames, "parallelism": "1 GPU",
                       "streams": nshard, "sequences_per_gpu": B * nshard,
                       "switches": B0.active_switches()},
            "checked": True, "check": checked,
            "frac_f16_peak_whole_step": round(value * 2 * sum(macs.values()) / 1e12 / PEAK_F16_TFLOPS, 5)}
    if single is not None:
        line["single_stream_frames_per_s"] = round(single, 1)
    if not a.no_extra:
        also = {}
        for name, prec, cached in (("f16_uncached", "f16", False), ("f32_cached", "f32", True), ("f32_uncached", "f32", False)):
            if name not in variants:
                continue
            q = Seq(geom, B, prec, cached)
            e = q.timed(frames // 2, 20)
            also[name + "_frames_per_s"] = round(B * (frames // 2) / e, 1)
            q.close()
        if "f16_uncached_frames_per_s" in also:
            also["cache_saving"] = round(1.0 - also["f16_uncached_frames_per_s"] / (single or value), 4)      # one stream against one stream
        if "f32_cached_frames_per_s" in also:
            also["f16_over_f32_cached"] = round((single or value) / also["f32_cached_frames_per_s"], 3)
        lz, L = (GEOMS[geom][0] // 16) ** 2, (GEOMS[geom][0] // 16) ** 2 + (GEOMS[geom][1] // 16) ** 2
        also["cached_macs_note"] = f"cached per frame: stem(z) + block-0 qkv of {lz} template rows of {L} tokens (SURVEY section 5: ~3.7 % of MACs at G256)"
        line["also"] = also
        # dominant kernel under f16: time the three stages (cached step: the stem sees only the search crop)
        q = Seq(geom, B, "f16", cached=True, ring=1)
        m, st = q.m, q.s
        tok = m.stem(q.z, q.xs[0])
        feat = m.blocks(tok)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(5):
                m.blocks(tok, stream=st, feat=feat)
            e0.record()
            for _ in range(50):
                m.blocks(tok, stream=st, feat=feat)
            e1.record()
        e1.synchronize()
        t_blocks = e0.elapsed_time(e1) * 1e3 / 50
        flop = 2 * macs["blocks"] * B
        ach = flop / (t_blocks * 1e-6) / 1e12
        line["roofline"] = {"kernel": "vtb::blocks_kernel (f16 build)", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 5), "traffic": None, "flop_per_launch": flop,
                            "avg_launch_us": round(t_blocks, 2),
                            "note": "with f16 MFMA the contractions are ~1/16 of their fp32 issue time; LayerNorm / softmax / GELU "
                                    "(f32 VALU) and LDS traffic bound this kernel, so the MFMA fraction is small by construction"}
        q.close()
    return line
