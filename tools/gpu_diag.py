#!/usr/bin/env python3
"""Per-stage error table and quick timings on the GPU box (development aid, not a test).

    python tools/gpu_diag.py [--geom G128] [--B 256]

Prints max |HIP - oracle| per stage (each stage fed the oracle's upstream activation), then the
eager / graph step time and per-stage times.  Never asserts, so one run shows everything.
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
    ap.add_argument("--geom", default="G128")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--check-B", type=int, default=4)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    import torch
    from oracle import vt_oracle_np as onp
    from vittracker_amd import native, synth
    tz, tx = {"G128": (64, 128), "G256": (128, 256)}[a.geom]
    lz, lx = (tz // 16) ** 2, (tx // 16) ** 2
    print("device:", torch.cuda.get_device_name(0), "|", native.lib().vt_version().decode())
    try:
        native.selftest_mfma()
        print("mfma lane-map selftest: ok")
    except Exception as e:  # noqa: BLE001
        print("mfma lane-map selftest FAILED:", e)

    sd = synth.synth_state_dict(0, len_z=lz, len_x=lx)
    z, x = synth.synth_inputs(0, a.check_B, tz, tx)
    ref = onp.forward(sd, z, x, want_acts=True)
    acts = ref["acts"]
    m = native.Model(tz, tx, max_batch=max(a.B, a.check_B))
    m.load_state_dict(sd)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()  # noqa: E731
    err = lambda got, want: float(np.abs(got.cpu().numpy() - want).max())  # noqa: E731

    print(f"--- stage errors ({a.geom}, B={a.check_B}) max|hip - oracle|")
    tok = m.stem(dev(z), dev(x))
    print(f"stem tokens        {err(tok, acts['tokens']):.3e}   (|ref| max {np.abs(acts['tokens']).max():.2f})")
    rt = dev(acts["tokens"])
    for nb in (1, 2, 3):
        feat, resid = m.blocks(rt, nblocks=nb, want_resid=True)
        print(f"resid after blk {nb - 1}  {err(resid, acts[f'block{nb - 1}']):.3e}   (|ref| max {np.abs(acts[f'block{nb - 1}']).max():.2f})")
    print(f"final norm (x)     {err(feat, acts['norm'][:, -lx:]):.3e}")
    out = m.head(dev(acts["norm"][:, -lx:]))
    for k in ("score_map", "size_map", "offset_map"):
        print(f"head {k:11s}   {err(getattr(out, k), ref[k]):.3e}")
    full = m.forward(dev(z), dev(x))
    for k in ("score_map", "size_map", "offset_map", "hann_boxes", "conf"):
        print(f"e2e  {k:11s}   {err(getattr(full, k), ref[k]):.3e}")
    print(f"e2e  pred_boxes    {err(full.pred_boxes, ref['pred_boxes'][:, 0]):.3e}")

    # ---- timings
    B = a.B
    z, x = synth.synth_inputs(1, B, tz, tx)
    zd, xd = dev(z), dev(x)
    s = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn, n=a.iters):
        with torch.cuda.stream(s):
            for _ in range(5):
                fn()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3   # us

    outb = native.Outputs(B, m.feat_sz, "cuda")
    tok = m.stem(zd, xd)
    feat = m.blocks(tok)
    torch.cuda.synchronize()
    t_stem = timeit(lambda: native._check(native.lib().vt_stem(m._h, native._ptr(zd), native._ptr(xd), B, native._stream(s), native._ptr(tok)), "stem"))
    t_blk = timeit(lambda: m.blocks(tok, stream=s, feat=feat))
    t_head = timeit(lambda: m.head(feat, outb, stream=s))
    t_eager = timeit(lambda: m.forward(zd, xd, outb, stream=s))
    graph, _ = m.capture(zd, xd, outb)
    t_graph = timeit(lambda: graph.launch(s))
    macs = onp.macs_per_frame(tz, tx)
    print(f"--- timings ({a.geom}, B={B}) us per step")
    for name, t, mk in (("stem", t_stem, "stem"), ("blocks", t_blk, "blocks"), ("head+decode", t_head, "head")):
        print(f"{name:12s} {t:9.1f} us   {2 * macs[mk] * B / t / 1e6:8.2f} TFLOP/s  ({2 * macs[mk] * B / t / 1e6 / 157.3 * 100:.1f}% of fp32 peak)")
    for name, t in (("eager step", t_eager), ("graph step", t_graph)):
        print(f"{name:12s} {t:9.1f} us   {B / t * 1e6:12.0f} frames/s   {2 * macs['total'] * B / t / 1e6 / 157.3 * 100:.1f}% of fp32 peak")


if __name__ == "__main__":
    main()
