#!/usr/bin/env python3
"""Two batches in flight: two models (own workspaces) with their own 4-step graphs on two streams, launched alternately, against
one model on one stream.  Does the next graph's first kernel fill the CUs that the previous graph's last kernel is leaving,
and does the gap between two graph launches disappear?

    python tools/two_stream_probe.py [--geom G128] [--B 256] [--iters 300]
"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="G128")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--spg", type=int, default=4)
    ap.add_argument("--models", type=int, default=2)
    a = ap.parse_args()
    import torch
    import bench
    rs = [bench.Runner(a.geom, a.B, seed=i, steps_per_graph=a.spg) for i in range(a.models)]
    for r in rs:
        r.check_against_golden()
        r.prewarm(0.3)

    def run(n_models, iters):
        for r in rs[:n_models]:
            r.stream.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            r = rs[i % n_models]
            r.graph_s.launch(r.stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (iters * rs[0].S) * 1e6

    for rep in range(3):
        print(f"{a.geom} B={a.B} {a.spg} steps per graph: " + "   ".join(
            f"{n} stream(s) {t:.2f} us per step ({a.B / t:.3f} M frames/s)" for n, t in ((n, run(n, a.iters)) for n in range(1, a.models + 1))))


if __name__ == "__main__":
    main()
