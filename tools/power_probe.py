#!/usr/bin/env python3
"""Is the batch-256 step power-bound?  Replays the bench graph back to back for --seconds while a thread samples the GPU's hwmon
files (package power, shader clock, temperature; whatever the box exposes to an ordinary user) and prints µs per step per
window beside them.  A/B by environment (VT_HEAD_BF3=1, VT_*) in the caller.

    python tools/power_probe.py [--geom G128] [--B 256] [--seconds 8] [--idle 2]
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def sensors():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out[f"{os.path.basename(os.path.dirname(card))}:{name}"] = p
        for name in ("gpu_busy_percent",):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out[f"{os.path.basename(os.path.dirname(card))}:{name}"] = p
    return out


def read(p):
    try:
        with open(p) as f:
            return float(f.read().split()[0])
    except Exception:
        return float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="G128")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--idle", type=float, default=2.0)
    ap.add_argument("--window", type=float, default=0.5)
    ap.add_argument("--lib", default="", help="another build of the library (build_variants/*.so)")
    a = ap.parse_args()
    import bench
    if a.lib:
        from vittracker_amd import native
        native.LIB_PATH = os.path.abspath(a.lib)
    sens = sensors()
    print("sensors:", json.dumps(sens))
    r = bench.Runner(a.geom, a.B, steps_per_graph=4)
    r.check_against_golden()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), {k: read(p) for k, p in sens.items()}))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(a.idle)
    torch = r.torch
    windows = []
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < a.seconds:
        n = max(50, int(a.window * 1e6 / (95.0 * r.S * (4 if a.geom == "G256" else 1))))
        w0 = time.perf_counter()
        us = r.time_us(lambda: r.graph_s.launch(r.stream), n, warm=0) / r.S
        windows.append((w0, time.perf_counter(), us))
    mhz, cpm, _ = r.native.probe_clock(20000, 1)
    time.sleep(a.idle)
    stop.set()
    th.join()

    def mean_in(t0, t1, key):
        v = [s[key] for t, s in samples if t0 <= t <= t1 and s[key] == s[key]]
        return sum(v) / len(v) if v else float("nan")

    keys = sorted(sens)
    print("idle before:", {k: round(mean_in(0, t_start, k), 1) for k in keys})
    for w0, w1, us in windows:
        print(f"t={w0 - t_start:6.2f}s  {us:7.2f} us/step  " + "  ".join(f"{k.split(':')[1]}={mean_in(w0, w1, k):.0f}" for k in keys))
    print(f"post-run clock probe: {mhz:.0f} MHz, {cpm:.2f} cycles per fp32 MFMA")
    print("RESULT " + json.dumps({"us_first": windows[0][2], "us_last": windows[-1][2], "us_mean": sum(w[2] for w in windows) / len(windows),
                                  "switches": {k: v for k, v in os.environ.items() if k.startswith(("VT_", "VB_"))}}))


if __name__ == "__main__":
    main()
