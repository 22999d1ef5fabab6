#!/usr/bin/env python3
"""Step latency at small batches for the kernel-selection switches (one process per combination: the switches are read at vt_create)."""
import os, subprocess, sys, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    geom = sys.argv[2]
    out = {}
    for B in [int(v) for v in os.environ.get("SWEEP_B", "96,128,160,192,224").split(",")]:
        r = bench.Runner(geom, B)
        out[B] = round(r.time_us(lambda: r.graph.launch(r.stream), 300, warm=50), 2)
        r.close()
    print(json.dumps(out))
    sys.exit(0)
for geom in ("G128", "G256"):
    envs = ({}, {"VT_STEM_FUSED": "0", "VT_STEM_PIPE": "0"}, {"VT_HEAD_FUSED": "0"})
    if os.environ.get("SWEEP_HEAD"):      # F = 16 head: conv1 as its own launch over row strips against the per-tower form
        envs = ({"VT_HEAD_SPLIT": "0"}, {"VT_HEAD_SPLIT": "1"})
    elif os.environ.get("SWEEP_TILE"):      # the tile-parallel blocks form against the one-workgroup-per-frame form
        envs = ({"VT_BLOCKS_TILE": "0"}, {"VT_BLOCKS_TILE": "1"})
    for env in envs:
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, __file__, "child", geom], env=e, capture_output=True, text=True)
        print(geom, env, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:])
