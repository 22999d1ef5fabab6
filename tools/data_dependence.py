#!/usr/bin/env python3
"""Do the kernels' durations depend on the DATA?  The three stages of the G128 step alone (B = 256, HIP events) on search crops that are
N(0,1) noise (bench.py's), normalised uniform-noise patches (what the demo's noise frames give), a smooth image, zeros -- same weights."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from vittracker_amd import synth
geom = os.environ.get("GEOM", "G128")
r = bench.Runner(geom, 256, steps_per_graph=4)
tz, tx = bench.GEOMS[geom]
B = 256
rs = np.random.RandomState(1)
noise_u8 = rs.randint(0, 256, (B, tx, tx, 3)).astype(np.uint8)
yy, xx = np.mgrid[0:tx, 0:tx]
smooth = np.stack([(127 + 100 * np.sin(xx / 17.0 + b) * np.cos(yy / 23.0)).astype(np.uint8) for b in range(B)])[..., None].repeat(3, -1)
cases = {"N(0,1) (bench.py)": r.x.clone(), "uniform-noise patches, normalised": torch.from_numpy(synth.normalise_patches(noise_u8)).cuda(),
         "smooth patches, normalised": torch.from_numpy(synth.normalise_patches(np.ascontiguousarray(smooth))).cuda(), "zeros": torch.zeros_like(r.x)}
for name, x in cases.items():
    r.x.copy_(x)
    torch.cuda.synchronize()
    st = r.stage_times(60)
    t = r.time_us(lambda: r.graph_s.launch(r.stream), 60) / r.S
    print(f"{geom} {name:36s} stem {st['stem']:6.2f}  blocks {st['blocks']:6.2f}  head {st['head']:6.2f}  step {t:6.2f} us")
