#!/usr/bin/env python3
"""fp32 build, B = 256: the step with the template cache (vt_set_template + forward(None, x): search-only stem, block kernel's ZC form) against
the plain step, as captured graphs of 4 steps -- us per step -- and the two block-kernel forms alone (stage API)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vittracker_amd import native, synth

def time_us(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n

for geom in ((64, 128), (128, 256)):
    B = 256
    m = native.Model(geom[0], geom[1], max_batch=B)
    m.load_state_dict(synth.synth_state_dict(0, len_z=(geom[0] // 16) ** 2, len_x=(geom[1] // 16) ** 2))
    z, x = synth.synth_inputs(0, B, geom[0], geom[1])
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    m.set_template(zd)
    g_plain, _ = m.capture_steps([zd] * 4, [xd] * 4)
    g_cache, _ = m.capture_steps(None, [xd] * 4)
    s = torch.cuda.current_stream()
    tp = time_us(lambda: g_plain.launch(s)) / 4
    tc = time_us(lambda: g_cache.launch(s)) / 4
    print(f"G{geom[1]}: plain step {tp:.2f} us, cached-template step {tc:.2f} us")
    m.close()
