#!/usr/bin/env python3
"""How much of a step is the gap BETWEEN graph launches?  Times G128 B=256 steps as (a) eager launches, (b) one captured step
per hipGraph (the shipped form), (c) n steps (distinct input batches) captured into one graph through the caller's stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vittracker_amd import native, synth

B = 256
m = native.Model(64, 128, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=16, len_x=64))
NB = 8
zs, xs = [], []
for i in range(NB):
    z, x = synth.synth_inputs(i, B, 64, 128)
    zs.append(torch.from_numpy(z).cuda()); xs.append(torch.from_numpy(x).cuda())
outs = [native.Outputs(B, 8, xs[0].device) for _ in range(NB)]

def timeit(fn, steps_per_call, calls=200, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(calls): fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return dt / (calls * steps_per_call) * 1e6

st = torch.cuda.Stream()
with torch.cuda.stream(st):
    i = [0]
    def eager():
        k = i[0] % NB; i[0] += 1
        m.forward(zs[k], xs[k], out=outs[k], stream=st)
    print("eager          us/step", round(timeit(eager, 1), 2))
    gs = [m.capture(zs[k], xs[k], out=outs[k])[0] for k in range(NB)]
    def single():
        k = i[0] % NB; i[0] += 1
        gs[k].launch(stream=st)
    print("1 step/graph   us/step", round(timeit(single, 1), 2))
    for n in (2, 4, 8):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for k in range(n):
                m.forward(zs[k], xs[k], out=outs[k], stream=torch.cuda.current_stream())
        print(f"{n} steps/graph  us/step", round(timeit(g.replay, n, calls=max(40, 200 // n)), 2))
