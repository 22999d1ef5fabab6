import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from vittracker_amd import native, synth
B = 256
ms = []
for k in range(2):
    m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
    m.load_state_dict(synth.synth_vitb_state_dict(26))
    z, x = synth.synth_inputs(k, B, 128, 256)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    g, out = m.capture(zd, xd)
    ms.append((m, g, out, zd, xd, torch.cuda.Stream()))
def run(n_models, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters):
        m, g, out, zd, xd, s = ms[i % n_models]
        g.launch(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for rep in range(3):
    run(1, 4); a = run(1, 16); run(2, 4); b = run(2, 16)
    print(f"ViT-Base B=256: one stream {a:.2f} ms per step ({B/a:.0f} f/ms -> {B/a*1e3:.0f} frames/s)   two streams {b:.2f} ms ({B/b*1e3:.0f} frames/s)")
