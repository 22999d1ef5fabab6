import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vittracker_amd import native, synth
B = 256
m = native.Model(64, 128, max_batch=B)
m.load_state_dict(synth.synth_state_dict(3, len_z=16, len_x=64))
z, x = synth.synth_inputs(1, B, 64, 128)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
m.set_template(zd)
g0, _ = m.capture(zd, xd)
g1, _ = m.capture(None, xd)
s = torch.cuda.Stream()
def t(g, n=400):
    with torch.cuda.stream(s):
        for _ in range(50): g.launch(s)
    s.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(n): g.launch(s)
    s.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(2):
    print("full step (z, x): %.1f us    cached template (x only): %.1f us" % (t(g0), t(g1)))
