#!/usr/bin/env python3
"""ViT-Base path: per-stage error of the bf16 HIP kernels against the pinned fp32 torch oracle (GPU box).
Prints max / mean abs error and the relative L2 error of every stage fed with the ORACLE's upstream activation."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from oracle import vitb_oracle_torch as ob
from vittracker_amd import native, synth

seed, B = int(sys.argv[1]) if len(sys.argv) > 1 else 26, int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = synth.synth_vitb_state_dict(seed)
z, x = synth.synth_inputs(seed, B, 128, 256)
m = ob.build_from_state(sd)
acts = {}
t0 = time.time()
with torch.no_grad():
    ref = m(torch.from_numpy(z), torch.from_numpy(x), acts)
print(f"oracle {time.time() - t0:.1f}s")
nat = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
nat.load_state_dict(sd)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()


def rep(name, got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    d = np.abs(got - want)
    print(f"{name:28s} max {d.max():.3e} mean {d.mean():.3e} relL2 {np.linalg.norm(got - want) / np.linalg.norm(want):.3e} |ref|max {np.abs(want).max():.2f}")


tok = nat.stem(zd, xd)
rep("stem tokens", tok.cpu().numpy(), acts["tokens"].numpy())
for k in (1, 2, 6, 12):
    feat, resid = nat.blocks(acts["tokens"].cuda().contiguous(), nblocks=k, want_resid=True)
    rep(f"blocks[0..{k}) resid", resid.cpu().numpy(), acts[f"block{k - 1}"].numpy())
feat = nat.blocks(acts["block10"].cuda().contiguous(), nblocks=0)   # norm only
lnref = torch.nn.functional.layer_norm(acts["block10"], (768,), m.backbone.norm.weight, m.backbone.norm.bias, 1e-6)
rep("final norm only", feat.cpu().numpy(), lnref[:, 64:].detach().numpy())
out = nat.head(acts["norm"][:, 64:].cuda().contiguous())
for k in ("score_map", "size_map", "offset_map"):
    rep("head(oracle norm) " + k, getattr(out, k).cpu().numpy(), ref[k].numpy())
out = nat.forward(zd, xd)
for k in ("score_map", "size_map", "offset_map"):
    rep("forward " + k, getattr(out, k).cpu().numpy(), ref[k].numpy())
rep("forward pred_boxes", out.pred_boxes.cpu().numpy(), ref["pred_boxes"].numpy()[:, 0])
g, o2 = nat.capture(zd, xd)
g.launch(); torch.cuda.synchronize()
print("graph == eager:", all(torch.equal(getattr(out, k), getattr(o2, k)) for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes")))
