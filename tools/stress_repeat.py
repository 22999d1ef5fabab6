#!/usr/bin/env python3
"""Race hunt: replays the captured step many times on fixed inputs and checks every replay is bit-identical to
the first, at several batch sizes and both geometries (the kernels use intra-workgroup rendezvous on LDS
counters, barrier-free set-up phases and buffers that alias in time)."""
import sys
sys.path.insert(0, ".")
import torch
from vittracker_amd import native, synth

bad = 0
for geom, (tz, tx) in (("G128", (64, 128)), ("G256", (128, 256))):
    for B in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("1", "5", "64", "256"))]:
        sd = synth.synth_state_dict(B, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
        m = native.Model(tz, tx, max_batch=B)
        m.load_state_dict(sd)
        z, x = synth.synth_inputs(B + 1, B, tz, tx)
        zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
        graph, out = m.capture(zd, xd)
        graph.launch(); torch.cuda.synchronize()
        ref = {k: getattr(out, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")}
        n = 400 if B >= 64 else 1500
        for it in range(n):
            graph.launch()
            if it % 50 == 49:
                torch.cuda.synchronize()
                for k, v in ref.items():
                    if not torch.equal(getattr(out, k), v):
                        bad += 1
                        print("MISMATCH", geom, B, it, k, float((getattr(out, k) - v).abs().max()))
        torch.cuda.synchronize()
        for k, v in ref.items():
            if not torch.equal(getattr(out, k), v):
                bad += 1
                print("MISMATCH at end", geom, B, k)
        print(geom, B, "ok" if not bad else "BAD", flush=True)
print("stress:", "PASS" if bad == 0 else f"FAIL ({bad})")
sys.exit(1 if bad else 0)
