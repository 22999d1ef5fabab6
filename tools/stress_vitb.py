#!/usr/bin/env python3
"""Race screen of the ViT-Base path (the GEMM k-loop reads LDS-DMA data by counted waits and barriers, two wave groups one barrier
apart): the captured step replayed many times on fixed inputs must be bit-identical to the first replay, at several batch sizes
(partial last tiles, one and several tiles per persistent workgroup)."""
import sys
sys.path.insert(0, ".")
import torch
from vittracker_amd import native, synth
sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("1", "5", "96", "256"))]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
sd = synth.synth_vitb_state_dict(26)
for B in sizes:
    m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=B)
    m.load_state_dict(sd)
    z, x = synth.synth_inputs(B + 3, B, 128, 256)
    zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
    graph, out = m.capture(zd, xd)
    graph.launch(); torch.cuda.synchronize()
    ref = {k: getattr(out, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")}
    for it in range(reps):
        graph.launch()
        if it % 10 == 9 or it == reps - 1:
            torch.cuda.synchronize()
            for k, v in ref.items():
                if not torch.equal(getattr(out, k), v):
                    bad += 1
                    print("MISMATCH", B, it, k, float((getattr(out, k) - v).abs().max()))
    print("vitb", B, "ok" if not bad else "BAD", flush=True)
    graph = None; m.close()
print("stress:", "PASS" if bad == 0 else f"FAIL ({bad})")
sys.exit(1 if bad else 0)
