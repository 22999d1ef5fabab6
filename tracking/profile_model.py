#!/usr/bin/env python3
"""Counterpart of the reference's tracking/profile_model.py / profile_model_cpu.py for the MI355X
path: builds the model from a YAML, feeds torch.randn crops (profile_model_cpu.py:104-105) and
times `model(template, search)` with the reference's loop -- 500 warm-up + 1000 timed forwards at
batch 1 (profile_model.py:31-54) -- plus the batched hipGraph replay the hardware is built for.

    python tracking/profile_model.py --config vit_48_h32_noKD [--batch 256]
"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--script", default="vit_dist", choices=["vit_dist"])
    ap.add_argument("--config", default="vit_48_h32_noKD")
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    import torch
    from vittracker_amd import config as C
    from vittracker_amd.model import build_ostrack_dist
    cfg = C.fresh_cfg()
    C.update_config_from_file(os.path.join(ROOT, "experiments", a.script, a.config + ".yaml"), cfg)
    z_sz, x_sz = cfg.TEST.TEMPLATE_SIZE, cfg.TEST.SEARCH_SIZE
    model = build_ostrack_dist(cfg, max_batch=max(1, a.batch)).cuda().eval()
    n_par = sum(v.numel() for k, v in model.state_dict().items() if v.is_floating_point() and "running" not in k)
    print(f"config {a.config}: template {z_sz}, search {x_sz}, params {n_par}")
    template, search = torch.randn(1, 3, z_sz, z_sz).cuda(), torch.randn(1, 3, x_sz, x_sz).cuda()
    with torch.no_grad():
        for _ in range(500):
            _ = model(template, search)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(1000):
            _ = model(template, search)
        torch.cuda.synchronize()
        lat = (time.time() - t0) / 1000
    print("The average overall latency is %.3f ms (batch 1, eager, incl. Python dispatch)" % (lat * 1000))
    print("FPS is %.2f fps" % (1.0 / lat))
    B = a.batch
    nat = model._native()
    zb, xb = torch.randn(B, 3, z_sz, z_sz).cuda(), torch.randn(B, 3, x_sz, x_sz).cuda()
    graph, _ = nat.capture(zb, xb)
    for _ in range(50):
        graph.launch()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(500):
        graph.launch()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 500
    print("batch %d hipGraph replay: %.1f us per step, %.0f frames/s" % (B, dt * 1e6, B / dt))


if __name__ == "__main__":
    main()
