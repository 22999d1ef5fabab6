#!/usr/bin/env python3
"""Counterpart of the reference's ``tracking/profile_model_cpu.py`` (BASELINE config 1).

    python tracking/profile_model_cpu.py --script vit_dist --config vit_48_h32_noKD

Same arguments (``:17-26``), same loop (``:36-49``: batch 1, 500 warm-up + 1000 timed forwards, torch.randn
crops, random-init weights, default thread count) and the same printed lines (MACs, params, latency, FPS).
The product has no CPU execution path, so what is timed is the CPU *baseline*: the torch restatement of the
reference module graph that bench.py's cpu_baseline leg owns (``bench.cpu_profile``).  Sizes come from
``cfg.TEST.*`` of the YAML (``:88-90``)."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def parse_args():
    p = argparse.ArgumentParser(description="Parse args for training")
    p.add_argument("--script", type=str, default="vit_dist", choices=["vit_dist"], help="training script name")
    p.add_argument("--config", type=str, default="vit_48_h32_noKD", help="yaml configure file name")
    p.add_argument("--warmup", type=int, default=500)
    p.add_argument("--timed", type=int, default=1000)
    return p.parse_args()


if __name__ == "__main__":
    a = parse_args()
    import bench
    from vittracker_amd import config as C
    cfg = C.fresh_cfg()
    C.update_config_from_file(os.path.join(ROOT, "experiments", a.script, a.config + ".yaml"), cfg)
    size = (cfg.TEST.TEMPLATE_SIZE, cfg.TEST.SEARCH_SIZE)
    geom = {v: k for k, v in bench.GEOMS.items()}.get(size)
    if geom is None:
        raise SystemExit(f"unsupported (template, search) = {size}")
    bench.cpu_profile((geom,), a.warmup, a.timed)
