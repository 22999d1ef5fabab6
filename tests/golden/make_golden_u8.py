#!/usr/bin/env python3
"""Golden vectors for the uint8-patch path: the REFERENCE's own model (imported by make_golden.py's loader,
never copied) on search crops that went through the reference's pre-processing arithmetic.

    python tests/golden/make_golden_u8.py        # build container only (needs /root/reference); writes tests/golden/ref_u8_*.npz

Inputs are regenerated from seeds (vittracker_amd.synth: synth_patches -- uint8 (B,S,S,3) arrays of the shape sample_target
returns --, synth_inputs for the fp32 template crop, synth_state_dict for the weights); only expected outputs are stored.
The normalisation is `Preprocessor.process` (lib/test/tracker/data_utils.py:11-17): the class itself cannot run here (its
__init__ and process() call .cuda()), so its one arithmetic line, `((img_tensor / 255.0) - self.mean) / self.std` with the class's
mean / std (:8-9) and its permute((2,0,1)), is evaluated with torch on the CPU below -- where `/ 255.0` is a true division (on a GPU
torch multiplies by the float reciprocal: a last-bit difference in the normalised crop, 1e-7, which the 1e-5 tolerance of the tests
that read these fixtures covers).  Every other multiply in the outputs is executed by reference source files."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from vittracker_amd import synth  # noqa: E402

CASES = [("G128", 5, 4), ("G256", 6, 2)]     # (geom, seed, B)


def preprocess(patches: np.ndarray) -> torch.Tensor:
    mean = torch.tensor([0.485, 0.456, 0.406]).view((1, 3, 1, 1))       # data_utils.py:8
    std = torch.tensor([0.229, 0.224, 0.225]).view((1, 3, 1, 1))        # data_utils.py:9
    img = torch.tensor(patches).float().permute((0, 3, 1, 2))           # :13 (batched)
    return (((img / 255.0) - mean) / std).contiguous()                  # :14


def main():
    torch.manual_seed(0)
    model_mod, config_mod, _box_ops, hann_mod = mg.import_reference()
    for geom, seed, B in CASES:
        tz, tx = mg.GEOMS[geom]
        net = mg.build_reference_model(model_mod, config_mod, geom)
        sd = synth.synth_state_dict(seed, C=48, depth=3, head_ch=32, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
        assert not missing and not unexpected
        z, _ = synth.synth_inputs(seed, B, tz, tx)
        patches = synth.synth_patches(seed, B, tx)
        x = preprocess(patches)
        with torch.no_grad():
            out = net(torch.from_numpy(z), x)
            F = net.box_head.feat_sz
            win = hann_mod.hann2d(torch.tensor([F, F]).long(), centered=True)
            hbox = net.box_head.cal_bbox(win * out["score_map"], out["size_map"], out["offset_map"])
            conf = out["score_map"].flatten(1).max(dim=1).values
        res = {"geom": geom, "seed": seed, "B": B, "state_checksum": synth.state_checksum(sd),
               "patch_checksum": int(patches.astype(np.uint64).sum()),
               "score_map": out["score_map"].numpy(), "size_map": out["size_map"].numpy(), "offset_map": out["offset_map"].numpy(),
               "pred_boxes": out["pred_boxes"].numpy(), "hann_boxes": hbox.numpy(), "conf": conf.numpy()}
        name = f"ref_u8_{geom}_s{seed}_b{B}.npz"
        np.savez_compressed(os.path.join(HERE, name), **res)
        sm = res["score_map"].reshape(B, -1)
        srt = np.sort(sm, axis=1)
        print(f"{name}: score range [{sm.min():.4f}, {sm.max():.4f}] top2 margin min {np.min(srt[:, -1] - srt[:, -2]):.2e}")


if __name__ == "__main__":
    main()
