import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
GEOMS = {"G256": (128, 256), "G128": (64, 128)}  # name -> (template_size, search_size)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_files(prefix="ref_G"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_case(path):
    """Return (fixture dict, state dict, z, x) for one golden file; weights and inputs are
    regenerated from the seed and checked against the checksum stored at generation time."""
    from vittracker_amd import synth
    g = dict(np.load(path, allow_pickle=False))
    geom, seed, B = str(g["geom"]), int(g["seed"]), int(g["B"])
    tz, tx = GEOMS[geom]
    sd = synth.synth_state_dict(seed, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    assert synth.state_checksum(sd) == str(g["state_checksum"]), \
        "vittracker_amd/synth.py drifted from the generator the fixtures were made with"
    z, x = synth.synth_inputs(seed, B, tz, tx)
    return g, sd, z, x


def cfg_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_cfg_*.npz")))


def load_cfg_case(path):
    """(fixture, state dict, z, x, (channels, heads, head_channels)) of a config-surface golden file (tests/golden/make_golden_cfg.py):
    the reference model built with other CHANNELS / HEADS / HEAD.NUM_CHANNELS than the shipped YAML's."""
    from vittracker_amd import synth
    g = dict(np.load(path, allow_pickle=False))
    geom, seed, B = str(g["geom"]), int(g["seed"]), int(g["B"])
    C, heads, W = int(g["channels"]), int(g["heads"]), int(g["head_channels"])
    tz, tx = GEOMS[geom]
    sd = synth.synth_state_dict(seed, C=C, depth=3, head_ch=W, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    assert synth.state_checksum(sd) == str(g["state_checksum"]), "synth_state_dict drifted from the fixture generator"
    z, x = synth.synth_inputs(seed, B, tz, tx)
    return g, sd, z, x, (C, heads, W)


def u8_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_u8_*.npz")))


def load_u8_case(path):
    """(fixture, state dict, z fp32 template crops, uint8 search patches (B,S,S,3)) of a uint8-patch golden file
    (tests/golden/make_golden_u8.py): the reference model on Preprocessor-normalised patches."""
    from vittracker_amd import synth
    g = dict(np.load(path, allow_pickle=False))
    geom, seed, B = str(g["geom"]), int(g["seed"]), int(g["B"])
    tz, tx = GEOMS[geom]
    sd = synth.synth_state_dict(seed, len_z=(tz // 16) ** 2, len_x=(tx // 16) ** 2)
    assert synth.state_checksum(sd) == str(g["state_checksum"]), "synth_state_dict drifted from the fixture generator"
    z, _ = synth.synth_inputs(seed, B, tz, tx)
    patches = synth.synth_patches(seed, B, tx)
    assert int(patches.astype(np.uint64).sum()) == int(g["patch_checksum"]), "synth_patches drifted from the fixture generator"
    return g, sd, z, patches


@pytest.fixture(scope="session")
def native():
    """The C-ABI library (loads everywhere; compute calls need a GPU)."""
    from vittracker_amd import native as nat
    return nat


def vitb_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_vitb_*.npz")))


def load_vitb_case(path):
    """(fixture, state dict, z, x) of a ViT-Base golden file; weights / inputs regenerated from the seed."""
    from vittracker_amd import synth
    g = dict(np.load(path, allow_pickle=False))
    seed, B = int(g["seed"]), int(g["B"])
    sd = synth.synth_vitb_state_dict(seed, common_mode=float(g["common_mode"]) if "common_mode" in g else 0.0)
    assert synth.state_checksum(sd) == str(g["state_checksum"]), "synth_vitb_state_dict drifted from the fixture generator"
    z, x = synth.synth_inputs(seed, B, 128, 256)
    return g, sd, z, x
