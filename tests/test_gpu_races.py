"""Race / hazard screen as a test (round 4 advisor): the large-batch step of each build replayed on fixed inputs must be bit-identical
every time.  tools/race_check.py is the same loop as a script (it also localises a mismatch); it caught the mixed K = 16 / K = 32
MFMA accumulator chain of round 4 (vt_bf3.h RULE: nothing but a comment and separate accumulators at the call sites enforces it), and
it is what screens every new LDS rendezvous / aliasing of the frame-form kernels (round 5: the block kernel's exchange areas inside
the staging buffers).  One subprocess per (build, geometry): the library is chosen at load time."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

BUILDS = {"fp32": "vittracker_amd/csrc/libvittrack_hip.so", "f16": "vittracker_amd/csrc/libvittrack_hip_f16.so"}


@pytest.mark.parametrize("geom", ["G128", "G256"])
@pytest.mark.parametrize("build", sorted(BUILDS))
def test_replays_of_the_large_batch_step_are_bit_identical(build, geom):
    lib = os.path.join(REPO, BUILDS[build])
    if not os.path.exists(lib):
        pytest.fail(f"{lib} is missing: build it (make -C vittracker_amd/csrc all)")
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "race_check.py"), "--lib", lib, "--geom", geom, "--B", "256", "--reps", "24"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-600:], p.stderr[-600:])
    assert "mismatching replays {'stem': 0, 'blocks': 0, 'head': 0, 'forward': 0}" in p.stdout, p.stdout[-600:]
