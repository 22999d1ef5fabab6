#!/usr/bin/env python3
"""Per-kernel resource usage of both translation units (the Makefile's flags, `-Rpass-analysis=kernel-resource-usage`): VGPRs,
ScratchSize, VGPR / SGPR spills, LDS.  `python tools/resource_table.py > profiles/r6_resource_usage.txt`; tests/test_resource_usage.py
holds the default-path kernels to ScratchSize 0 and no VGPR spill.  (A kernel may show a non-zero ScratchSize with `VGPRs Spill: 0`
and no scratch instruction: SGPRs spilled to VGPR lanes reserve a frame that is never touched -- stem_a2.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "vittracker_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-mllvm", "-align-all-functions=14"]


def table(src, extra=()):
    err = subprocess.run(["hipcc"] + FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                         cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    pats = (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
            ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
    for ln in err.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"^void ", "", name.split("(")[0])}
            continue
        for key, pat in pats:
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
                if key == "lds":
                    rows.append(cur)
                    cur = None
    if not rows:
        raise SystemExit("no kernels found:\n" + err[-2000:])
    return rows


def main():
    print("# hipcc " + " ".join(FLAGS) + " -Rpass-analysis=kernel-resource-usage   (tools/resource_table.py)")
    for src, extra in (("vittrack.hip", ()), ("vittrack.hip", ("-DVT_F16=1",)), ("vitb.hip", ())):
        print(f"## {src} {' '.join(extra)}")
        for r in table(src, extra):
            print(f"{r['name'][:96]:96s} vgpr {r['vgpr']:4d}  scratch {r['scratch']:4d}  vgpr-spill {r['vspill']:3d}  sgpr-spill {r['sspill']:3d}  occupancy {r['occ']}")


if __name__ == "__main__":
    main()
