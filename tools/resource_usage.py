#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage for a .hip file, one line per kernel (VGPRs, scratch, occupancy)."""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1])
flags = sys.argv[2:]
d = os.path.dirname(os.path.abspath(src))
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + flags, cwd=d, capture_output=True, text=True).stderr
cur = None
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
            if key == "lds":
                print(f"{cur['name'][:80]:80s} vgpr {cur.get('vgpr'):4d} agpr {cur.get('agpr', 0):3d} scratch {cur.get('scratch'):4d} occ {cur.get('occ')}")
                cur = None
