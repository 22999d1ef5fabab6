#!/usr/bin/env python3
"""Instruction mix per barrier-delimited segment of a kernel's ISA (MFMA / VALU / LDS / VMEM counts, optionally the VALU opcode
histogram of one segment).  f32 MFMA and VALU issue share a pipe on gfx950, so VALU instructions around a phase's MFMAs are phase
time: this is how the ~15-instruction-per-chunk offset arithmetic in the stems' layer-3 / layer-4 phases was found.

    python tools/isa_segments.py vittrack stem_fused_kernelILi0ELb0            # vittrack.hip, kernels whose mangled name contains ...
    python tools/isa_segments.py vitb gemm_kernelILi256ELi256ELi2ELi4ELi0ELi3 --blocks   # split at basic blocks instead of barriers
    python tools/isa_segments.py vittrack stem_fused_kernelILi0ELb0 --hist 12 # opcode histogram of segment 12
"""
import argparse, os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit", choices=["vittrack", "vitb"])
    ap.add_argument("kernel")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--hist", type=int, default=-1)
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    out = os.path.join(tempfile.gettempdir(), f"{a.unit}.s")
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-align-all-functions=14", "--cuda-device-only", "-S", "-o", out,
           os.path.join(ROOT, "vittracker_amd", "csrc", a.unit + ".hip")] + a.flags.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if a.kernel in m.group(1)]
    if not names:
        sys.exit(f"no kernel matching {a.kernel!r}")
    for name in names:
        i = s.index(name + ":"); j = s.index("s_endpgm", i)
        body = s[i:j]
        if a.blocks:
            segs = [[t for t in (ln.strip().split(";")[0].strip() for ln in b.split("\n")) if t and not t.startswith(".") and not t.endswith(":")]
                    for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body)]
        else:
            segs = [[]]
            for ln in body.split("\n"):
                t = ln.strip().split(";")[0].strip()
                if not t or t.startswith(".") or t.endswith(":"):
                    continue
                segs[-1].append(t)
                if t.startswith("s_barrier"):
                    segs.append([])
        print(name)
        for k, sg in enumerate(segs):
            c = Counter()
            for t in sg:
                op = t.split()[0]
                if "mfma" in op: c["mfma"] += 1
                elif op.startswith("ds_read"): c["ds_read"] += 1
                elif op.startswith("ds_write"): c["ds_write"] += 1
                elif "load_lds" in op: c["lds_dma"] += 1
                elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
                elif op.startswith("s_nop"): c["nop"] += 1
            if len(sg) > 20:
                print(f"  segment {k:3d}: {len(sg):5d} instructions  {dict(c)}")
            if k == a.hist:
                print("     ", Counter(t.split()[0] for t in sg if t.startswith("v_") and "mfma" not in t).most_common(30))

if __name__ == "__main__":
    main()
