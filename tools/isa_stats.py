#!/usr/bin/env python3
"""Instruction-mix summary per kernel from `hipcc -S --cuda-device-only` output (development aid)."""
import re
import sys

src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/vt.s").read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
starts.append((len(src), "end"))
pats = {"fma": r"\bv_fma_f32|\bv_fmac_f32", "pk_fma": r"v_pk_fma", "mfma": r"v_mfma", "valu": r"^\s+v_", "salu": r"^\s+s_(?!load|waitcnt|barrier|nop)",
        "s_load": r"s_load_", "gload": r"global_load", "gstore": r"global_store", "ds_rd": r"ds_read", "ds_wr": r"ds_write",
        "wait": r"s_waitcnt", "vm0": r"vmcnt\(0\)", "lgkm0": r"lgkmcnt\(0\)", "barrier": r"s_barrier", "bperm": r"ds_bpermute|v_permlane|_dpp",
        "branch": r"s_cbranch"}
for (a, name), (b, _) in zip(starts, starts[1:]):
    body = src[a:b]
    end = next((i for i, l in enumerate(body) if "s_endpgm" in l), len(body))
    body = body[:end]
    row = {k: sum(1 for l in body if re.search(p, l)) for k, p in pats.items()}
    print(name[:48].ljust(48), " ".join(f"{k}={v}" for k, v in row.items()))
