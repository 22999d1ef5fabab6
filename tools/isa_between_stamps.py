#!/usr/bin/env python3
"""Instruction mix between consecutive s_memtime stamps of the WLDS block kernel, in program order
(development aid; input: `hipcc -S --cuda-device-only` output)."""
import sys
from collections import Counter

lines = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/vt.s").read().split("\n")
name = sys.argv[2] if len(sys.argv) > 2 else "_ZN3vtb13blocks_kernelILi5ELi5ELi1ELb1EEE"
start = next(i for i, l in enumerate(lines) if l.startswith(name) and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
idx = [i for i, l in enumerate(body) if "s_memtime" in l]
print(len(body), "lines,", len(idx), "stamps")


def op_of(l):
    l = l.strip()
    if not l or l[0] in ";." or l.split(";")[0].strip().endswith(":"):
        return None
    return l.split()[0]


prev = 0
for k, i in enumerate(idx + [len(body)]):
    c = Counter()
    for l in body[prev:i]:
        op = op_of(l)
        if op is None:
            continue
        if "mfma" in op:
            c["mfma"] += 1
        elif op.startswith("ds_"):
            c[op] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["_".join(op.split("_")[:2])] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            c["branch"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(k, dict(sorted(c.items())))
    prev = i
