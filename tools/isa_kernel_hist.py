#!/usr/bin/env python3
"""Static opcode histogram of ONE kernel of a `hipcc -S --cuda-device-only` listing (development aid).
    tools/isa_kernel_hist.py /tmp/vt.s 'blocks_kernel<5, 8, 1, true, true, false, true>' [--dump out.s]
Prints VALU (without MFMA) / MFMA / LDS / scalar totals and the most frequent opcodes."""
import collections, re, subprocess, sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
starts.append((len(src), "end"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if want not in dn:
        continue
    body = src[a:b]
    end = next((i for i, l in enumerate(body) if "s_endpgm" in l), len(body))
    body = body[:end]
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
    ops = collections.Counter()
    for l in body:
        m = re.match(r"^\s+([vsd]\w+|global\w+|buffer\w+|ds_\w+|scratch\w+)", l)
        if m:
            ops[m.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k)
    print(dn, "lines", len(body))
    print("VALU", valu, "MFMA", sum(v for k, v in ops.items() if "mfma" in k), "LDS", sum(v for k, v in ops.items() if k.startswith("ds_")),
          "scratch", sum(v for k, v in ops.items() if k.startswith("scratch")), "s_nop", ops["s_nop"])
    for k, v in ops.most_common(40):
        print(f"  {k:32s}{v}")
