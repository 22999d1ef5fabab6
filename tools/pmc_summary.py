#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over all dispatches found under a directory."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "rocclr" in k or "at::" in k:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(f"== {k}  (dispatches: {len(next(iter(acc[k].values())))})")
    for n in sorted(c):
        print(f"   {n:28s} {c[n]:16.1f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        def pct(n):
            return 100.0 * c.get(n, 0.0) / wc
        print(f"   -> of wave-cycles: wait_any {pct('SQ_WAIT_ANY'):.1f}%  wait_inst {pct('SQ_WAIT_INST_ANY'):.1f}%  "
              f"active_any {pct('SQ_ACTIVE_INST_ANY'):.1f}%  active_valu {pct('SQ_ACTIVE_INST_VALU'):.1f}%")
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        print(f"   -> MFMA busy cycles / SQ busy cycles = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
