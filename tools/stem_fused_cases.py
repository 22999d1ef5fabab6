#!/usr/bin/env python3
"""Phase costs of stem_fused_kernel via the VT_SKIP_STEM_A mask (1: no L1/L2 pipeline, 2: no layer 2,
4: no layer 3, 8: no layer 4).  Results are wrong by design; only durations matter."""
import sys
sys.path.insert(0, "tools")
import phase_times as pt
for name, env in [("baseline", {}), ("nothing (15)", {"VT_SKIP_STEM_A": "15"}), ("only L1 (14)", {"VT_SKIP_STEM_A": "14"}),
                  ("L1+L2 (12)", {"VT_SKIP_STEM_A": "12"}), ("L1+L2+L3 (8)", {"VT_SKIP_STEM_A": "8"}),
                  ("only L3+L4 (1)", {"VT_SKIP_STEM_A": "1"}), ("only L3 (9)", {"VT_SKIP_STEM_A": "9"}), ("only L4 (5)", {"VT_SKIP_STEM_A": "5"})]:
    print(f"{name:24s} {pt.run(env, sys.argv[1] if len(sys.argv) > 1 else 'G128')}", flush=True)
