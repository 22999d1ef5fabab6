#!/usr/bin/env python3
"""Phase costs of the G256 stem (stem_pipe + stem_b) via VT_SKIP_STEM_A (1: no layer 1, 2: no layer 2) and
VT_SKIP_STEM_B (15: stem_b does nothing)."""
import sys
sys.path.insert(0, "tools")
import phase_times as pt
for name, env in [("baseline", {}), ("stem_b off", {"VT_SKIP_STEM_B": "15"}),
                  ("stem_b off, pipe: no L1", {"VT_SKIP_STEM_B": "15", "VT_SKIP_STEM_A": "1"}),
                  ("stem_b off, pipe: no L2", {"VT_SKIP_STEM_B": "15", "VT_SKIP_STEM_A": "2"}),
                  ("stem_b off, pipe: nothing", {"VT_SKIP_STEM_B": "15", "VT_SKIP_STEM_A": "3"}),
                  ("pipe: nothing", {"VT_SKIP_STEM_A": "3"})]:
    print(f"{name:28s} {pt.run(env, 'G256')}", flush=True)
