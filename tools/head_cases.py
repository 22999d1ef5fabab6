#!/usr/bin/env python3
"""Phase costs of the head kernel via the VT_SKIP_HEAD mask (1 zero, 2 load, 4 conv1, 8 conv2, 16 conv3+4)."""
import sys
sys.path.insert(0, "tools")
import phase_times as pt
g = sys.argv[1] if len(sys.argv) > 1 else "G128"
for name, env in [("baseline", {}), ("nothing (31)", {"VT_SKIP_HEAD": "31"}), ("no zero (1)", {"VT_SKIP_HEAD": "1"}),
                  ("no load (2)", {"VT_SKIP_HEAD": "2"}), ("no conv1 (4)", {"VT_SKIP_HEAD": "4"}), ("no conv2 (8)", {"VT_SKIP_HEAD": "8"}),
                  ("no conv3,4 (16)", {"VT_SKIP_HEAD": "16"}), ("only conv1 (27)", {"VT_SKIP_HEAD": "27"})]:
    print(f"{name:20s} {pt.run(env, g)}", flush=True)
