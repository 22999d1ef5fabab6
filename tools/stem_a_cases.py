import sys
sys.path.insert(0, "tools")
import phase_times as pt
for name, env in [("baseline", {}), ("a: nothing (3)", {"VT_SKIP_STEM_A": "3"}), ("a: no L1 (1)", {"VT_SKIP_STEM_A": "1"}),
                  ("a: no L1, no stores (5)", {"VT_SKIP_STEM_A": "5"}), ("a: no L1, no mfma (9)", {"VT_SKIP_STEM_A": "9"}),
                  ("a: no L1, no st, no mfma (13)", {"VT_SKIP_STEM_A": "13"}),
                  ("a: full L1, L2 no stores (4)", {"VT_SKIP_STEM_A": "4"}), ("a: full L1, L2 no mfma (8)", {"VT_SKIP_STEM_A": "8"}),
                  ("a+b nothing", {"VT_SKIP_STEM_A": "3", "VT_SKIP_STEM_B": "15"})]:
    print(f"{name:32s} {pt.run(env, 'G128')}", flush=True)
