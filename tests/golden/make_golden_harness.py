#!/usr/bin/env python3
"""Golden fixture of the reference's OWN `parameters('vit_48_h32_noKD')` (lib/test/parameter/vit_dist.py:7-30, with
lib/test/utils/params.py, lib/config/vit_dist/config.py and experiments/vit_dist/vit_48_h32_noKD.yaml), obtained by driving
the reference's harness class (`lib/test/evaluation/tracker.py::Tracker.get_parameters`, :276-280) in a symlink overlay of
/root/reference (tests/ref_overlay.py).  Build container only; the JSON is data (attribute values), no reference source.

    python tests/golden/make_golden_harness.py        # writes tests/golden/ref_params_vit_48_h32_noKD.json
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "ref_overlay.py"), "probe", os.path.join(tmp, "tree"), "--ref-params"],
                           capture_output=True, text=True, check=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["params_module_file"].endswith("reference/lib/test/parameter/vit_dist.py"), d["params_module_file"]
    out = {"yaml_name": "vit_48_h32_noKD", "source": "reference lib/test/parameter/vit_dist.py::parameters, through "
           "lib/test/evaluation/tracker.py::Tracker.get_parameters", "checkpoint_relative_to": "env_settings().save_dir",
           "results_dir_relative_to": "env_settings().save_dir", "results_dir": d["results_dir"], "params": d["params"]}
    path = os.path.join(HERE, "ref_params_vit_48_h32_noKD.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
