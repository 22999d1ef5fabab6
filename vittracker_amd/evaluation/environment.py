"""Path settings (stand-in for the generated ``lib/test/evaluation/local.py`` of the reference,
``lib/test/evaluation/environment.py:87-124``): only the two fields the vit_dist parameter file
reads -- ``prj_dir`` and ``save_dir`` (``lib/test/parameter/vit_dist.py:9-10``)."""
import os

_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))


class EnvSettings:
    def __init__(self):
        self.prj_dir = os.environ.get("VITTRACK_PRJ_DIR", _REPO)
        self.save_dir = os.environ.get("VITTRACK_SAVE_DIR", os.path.join(self.prj_dir, "output"))
        self.results_path = os.path.join(self.save_dir, "test", "tracking_results")


def env_settings():
    return EnvSettings()
