"""Test infrastructure (build container only): run the REFERENCE's own evaluation harness against this repo's plug-in.

`overlay(dst)` mirrors /root/reference into `dst` with symlinks (nothing of the reference is copied or modified) and then does
what INTEGRATION.md section 1 tells a maintainer to do: drops `integration/lib/test/tracker/vit_dist.py` (and optionally
`integration/lib/test/parameter/vit_dist.py`) over the reference's files, and writes the user's `lib/test/evaluation/local.py`
(the reference generates that file per installation: lib/test/evaluation/environment.py:87-124).

`install_stand_ins()` registers import-time stand-ins for the third-party modules the reference harness imports but this image
lacks (cv2, lmdb, visdom, jpeg4py, tensorboardX, torchvision, timm, ... : SURVEY.md 8(c)); none of them is called on the paths
exercised here (plug-in discovery, `create_tracker`'s class lookup, `parameters()`).  `easydict.EasyDict` is an attribute dict;
`torch._six` (removed from torch 2.x, imported by lib/train/data/loader.py:5) gets its two constants.

Run as a script (always in a fresh process: it edits sys.meta_path / sys.modules):
    python tests/ref_overlay.py probe <dst> [--ref-params]     -> one JSON line describing what the reference harness resolved
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import json
import os
import shutil
import sys
import types

REF = os.environ.get("VT_REFERENCE", "/root/reference")
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ABSENT = ("cv2", "lmdb", "visdom", "jpeg4py", "torchvision", "timm", "thop", "tensorboardX", "pycocotools", "wandb", "tikzplotlib",
          "matplotlib", "colorama")
PLUGIN = "lib/test/tracker/vit_dist.py"
PARAMS = "lib/test/parameter/vit_dist.py"
LOCAL = "lib/test/evaluation/local.py"


def overlay(dst: str, shim_params: bool = True) -> str:
    real = {"", "lib", "lib/test", "lib/test/tracker", "lib/test/parameter", "lib/test/evaluation"}
    for d in sorted(real):
        os.makedirs(os.path.join(dst, d), exist_ok=True)
        for e in os.listdir(os.path.join(REF, d)):
            rel = os.path.join(d, e) if d else e
            if rel in real or e == "__pycache__":
                continue
            os.symlink(os.path.join(REF, rel), os.path.join(dst, rel))
    for f in [PLUGIN] + ([PARAMS] if shim_params else []):
        os.remove(os.path.join(dst, f))                                   # the symlink, not the reference's file
        shutil.copy(os.path.join(REPO, "integration", f), os.path.join(dst, f))
    os.remove(os.path.join(dst, LOCAL))
    with open(os.path.join(dst, LOCAL), "w") as fh:
        fh.write("from lib.test.evaluation.environment import EnvSettings\n\n\ndef local_env_settings():\n    s = EnvSettings()\n"
                 f"    s.prj_dir = {dst!r}\n    s.save_dir = {os.path.join(dst, 'output')!r}\n"
                 f"    s.results_path = {os.path.join(dst, 'output', 'test', 'tracking_results')!r}\n    return s\n")
    return dst


class _StandIn(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        t = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, t)
        return t


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StandIn(spec.name)

    def exec_module(self, module):
        pass


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def install_stand_ins():
    sys.meta_path.insert(0, _Finder())
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    import torch  # noqa: F401
    six = types.ModuleType("torch._six")
    six.string_classes, six.int_classes = (str, bytes), int
    sys.modules["torch._six"] = six


def plain(o):
    """cfg (EasyDict) -> plain nested dict of JSON types"""
    if isinstance(o, dict):
        return {str(k): plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    return o


def probe(dst: str, shim_params: bool):
    """What the reference's harness resolves for `tracking/test.py vit_dist vit_48_h32_noKD`: lib/test/evaluation/tracker.py
    :54-64 (discovery), :276-280 (parameters)."""
    overlay(dst, shim_params)
    install_stand_ins()
    sys.path.insert(0, dst)
    sys.path.insert(1, REPO)
    T = importlib.import_module("lib.test.evaluation.tracker")
    assert os.path.realpath(T.__file__) == os.path.realpath(os.path.join(REF, "lib/test/evaluation/tracker.py")), T.__file__
    tr = T.Tracker("vit_dist", "vit_48_h32_noKD", "synthetic", run_id=None)
    cls = tr.tracker_class
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):        # the reference prints the whole cfg
        params = tr.get_parameters()
    pm = sys.modules["lib.test.parameter.vit_dist"]
    save_dir = os.path.join(dst, "output")
    return {
        "harness_file": os.path.relpath(os.path.realpath(T.__file__), REF),
        "tracker_class": f"{cls.__module__}.{cls.__qualname__}",
        "tracker_class_file": os.path.relpath(os.path.realpath(sys.modules[cls.__module__].__file__), REPO),
        "has_methods": [m for m in ("initialize", "track") if callable(getattr(cls, m, None))],
        "results_dir": os.path.relpath(tr.results_dir, save_dir),
        "params_module_file": os.path.realpath(pm.__file__),
        "params_function_module": pm.parameters.__module__,
        "params": {k: (os.path.relpath(v, save_dir) if k == "checkpoint" else plain(v)) for k, v in sorted(vars(params).items())},
    }


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "probe":
        print(json.dumps(probe(sys.argv[2], "--ref-params" not in sys.argv)))
    else:
        raise SystemExit(__doc__)
