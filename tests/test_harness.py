"""CPU: the evaluation harness around the plugin -- result-file formats (byte for byte what the reference's
np.savetxt calls write, lib/test/evaluation/running.py:29-35), the per-sequence loop, run_video, the lock-step batched
runner's bookkeeping (ragged lengths, s % world sharding) and the deployment wire signature -- with a fake tracker in
place of the GPU."""
import io
import os

import numpy as np
import pytest

from conftest import REPO


class _Params:
    save_all_boxes = False
    debug = 0
    template_factor, search_factor, template_size, search_size = 2.0, 4.0, 128, 256


class _FakeTracker:
    """Moves the box by (+1.5, +0.25) per frame; deterministic, no GPU."""

    def __init__(self, params, dataset_name):
        self.params = params
        self.calls = []

    def initialize(self, image, info):
        self.state = [float(v) for v in info["init_bbox"]]
        self.calls.append(("init", image.shape, sorted(info)))

    def track(self, image, info=None):
        self.state = [self.state[0] + 1.5, self.state[1] + 0.25, self.state[2], self.state[3]]
        self.calls.append(("track", image.shape, None if info is None else sorted(info)))
        return {"target_bbox": list(self.state), "confidence": 0.5}


def _savetxt(data, fmt):
    b = io.BytesIO()
    np.savetxt(b, data, delimiter="\t", fmt=fmt)
    return b.getvalue().decode()


def test_result_file_formats_match_np_savetxt_byte_for_byte(tmp_path):
    from vittracker_amd.evaluation import results as R
    from vittracker_amd.evaluation.data import Sequence
    rs = np.random.RandomState(0)
    boxes = (rs.uniform(-30, 700, (40, 4))).tolist() + [[-0.9, 0.9, 10.999, 1e-3]]      # truncation toward zero
    times = rs.uniform(1e-4, 0.2, 41).tolist()
    assert R.format_boxes(boxes) == _savetxt(np.array(boxes).astype(int), "%d")
    assert R.format_floats(times) == _savetxt(np.array(times).astype(float), "%f")
    assert R.format_floats(times, "%.2f") == _savetxt(np.array(times).astype(float), "%.2f")
    for dataset, sub in (("synthetic", ""), ("got10k", "got10k"), ("trackingnet", "trackingnet")):
        seq = Sequence("seqA", [np.zeros((4, 4, 3), np.uint8)] * 41, dataset, np.array(boxes))
        out = {"target_bbox": boxes, "time": times, "all_boxes": None}
        written = R.save_tracker_output(seq, str(tmp_path / "res"), out)
        base = tmp_path / "res" / sub / "seqA"
        assert sorted(written) == sorted([str(base) + ".txt", str(base) + "_time.txt"])
        assert open(str(base) + ".txt").read() == _savetxt(np.array(boxes).astype(int), "%d")
        assert open(str(base) + "_time.txt").read() == _savetxt(np.array(times), "%f")
        assert R.results_exist(str(tmp_path / "res"), seq)


def test_track_sequence_loop_and_output_lists(monkeypatch, tmp_path):
    from vittracker_amd.evaluation import Tracker, get_dataset
    monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path))
    ds = get_dataset("synthetic:2x6")
    t = Tracker("vit_dist", "vit_48_h32_noKD", "synthetic", run_id=3)
    assert t.results_dir.endswith("test/tracking_results/vit_dist/vit_48_h32_noKD_003")
    fake = _FakeTracker(_Params(), "synthetic")
    out = t._track_sequence(fake, ds[1], ds[1].init_info())
    n = len(ds[1])
    assert len(out["target_bbox"]) == n and len(out["time"]) == n and all(x >= 0 for x in out["time"])
    assert out["target_bbox"][0] == ds[1].init_info()["init_bbox"]                 # frame 0 = the init box
    assert out["target_bbox"][3][0] == pytest.approx(out["target_bbox"][0][0] + 4.5)
    assert fake.calls[0][0] == "init" and fake.calls[1] == ("track", (240, 320, 3), ["gt_bbox", "previous_output"])


def test_run_dataset_sequential_writes_files_and_skips_existing(monkeypatch, tmp_path, capsys):
    from vittracker_amd.evaluation import Tracker, get_dataset
    from vittracker_amd.evaluation.running import run_dataset
    monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path))
    ds = get_dataset("synthetic:3x5")
    t = Tracker("vit_dist", "vit_48_h32_noKD", "synthetic")
    t.tracker_class = _FakeTracker
    t.get_parameters = lambda: _Params()
    run_dataset(ds, [t], debug=False, threads=0)
    for s in ds:
        rows = open(os.path.join(t.results_dir, s.name + ".txt")).read().splitlines()
        assert len(rows) == len(s) and all(len(r.split("\t")) == 4 for r in rows)
        assert len(open(os.path.join(t.results_dir, s.name + "_time.txt")).read().splitlines()) == len(s)
    capsys.readouterr()
    run_dataset(ds, [t], debug=False, threads=0)                                   # second run: results exist
    assert capsys.readouterr().out.count("FPS: -1") == 3


def test_run_video_headless(monkeypatch, tmp_path):
    from vittracker_amd.evaluation import Tracker
    monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path))
    vid = np.random.RandomState(0).randint(0, 256, (7, 48, 64, 3)).astype(np.uint8)
    np.save(tmp_path / "clip.npy", vid)
    t = Tracker("vit_dist", "vit_48_h32_noKD", "video")
    t.tracker_class = _FakeTracker
    boxes = t.run_video(str(tmp_path / "clip.npy"), optional_box=[10.0, 12.0, 8.0, 6.0], save_results=True, params=_Params())
    assert len(boxes) == 7 and boxes[0] == [10.0, 12.0, 8.0, 6.0] and boxes[2] == [13, 12, 8, 6]   # int() per frame
    txt = open(os.path.join(t.results_dir, "video_clip.txt")).read()
    assert txt == _savetxt(np.array(boxes).astype(int), "%d")
    with pytest.raises(ValueError, match="optional_box"):
        t.run_video(str(tmp_path / "clip.npy"), params=_Params())


class _FakeBatched:
    """Stands in for BatchedVitTracker: every sequence's box moves +2 in x per step."""
    instances = []

    def __init__(self, params, batch):
        self.B, self.steps = batch, 0
        _FakeBatched.instances.append(self)

    def initialize(self, frames, boxes):
        assert frames.shape[0] == self.B and frames.dtype == np.uint8
        self.state = np.asarray(boxes, dtype=np.float64).copy()

    def track(self, frames, sync=True):
        import torch
        assert frames.shape[0] == self.B
        self.steps += 1
        self.state[:, 0] += 2.0
        return {"target_bbox": torch.from_numpy(self.state.copy()), "confidence": torch.zeros(self.B)}

    def track_chunk(self, frames, sync=True):
        import torch
        assert frames.ndim == 5 and frames.shape[1] == self.B
        self.chunks = getattr(self, "chunks", 0) + 1
        rows = []
        for _ in range(frames.shape[0]):
            self.steps += 1
            self.state[:, 0] += 2.0
            rows.append(self.state.copy())
        return {"target_bbox": torch.from_numpy(np.stack(rows)), "confidence": torch.zeros(frames.shape[0], self.B)}


def test_batched_runner_ragged_lengths_and_rank_sharding(monkeypatch, tmp_path):
    from vittracker_amd.evaluation import Tracker, get_dataset
    from vittracker_amd.evaluation.running import run_dataset_batched
    monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path))
    ds = get_dataset("synthetic:7x5")            # lengths 5,7,9,5,7,9,5
    t = Tracker("vit_dist", "vit_48_h32_noKD", "synthetic")
    _FakeBatched.instances.clear()
    out0 = run_dataset_batched(ds, t, batch=3, rank=0, world=2, params=_Params(), make_batched=_FakeBatched)
    out1 = run_dataset_batched(ds, t, batch=3, rank=1, world=2, params=_Params(), make_batched=_FakeBatched)
    assert sorted(out0) == [ds[i].name for i in (0, 2, 4, 6)] and sorted(out1) == [ds[i].name for i in (1, 3, 5)]
    assert [b.B for b in _FakeBatched.instances] == [3, 1, 3]                       # rank 0: groups of 3 + 1; rank 1: 3
    for s in ds:
        rows = [list(map(int, r.split("\t"))) for r in open(os.path.join(t.results_dir, s.name + ".txt")).read().splitlines()]
        gt0 = [int(v) for v in s.init_info()["init_bbox"]]
        assert len(rows) == len(s) and rows[0] == gt0
        assert rows[-1][0] == gt0[0] + 2 * (len(s) - 1)                             # no extra steps leak into a short sequence
        assert len(open(os.path.join(t.results_dir, s.name + "_time.txt")).read().splitlines()) == len(s)
    # everything exists now: nothing left to do
    assert run_dataset_batched(ds, t, batch=3, params=_Params(), make_batched=_FakeBatched) == {}


def test_batched_runner_frames_per_launch_writes_the_same_files(monkeypatch, tmp_path):
    """frames_per_launch = 3: whole chunks through track_chunk, the tail frame by frame; same rows as one frame per launch."""
    from vittracker_amd.evaluation import Tracker, get_dataset
    from vittracker_amd.evaluation.running import run_dataset_batched
    ds = get_dataset("synthetic:4x5")            # lengths 5,7,9,5
    texts = []
    for n in (1, 3):
        monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path / f"n{n}"))
        t = Tracker("vit_dist", "vit_48_h32_noKD", "synthetic")
        _FakeBatched.instances.clear()
        run_dataset_batched(ds, t, batch=4, params=_Params(), make_batched=_FakeBatched, frames_per_launch=n)
        fb = _FakeBatched.instances[0]
        assert fb.steps == 8 and getattr(fb, "chunks", 0) == (0 if n == 1 else 2)      # T = 9: 8 steps = 2 chunks of 3 + 2 single
        texts.append([open(os.path.join(t.results_dir, s.name + ".txt")).read() for s in ds])
        for s in ds:
            assert len(open(os.path.join(t.results_dir, s.name + "_time.txt")).read().splitlines()) == len(s)
    assert texts[0] == texts[1]


def test_batched_runner_isolates_bad_sequences_and_failing_groups(monkeypatch, tmp_path, capsys):
    """The reference prints the error and skips only that sequence (lib/test/evaluation/running.py:138-142).  Lock-step form: a
    too-small init box is screened out per sequence before grouping; a group that raises mid-run is reported and skipped while
    the groups before and after it are written; one pipeline per batch size is built and reused."""
    from vittracker_amd.evaluation import Tracker, get_dataset
    from vittracker_amd.evaluation.running import run_dataset_batched
    monkeypatch.setenv("VITTRACK_SAVE_DIR", str(tmp_path))
    ds = get_dataset("synthetic:10x5")
    ds[1].ground_truth_rect[0, 2:] = 0.0                  # zero-size init box -> "Too small bounding box."
    t = Tracker("vit_dist", "vit_48_h32_noKD", "synthetic")

    class Flaky(_FakeBatched):
        calls = 0

        def initialize(self, frames, boxes):
            Flaky.calls += 1
            if Flaky.calls == 2:
                raise RuntimeError("boom in group 2")
            super().initialize(frames, boxes)
    _FakeBatched.instances.clear()
    out = run_dataset_batched(ds, t, batch=3, params=_Params(), make_batched=Flaky)      # 9 good sequences -> groups of 3, 3, 3
    txt = capsys.readouterr().out
    assert "Too small bounding box." in txt and ds[1].name in txt and "boom in group 2" in txt
    good = [s for i, s in enumerate(ds) if i != 1]
    done = [s.name for s in good[:3] + good[6:]]
    assert sorted(out) == sorted(done)
    for s in ds:
        assert os.path.exists(os.path.join(t.results_dir, s.name + ".txt")) == (s.name in done)
    assert [b.B for b in _FakeBatched.instances] == [3, 3]      # built once, reused for group 2 (which failed), rebuilt for group 3


def test_deploy_wire_signature():
    from vittracker_amd import deploy
    ins, outs = deploy.wire_signature()
    assert [(a.name, a.shape) for a in ins] == [("template", [1, 3, 128, 128]), ("search", [1, 3, 256, 256])]
    assert [(a.name, a.shape) for a in outs] == [("output1", [1, 1, 16, 16]), ("output2", [1, 2, 16, 16]), ("output3", [1, 2, 16, 16])]


def test_cli_scripts_parse(tmp_path):
    import subprocess
    import sys
    for script in ("test.py", "video_demo.py", "profile_model_cpu.py"):
        r = subprocess.run([sys.executable, os.path.join(REPO, "tracking", script), "--help"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "usage" in r.stdout.lower(), script
