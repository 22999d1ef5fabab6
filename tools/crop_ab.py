#!/usr/bin/env python3
"""The device crop alone, uint8-patch and fp32 forms, per library build: python tools/crop_ab.py [lib.so ...]  (default: the in-tree build
plus every build_variants/cropdbg*.so -- the -DVT_CROPF_DBG timing builds of crop_fast_kernel, wrong results by design).
20 launches per captured graph, HIP events around replays: us per launch without the eager launch floor."""
import glob
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from vittracker_amd import native, synth
    B, H, W = 256, 480, 640
    rs = np.random.RandomState(0)
    frames = torch.from_numpy(rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)).cuda()
    frames2 = frames.flip(0).contiguous()       # a second 236 MB set: alternating between the two defeats the 256 MB Infinity Cache
    cold = bool(os.environ.get("CROP_AB_COLD"))
    boxes = np.stack([rs.uniform(50, W - 150, B), rs.uniform(50, H - 150, B), rs.uniform(30, 90, B), rs.uniform(30, 90, B)], 1)
    st = torch.tensor(boxes, dtype=torch.float64).cuda()
    m = native.Model(64, 128, max_batch=B)
    m.load_state_dict(synth.synth_state_dict(0, len_z=16, len_x=64))
    MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    res = []
    for T in (128, 256):
        for form in ("u8", "f32"):
            out = torch.empty(B, T, T, 3, dtype=torch.uint8, device="cuda") if form == "u8" else torch.empty(B, 3, T, T, device="cuda")
            rf = torch.empty(B, dtype=torch.float64, device="cuda")
            call = (lambda s=None, f=frames: m.crop_u8(f, st, 4.0, T, out=out, resize_factor=rf, stream=s)) if form == "u8" else \
                   (lambda s=None, f=frames: m.crop(f, st, 4.0, T, MEAN, STD, out=out, resize_factor=rf, stream=s))
            call()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side):
                for i in range(20):
                    call(torch.cuda.current_stream(), frames2 if (cold and (i & 1)) else frames)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(f"T={T} {form}: {e0.elapsed_time(e1) * 1000 / 200:6.2f} us")
    print("  ".join(res))


if __name__ == "__main__":
    if os.environ.get("CROP_AB_CHILD"):
        child()
        sys.exit(0)
    libs = sys.argv[1:] or ([""] + sorted(glob.glob(os.path.join(ROOT, "build_variants", "cropdbg*.so"))))
    for _ in range(2):
        for lib in libs:
            env = dict(os.environ, CROP_AB_CHILD="1", VT_CROP_BYTES="0")      # the timing builds fail the device self test by design: force the fast form
            if lib:
                env["VITTRACK_LIB"] = lib
            p = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
            print(f"{os.path.basename(lib) or 'in-tree':16s} {p.stdout.strip() or p.stderr[-400:]}", flush=True)
