#!/usr/bin/env python3
"""Start phase of head_seq3_kernel (G256 head) by wave, from s_memtime stamps of a -DVT_SEQ3_START_STAMPS build placed at
build_variants/ss.so (cd vittracker_amd/csrc && hipcc ... -DVT_SEQ3_START_STAMPS -c vittrack.hip; link with vitb.o): entry -> token and
weight loads issued -> borders zeroed -> loads landed -> pieces written -> barrier.  NOTES R5-9."""
import os, sys
os.environ["VT_DBG_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from vittracker_amd import native, synth
native.LIB_PATH = "/root/repo/build_variants/ss.so"
B = 256
m = native.Model(128, 256, max_batch=B)
m.load_state_dict(synth.synth_state_dict(0, len_z=64, len_x=256))
z, x = synth.synth_inputs(1, B, 128, 256)
zd, xd = torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda()
for _ in range(3): m.forward(zd, xd)
torch.cuda.synchronize()
buf = np.zeros((B * 8 * 64,), dtype=np.uint64)
native._check(native.lib().vt_debug_stamps(m._h, B, buf.ctypes.data), "stamps")
st = buf.reshape(B, 8, 64).astype(np.int64)
d = np.diff(st[:, :, :8], axis=2)
names = ["entry -> loads issued", "halo loop", "wait vmcnt(0)", "split + writes", "barrier", "conv1...", "x"]
for k in range(6):
    print(names[k].ljust(24) + "".join(f"{d[:, w, k].mean():7.0f}" for w in range(8)))
# spread of entry times across workgroups
e = st[:, 0, 0]
print("entry time spread over workgroups (cycles): min 0, median %d, p90 %d, max %d" % (np.median(e - e.min()), np.percentile(e - e.min(), 90), (e - e.min()).max()))
