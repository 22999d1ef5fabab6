#!/usr/bin/env python3
"""What does the per-unit result gather cost, and why?  One GPU, a 1-rank RCCL communicator (run under torch.distributed.run).
Times K graph launches of the 4-step record graph with, between launches: nothing | async all_gather | sync all_gather on the same
stream | a plain 20 KB device copy on a side stream behind an event | the all_gather every 4th launch only.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/gather_probe.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
B, S = 256, 4
r = bench.Runner("G128", B, steps_per_graph=S)
REC = 5 * B
res = [torch.empty(S * REC, device="cuda") for _ in range(2)]
gat = [torch.empty(S * REC, device="cuda") for _ in range(2)]
def outs(buf):
    o = []
    for j in range(S):
        x = r.native.Outputs(B, r.model.feat_sz, "cuda")
        x.hann_boxes = buf[j * REC:j * REC + 4 * B].view(B, 4); x.conf = buf[j * REC + 4 * B:(j + 1) * REC]
        o.append(x)
    return o
graphs = [r.model.capture_steps([r.z] * S, [r.x] * S, outs(res[k]))[0] for k in range(2)]
side = torch.cuda.Stream()
ev = [torch.cuda.Event() for _ in range(2)]
def run(mode, K=200):
    pend = [None, None]
    torch.cuda.synchronize()
    with torch.cuda.stream(r.stream):
        t0 = time.perf_counter()
        for i in range(K):
            k = i & 1
            if pend[k] is not None:
                pend[k].wait(); pend[k] = None
            graphs[k].launch(r.stream)
            if mode == "async":
                pend[k] = dist.all_gather_into_tensor(gat[k], res[k], async_op=True)
            elif mode == "sync":
                dist.all_gather_into_tensor(gat[k], res[k])
            elif mode == "every4" and i % 4 == 3:
                pend[k] = dist.all_gather_into_tensor(gat[k], res[k], async_op=True)
            elif mode == "sidecopy":
                ev[k].record(r.stream)
                with torch.cuda.stream(side):
                    side.wait_event(ev[k])
                    gat[k].copy_(res[k])
            elif mode == "samecopy":
                gat[k].copy_(res[k])
            elif mode == "eventonly":
                ev[k].record(r.stream)
        for p in pend:
            if p is not None: p.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (K * S) * 1e6
for _ in range(2):
    for mode in ("none", "async", "sync", "every4", "sidecopy", "samecopy", "eventonly", "none"):
        run(mode, 40)
        print(f"{mode:10s} {run(mode):7.2f} us per step", flush=True)
dist.destroy_process_group()
