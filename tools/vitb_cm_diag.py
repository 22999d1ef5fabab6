import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_vitb_case, vitb_golden_files
from vittracker_amd import native
from oracle import vitb_oracle_torch as ob
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)
for path in vitb_golden_files():
    g, sd, z, x = load_vitb_case(path)
    m = native.Model(128, 256, channels=768, heads=12, depth=12, head_channels=256, max_batch=int(g["B"]))
    m.load_state_dict(sd)
    out = m.forward(torch.from_numpy(z).cuda(), torch.from_numpy(x).cuda())
    e = {k: float(np.abs(getattr(out, k).cpu().numpy() - g[k]).max()) for k in ("score_map", "size_map", "offset_map")}
    eb = float(np.abs(out.pred_boxes.cpu().numpy() - g["pred_boxes"][:, 0]).max()); eh = float(np.abs(out.hann_boxes.cpu().numpy() - g["hann_boxes"]).max())
    line = f"{os.path.basename(path):28s} cm {float(g['common_mode']) if 'common_mode' in g else 0:3.0f} maps {e['score_map']:.2e} {e['size_map']:.2e} {e['offset_map']:.2e} boxes {eb:.2e} {eh:.2e}"
    if "act_norm" in g:
        rows = g["act_rows"]
        orc = ob.build_from_state(sd); acts = {}
        with torch.no_grad(): orc(torch.from_numpy(z), torch.from_numpy(x), acts)
        r = []
        for k in (1, 4, 12):
            _, resid = m.blocks(acts["tokens"].cuda().contiguous(), nblocks=k, want_resid=True)
            got, want = resid[:1, rows].cpu().numpy(), g[f"act_block{k-1}"]
            # error relative to the rows' CENTRED norm (what LayerNorm sees) beside the plain relative L2
            wc = want - want.mean(-1, keepdims=True)
            r.append((rel(got, want), np.linalg.norm(got - want) / np.linalg.norm(wc), float(np.abs(want.mean(-1)).mean() / want.std(-1).mean())))
        line += "  blocks 1/4/12 rel-L2 " + " ".join(f"{a:.2e}" for a, _, _ in r) + " | vs centred " + " ".join(f"{b:.2e}" for _, b, _ in r) + f" | mean/sigma {r[0][2]:.2f} {r[2][2]:.2f}"
    print(line, flush=True)
