"""CPU emulation (torch, fp64 accumulate) of the ViT-Base block with bf16-rounded contraction operands, two ways:

  current : LayerNorm in f32 -> xn rounded to bf16 -> xn @ bf16(W)^T + b                    (layernorm_kernel + GEMM)
  fold    : raw residual rounded to bf16 -> x @ bf16(W')^T, y = rstd * acc + b'              (vb_gemm.h, round 5)
            W' = (W * gamma) (I - 11^T / K)  (k-centred rows: the mean subtraction lives in the weights),  b' = b + W beta

and reports each one's relative L2 error of the residual stream against the fp32 oracle, block by block, plus the same
with a common-mode offset added to the tokens (mean / std of a row up to `--offset`), where the fold is the weaker form.
Not part of the product or the tests; the numbers are quoted in NOTES.md (R5-1).

    python tools/vitb_lnfold_emul.py [--B 1] [--offset 0 2 8]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import vitb_oracle_torch as ob          # noqa: E402  (a tool, not the product)
from vittracker_amd import synth                     # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def centre_round(w64, feedback):
    """bf16 rounding of k-centred weight rows; feedback = adjust roundings so that each ROW's bf16 values sum to ~0."""
    r = bf(w64)
    if not feedback:
        return r
    r = r.clone()
    for _ in range(3):
        resid = r.sum(dim=1)                                             # what the common mode still sees
        # move the element whose rounding error has the sign of the residual by one bf16 ulp, greedily, largest residuals first
        err = r - w64
        ulp = torch.pow(2.0, torch.floor(torch.log2(r.abs().clamp_min(1e-30))) - 7)
        idx = torch.argmax(err * torch.sign(resid)[:, None] / ulp, dim=1)   # most over-rounded in the residual's direction
        rows = torch.arange(r.shape[0])
        step = torch.sign(resid) * ulp[rows, idx]
        better = (resid - step).abs() < resid.abs()
        r[rows[better], idx[better]] -= step[better]
    return r


def block_emul(x, sd, i, mode, feedback=False):
    p = f"backbone.blocks.{i}."
    g = lambda k: torch.from_numpy(sd[p + k]).double()
    C, H, hd = 768, 12, 64

    def ln_linear(x, nw, nb, W, b):
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        if mode == "current":
            xn = ((x.float() - mean.float()) * rstd.float() * nw.float() + nb.float())          # f32 LayerNorm
            return bf(xn) @ bf(W).T + b
        Wg = W * nw[None, :]
        Wc = Wg - Wg.mean(dim=1, keepdim=True)
        bb = b + W @ nb
        return rstd * (bf(x.float()) @ centre_round(Wc, feedback).T) + bb

    Wqkv, bqkv = g("attn.qkv.weight"), g("attn.qkv.bias")
    qkv = ln_linear(x, g("norm1.weight"), g("norm1.bias"), Wqkv, bqkv)
    B, N, _ = x.shape
    qkv = bf(qkv.float()).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ao = bf((bf(a.float()) @ v).transpose(1, 2).reshape(B, N, C).float())
    x = (x + ao @ bf(g("attn.proj.weight")).T + g("attn.proj.bias")).float().double()           # f32 residual stream
    h = ln_linear(x, g("norm2.weight"), g("norm2.bias"), g("mlp.fc1.weight"), g("mlp.fc1.bias"))
    h = bf(torch.nn.functional.gelu(h.float()))
    return (x + h @ bf(g("mlp.fc2.weight")).T + g("mlp.fc2.bias")).float().double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--offset", type=float, nargs="*", default=[0.0, 2.0, 8.0])
    args = ap.parse_args()
    torch.manual_seed(0)
    sd = synth.synth_vitb_state_dict(26)
    z, x = synth.synth_inputs(3, args.B, 128, 256)
    orc = ob.build_from_state(sd).double()
    acts = {}
    with torch.no_grad():
        orc(torch.from_numpy(z).double(), torch.from_numpy(x).double(), acts)
    rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
    for off in args.offset:
        print(f"--- common-mode offset {off} x row std added to block 0's input")
        t0 = acts["tokens"].clone()
        t0 = t0 + off * t0.std(-1, keepdim=True)
        with torch.no_grad():
            ref = [t0]
            for i in range(args.blocks):
                ref.append(orc.backbone.blocks[i](ref[-1]))
            for mode, fb in (("current", False), ("fold", False), ("fold", True)):
                t = t0.float().double()
                errs = []
                for i in range(args.blocks):
                    t = block_emul(t, sd, i, mode, fb)
                    errs.append(rel(t, ref[i + 1]))
                one = [rel(block_emul(ref[i].float().double(), sd, i, mode, fb), ref[i + 1]) for i in (0, 5, 11) if i < args.blocks]
                print(f"{mode:8s} feedback={int(fb)}  chained rel-L2 after blocks 1/4/12: "
                      + " ".join(f"{errs[k]:.2e}" for k in (0, 3, args.blocks - 1) if k < args.blocks)
                      + "   single block 0/5/11: " + " ".join(f"{e:.2e}" for e in one))


if __name__ == "__main__":
    main()
