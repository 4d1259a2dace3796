#!/usr/bin/env python3
"""Timing of the fused block half, y = x + mix(LayerNorm(x)) fwd+bwd, against the composition of
torch LayerNorm + smx mix + torch add (what SpectralMLPBlock did before the fusion).

Algorithmic bytes of the block half: fwd read x + write y, bwd read g + read x + write grad_x
= 20 B/sample (the LayerNorm backward needs x again).  Reported: ms per fwd+bwd, GSamples/s and the
fraction of 8 TB/s at 20 B/sample, for both variants, inside one hipGraph each.
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import functional as fn


def graph_time(step, iters, reps=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64x4096x256x128")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    B, N, D, F = map(int, args.shape.split("x"))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
    lw = 1 + 0.3 * torch.randn(D, device=dev); lb = 0.2 * torch.randn(D, device=dev)
    wr = 1 + 0.5 * torch.randn(D, F, device=dev); wi = 0.5 * torch.randn(D, F, device=dev)
    bias = 0.1 * torch.randn(D, device=dev)

    def fused():
        y, xk, st = fn.block_forward_raw(x, lw, lb, 1e-5, wr, wi, bias)
        fn.block_backward_raw(g, x, st, lw, xk, wr, wi)

    leaves = [t.requires_grad_(True) for t in (x.clone(), lw.clone(), lb.clone(), wr.clone(),
                                               wi.clone(), bias.clone())]

    def unfused():
        xx, a, b, c, d, e = leaves
        y = xx + fn.spectral_mix(torch.nn.functional.layer_norm(xx, (D,), a, b, 1e-5), c, d, e)
        y.backward(g)
        for t in leaves:
            t.grad = None

    smp = B * N * D
    for name, f in (("fused_block", fused), ("unfused_composition", unfused)):
        if args.only and args.only != name:
            continue
        ms = graph_time(f, args.iters)
        print(json.dumps({"what": name, "shape": args.shape, "ms_fwd_bwd": round(ms, 4),
                          "GSamples_s": round(smp / ms / 1e6, 1),
                          "roofline_frac_20B": round(20 * smp / (ms * 1e-3) / 8e12, 3)}), flush=True)


if __name__ == "__main__":
    main()
