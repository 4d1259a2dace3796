#!/usr/bin/env python3
"""Kernel-level timing of the raw C-ABI calls (HIP events), for tuning.  Not the headline bench."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib, functional as fn

def timeit(f, iters=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="64x4096x256x128,8x65536x256x128,64x4096x512x256")
    ap.add_argument("--opts", default="")   # e.g. "stagger=0;nsplit=2"
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dropout", type=float, default=0.0, help="also time the launches with fused dropout")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for o in filter(None, args.opts.split(";")):
        k, v = o.split("="); _lib.set_option(k, int(v))
    # reference point: torch copy of the same bytes
    a = torch.randn(64, 4096, 256, device=dev); b = torch.empty_like(a)
    med, mn = timeit(lambda: b.copy_(a), args.iters)
    print(json.dumps({"what": "torch copy 256MiB", "ms": med, "min_ms": mn, "GBps": 2 * a.numel() * 4 / mn / 1e6}))
    for sh in args.shapes.split(","):
        B, N, D, F = map(int, sh.split("x"))
        x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
        wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
        p = _lib.plan(B, N, D, F)
        y, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True)
        f_med, f_min = timeit(lambda: fn.forward_raw(x, wr, wi, bias, save_spectrum=True), args.iters)
        b_med, b_min = timeit(lambda: fn.backward_raw(g, xk, wr, wi), args.iters)
        s_med, s_min = timeit(lambda: pkg.pruned_rfft(x, F), args.iters)
        smp = B * N * D
        if args.dropout > 0:
            rng = fn.DropoutState(dev).next()
            kw = dict(dropout_p=args.dropout, rng=rng)
            fd = timeit(lambda: fn.forward_raw(x, wr, wi, bias, save_spectrum=True, **kw), args.iters)
            bd = timeit(lambda: fn.backward_raw(g, xk, wr, wi, **kw), args.iters)
            td = timeit(lambda: torch.nn.functional.dropout(y, args.dropout, True), args.iters)
            print(json.dumps({"shape": sh, "dropout": args.dropout, "fwd_drop_us": round(fd[0] * 1e3, 1),
                              "bwd_drop_us": round(bd[0] * 1e3, 1), "fwd_us": round(f_med * 1e3, 1),
                              "bwd_us": round(b_med * 1e3, 1),
                              "torch_dropout_pass_us": round(td[0] * 1e3, 1)}), flush=True)
        print(json.dumps({"shape": sh, "opts": args.opts, "nsplit": p.nsplit, "path": p.path,
                          "fwd_ms": f_med, "fwd_min": f_min, "bwd_ms": b_med, "bwd_min": b_min,
                          "spec_ms": s_med,
                          "fwd_GBps": 8 * smp / f_min / 1e6, "bwd_GBps": 8 * smp / b_min / 1e6,
                          "fwdbwd_GS": smp / (f_med + b_med) / 1e6,
                          "roofline_frac": 16 * smp / ((f_med + b_med) * 1e-3) / 8e12}), flush=True)

if __name__ == "__main__":
    main()
