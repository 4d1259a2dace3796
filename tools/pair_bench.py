#!/usr/bin/env python3
"""Timing of the two halves of the transform pair on their own (HIP events, median): functional.rfft_bins
(smx_rfft_ex) and functional._irfft_raw (smx_irfft_ex), with the algorithmic bytes of each -- the real tensor once
and the one-sided spectrum once -- beside torch.fft on the same GPU.  --opts selects plans ("fourstep=0")."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn


def timeit(f, iters, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="64x1024x512x2048,64x1024x256x1024,64x512x256x512,16x4096x256x8192,64x4096x256x4096")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--opts", default="")
ap.add_argument("--no-torch", action="store_true")
args = ap.parse_args()
for o in filter(None, args.opts.split(";")):
    k_, v_ = o.split("=")
    _lib.set_option(k_, int(v_))
for sh in args.shapes.split(","):
    B, T, D, n = map(int, sh.split("x"))
    k = n // 2 + 1
    x = torch.randn(B, T, D, device="cuda")
    X = fn.rfft_bins(x, k, n)
    nbytes = 4 * B * T * D + 8 * B * k * D
    p = _lib.plan_ex(_lib.smx_shape(B, T, D, k, n, k))
    rec = {"shape": sh, "plan": {"bands": p.bands, "groups": p.groups, "nsplit": p.nsplit}, "algorithmic_MB": round(nbytes / 1e6, 1)}
    for name, f in (("rfft", lambda: fn.rfft_bins(x, k, n)), ("irfft", lambda: fn._irfft_raw(X, n, T, 1.0 / n, True))):
        ms = timeit(f, args.iters)
        rec[name + "_ms"] = round(ms, 4)
        rec[name + "_frac_of_8TBps"] = round(nbytes / (ms * 1e-3) / 8e12, 3)
    if not args.no_torch:
        xp = torch.nn.functional.pad(x, (0, 0, 0, n - T))
        rec["torch_rfft_ms"] = round(timeit(lambda: torch.fft.rfft(xp, dim=1), 10), 4)
        rec["torch_irfft_ms"] = round(timeit(lambda: torch.fft.irfft(X, n=n, dim=1), 10), 4)
    print(json.dumps(rec), flush=True)
