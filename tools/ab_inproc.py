#!/usr/bin/env python3
"""A/B of several builds of libsmx.so INSIDE ONE PROCESS: same clocks, same allocations, same process-level
"mode"  for every candidate; launches interleaved round by round, HIP events.

    python tools/ab_inproc.py --libs libsmx.so,libsmx_pk1.so --shapes 64x4096x256x128,64x4096x512x256

Prints per lib and shape the median / min of the forward and backward transform launches (no k_gradw: phases
SPECTRUM | INVERSE) and checks that the candidates agree with the first lib to 1e-5 (max-normalised)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tensor_cuda_fft_amd as pkg                                   # noqa: E402
from tensor_cuda_fft_amd import _lib, functional as fn              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="libsmx.so")
    ap.add_argument("--shapes", default="64x4096x256x128")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--full", action="store_true", help="backward with PHASE_ALL (includes k_gradw)")
    ap.add_argument("--clean", action="store_true", help="backward on a workspace of its own with sync_clean=True "
                    "(what the autograd functions do: the library skips clearing the flag words)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    csrc = os.path.join(ROOT, "tensor-cuda-fft-_amd", "csrc")
    # a candidate is "lib.so" or "lib.so:knob=value;knob=value" (scoped plan options, _lib.options)
    names = args.libs.split(",")
    files = {n: n.split(":")[0] for n in names}
    knobs = {n: dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in n.split(":")[1].split(";")) if ":" in n else {}
             for n in names}
    loaded = {}
    for f in set(files.values()):
        loaded[f] = _lib.load(f if os.path.isabs(f) else os.path.join(csrc, f))
    ctx = [None]

    def use(n):
        if ctx[0] is not None:
            ctx[0].__exit__(None, None, None)
            ctx[0] = None
        _lib._lib = loaded[files[n]]
        if knobs[n]:
            ctx[0] = _lib.options(**knobs[n])
            ctx[0].__enter__()

    for sh in args.shapes.split(","):
        B, N, D, F = map(int, sh.split("x"))
        torch.manual_seed(0)
        x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
        wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
        gx = torch.empty_like(g)
        flat = torch.empty(2 * D * F + D, device=dev)
        wsb = torch.zeros(fn._ws_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
        ph = fn.PHASE_ALL if args.full else (fn.PHASE_SPECTRUM | fn.PHASE_INVERSE)
        ref = None
        res = {n: {"fwd": [], "bwd": []} for n in names}
        xks = {}
        for n in names:                                              # warm-up + agreement
            use(n)
            pack = fn._new_pack(x, wr)
            y, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True, pack=pack)
            fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_ALL, grad_x=gx, flat=flat, pack=pack)
            torch.cuda.synchronize()
            out = (y.clone(), gx.clone(), flat.clone())
            xks[n] = (xk, pack)
            if ref is None:
                ref = out
            else:
                errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref)]
                print(json.dumps({"shape": sh, "lib": n, "max_norm_diff_vs_first": errs}), flush=True)
        for _ in range(args.rounds):
            for n in names:
                use(n)
                xk, pack = xks[n]
                for key, f in (("fwd", lambda: fn.forward_raw(x, wr, wi, bias, save_spectrum=True, pack=pack,
                                                              pack_ready=pack is not None)),
                               ("bwd", lambda: fn.backward_raw(g, xk, wr, wi, phases=ph, grad_x=gx, flat=flat,
                                                               pack=pack, ws=wsb if args.clean else None,
                                                               sync_clean=args.clean))):
                    f(); f()
                    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                           for _ in range(args.iters)]
                    for a, b in evs:
                        a.record(); f(); b.record()
                    torch.cuda.synchronize()
                    res[n][key] += [a.elapsed_time(b) * 1e3 for a, b in evs]
        for n in names:
            o = {"shape": sh, "lib": n}
            for key in ("fwd", "bwd"):
                t = sorted(res[n][key])
                o[key + "_med_us"] = round(t[len(t) // 2], 1)
                o[key + "_min_us"] = round(t[0], 1)
            o["sum_med_us"] = round(o["fwd_med_us"] + o["bwd_med_us"], 1)
            print(json.dumps(o), flush=True)
        use(names[0])
        del x, g, gx
        fn.release_workspaces()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
