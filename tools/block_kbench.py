#!/usr/bin/env python3
"""HIP-event timing of the raw block calls (smx_block_forward / smx_block_backward), for tuning."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import functional as fn
from kbench import timeit

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64x4096x256x128")
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
B, N, D, F = map(int, args.shape.split("x"))
dev = torch.device("cuda:0")
x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
lw = torch.randn(D, device=dev); lb = torch.randn(D, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
y, xk, st = fn.block_forward_raw(x, lw, lb, 1e-5, wr, wi, bias)
f = timeit(lambda: fn.block_forward_raw(x, lw, lb, 1e-5, wr, wi, bias), args.iters)
b = timeit(lambda: fn.block_backward_raw(g, x, st, lw, xk, wr, wi), args.iters)
print(json.dumps({"lib": os.path.basename(os.environ.get("SMX_LIB", "libsmx.so")), "shape": args.shape,
                  "blk_fwd_us": round(f[0] * 1e3, 1), "blk_fwd_min": round(f[1] * 1e3, 1),
                  "blk_bwd_us": round(b[0] * 1e3, 1), "blk_bwd_min": round(b[1] * 1e3, 1)}))
