#!/usr/bin/env python3
"""A few fwd+bwd steps at one shape, for rocprofv3 (kernel trace / PMC)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn
B, N, D, F = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x4096x256x128").split("x"))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for o in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(";")):
    k, v = o.split("="); _lib.set_option(k, int(v))
dev = torch.device("cuda:0")
x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
for _ in range(steps):
    y, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True)
    gx, flat = fn.backward_raw(g, xk, wr, wi)
torch.cuda.synchronize()
print("done")
