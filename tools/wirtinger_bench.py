#!/usr/bin/env python3
"""Timing of the standalone complex-in / complex-out Wirtinger API (wirtinger_ops.py: WirtingerSpectralFilter,
WirtingerGradient) at the C5 size -- elementwise / reduction kernels, reported against the bytes they move."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib
from tensor_cuda_fft_amd.functional import _stream


def timeit(f, iters=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    B, N, D, F = 64, 4096, 512, 256
    k = min(F, N // 2)
    lib = _lib.lib()
    x = torch.randn(B, N, D, device=dev, dtype=torch.complex64)
    g = torch.randn(B, N, D, device=dev, dtype=torch.complex64)
    out = torch.empty_like(x)
    wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev)
    gwr = torch.empty_like(wr); gwi = torch.empty_like(wi)
    s = _stream(dev)
    full = B * N * D * 8
    kept = B * k * D * 8
    t = timeit(lambda: lib.smx_wfilter_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), out.data_ptr(), B, N, D, F, 0, s))
    print(json.dumps({"op": "smx_wfilter_forward", "shape": [B, N, D, F], "ms": round(t, 4),
                      "GBps": round((full + kept) / t / 1e6, 1), "bytes": "write (B,N,D) c64 + read kept rows"}))
    t = timeit(lambda: lib.smx_wfilter_grad_w(x.data_ptr(), g.data_ptr(), gwr.data_ptr(), gwi.data_ptr(), B, N, D, F, s))
    print(json.dumps({"op": "smx_wfilter_grad_w", "ms": round(t, 4), "GBps": round(2 * kept / t / 1e6, 1),
                      "bytes": "read kept rows of x and g"}))
    xs = x[:, :k].contiguous(); gs = g[:, :k].contiguous(); os_ = torch.empty_like(xs)
    w = torch.randn(1, k, D, device=dev, dtype=torch.complex64); gw = torch.empty_like(w)
    t = timeit(lambda: lib.smx_cmul(xs.data_ptr(), w.data_ptr(), os_.data_ptr(), B, k * D, 0, s))
    print(json.dumps({"op": "smx_cmul (B,k,D)", "ms": round(t, 4), "GBps": round(2 * kept / t / 1e6, 1)}))
    t = timeit(lambda: lib.smx_cmul_grad_w(xs.data_ptr(), gs.data_ptr(), gw.data_ptr(), B, k * D, s))
    print(json.dumps({"op": "smx_cmul_grad_w (B,k,D)", "ms": round(t, 4), "GBps": round(2 * kept / t / 1e6, 1)}))


if __name__ == "__main__":
    main()
