#!/usr/bin/env python3
"""Is a (64,4096,512) launch slower than two (32,4096,512) launches because of the kernel or the footprint?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn
from kbench import timeit
dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 512, 128
x = torch.randn(B, N, D, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
y = torch.empty_like(x)
lib = _lib.lib()
def fwd(xs, ys):
    b = xs.shape[0]
    _lib.check(lib.smx_forward(xs.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), ys.data_ptr(), None,
                               None, 0, b, N, D, F, 0, torch.cuda.current_stream().cuda_stream))
for rep in range(2):
    print("whole (64)        : %.1f us" % (timeit(lambda: fwd(x, y))[0] * 1e3))
    print("first half (32)   : %.1f us" % (timeit(lambda: fwd(x[:32], y[:32]))[0] * 1e3))
    print("second half (32)  : %.1f us" % (timeit(lambda: fwd(x[32:], y[32:]))[0] * 1e3))
    print("both halves       : %.1f us" % (timeit(lambda: (fwd(x[:32], y[:32]), fwd(x[32:], y[32:])))[0] * 1e3))
    xs = torch.randn(32, N, D, device=dev); ys = torch.empty_like(xs)
    print("separate 32 alloc : %.1f us" % (timeit(lambda: fwd(xs, ys))[0] * 1e3))
    del xs, ys
