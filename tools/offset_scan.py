#!/usr/bin/env python3
"""Does the forward launch depend on the RELATIVE placement of x and y?  One 2 GiB allocation, x at its start,
y at 768 MiB + delta for a range of deltas; raw smx_forward calls timed with HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn
from kbench import timeit

dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 256, 128
n = B * N * D
pool = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)     # 2 GiB
pool[:n].normal_()
x = pool[:n].view(B, N, D)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
ws = torch.empty(fn._ws_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
lib = _lib.lib()
base = 192 * 1024 * 1024            # floats: y starts 768 MiB into the pool
deltas = [0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 64,
          3 << 20, (1 << 22) + 4096 + 64]
res = []
for rep in range(2):
    for dl in deltas:
        y = pool[base + dl: base + dl + n].view(B, N, D)
        def f():
            _lib.check(lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                       None, ws.data_ptr(), ws.numel(), B, N, D, F, 0,
                                       torch.cuda.current_stream().cuda_stream))
        med, mn = timeit(f, 30)
        res.append((dl * 4, med * 1e3, mn * 1e3))
        print(f"rep {rep} delta {dl*4:>10d} B  fwd {med*1e3:6.1f} us (min {mn*1e3:6.1f})", flush=True)
