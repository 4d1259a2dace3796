#!/usr/bin/env python3
"""Many fresh allocations of the four tensors in one process: how often is a step fast, and does the
mode follow the allocation?  Also re-times the same buffers twice to separate allocation from time."""
import os, sys, time, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib
dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 256, 128
n = B * N * D
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
xk = torch.empty(B, F, D, dtype=torch.complex64, device=dev)
flat = torch.empty(2 * D * F + D, device=dev)
ws = torch.empty(_lib.workspace_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(x, y, g, gx, iters=150):
    def step():
        lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(), xk.data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)
        lib.smx_backward(g.data_ptr(), xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), gx.data_ptr(), flat.data_ptr(), flat[D*F:].data_ptr(), flat[2*D*F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 3, st)
    for _ in range(200): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
random.seed(1)
junk = []
for trial in range(24):
    if trial % 2 == 1:   # perturb the driver's free list between allocations
        junk = [torch.empty(random.randint(1, 300) << 20, dtype=torch.uint8, device=dev) for _ in range(random.randint(1, 4))]
    ts = [torch.empty(B, N, D, device=dev) for _ in range(4)]
    for t in ts: t.normal_()
    order = list(range(4)); random.shuffle(order)
    x, y, g, gx = (ts[i] for i in order)
    t1 = timed(x, y, g, gx); t2 = timed(x, y, g, gx)
    # same four buffers, roles swapped
    t3 = timed(y, x, gx, g)
    print(f"trial {trial:2d}: {t1:.1f} {t2:.1f} | swapped roles {t3:.1f} us   x {x.data_ptr()>>20:#x} y {y.data_ptr()>>20:#x} g {g.data_ptr()>>20:#x} gx {gx.data_ptr()>>20:#x}", flush=True)
    del ts, x, y, g, gx, junk
    junk = []
    torch.cuda.empty_cache()
