#!/usr/bin/env python3
"""Which kernel is slow in the slow placement mode?  Event-times fwd and bwd inside the alternating loop."""
import os, sys, time, random, itertools, torch
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
def fwd(x, y): lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(), xk.data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)
def bwd(g, gx): lib.smx_backward(g.data_ptr(), xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), gx.data_ptr(), flat.data_ptr(), flat[D*F:].data_ptr(), flat[2*D*F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 3, st)
def timed(x, y, g, gx, iters=100):
    for _ in range(150): fwd(x, y); bwd(g, gx)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    t0 = time.perf_counter()
    for e in ev:
        e[0].record(); fwd(x, y); e[1].record(); bwd(g, gx); e[2].record()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / iters * 1e6
    f = sum(e[0].elapsed_time(e[1]) for e in ev) / iters * 1e3
    b = sum(e[1].elapsed_time(e[2]) for e in ev) / iters * 1e3
    return tot, f, b
ndummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummies = [torch.empty(B, N, D, device=dev) for _ in range(ndummy)]     # allocated before the four tensors
ts = [torch.empty(B, N, D, device=dev) for _ in range(4)]
for t in ts: t.normal_()
print("VA (MiB):", [hex(t.data_ptr() >> 20) for t in ts])
for perm in [(0, 1, 2, 3), (1, 0, 3, 2), (2, 3, 0, 1), (3, 2, 1, 0), (1, 2, 3, 0), (0, 3, 2, 1)]:
    x, y, g, gx = (ts[i] for i in perm)
    tot, f, b = timed(x, y, g, gx)
    print(f"x={perm[0]} y={perm[1]} g={perm[2]} gx={perm[3]}: step {tot:.1f}  fwd {f:.1f}  bwd {b:.1f}", flush=True)
