#!/usr/bin/env python3
"""Does the relative placement of x / y / g / grad_x in device memory change the step time?
Carves the four (B,N,D) tensors out of one arena at controlled offsets and times raw fwd+bwd."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn
import ctypes
dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 256, 128
n = B * N * D
arena = torch.empty(6 * n + (64 << 20), dtype=torch.float32, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
xk = torch.empty(B, F, D, dtype=torch.complex64, device=dev)
flat = torch.empty(2 * D * F + D, device=dev)
ws = torch.empty(_lib.workspace_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def view(off_floats):
    return arena[off_floats:off_floats + n].view(B, N, D)
def run(pads, iters=200):
    offs = [0]
    for p in pads: offs.append(offs[-1] + n + p // 4)
    x, y, g, gx = (view(o) for o in offs)
    x.normal_(); g.normal_()
    def step():
        lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(), xk.data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)
        lib.smx_backward(g.data_ptr(), xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), gx.data_ptr(), flat.data_ptr(), flat[D*F:].data_ptr(), flat[2*D*F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 3, st)
    for _ in range(300): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6, x.data_ptr()
for rep in range(2):
    for pads in [(0, 0, 0), (256, 256, 256), (4096, 4096, 4096), (65536 + 256,) * 3, ((1 << 20) + 4352,) * 3, ((2 << 20) + 256,) * 3,
                 (0, 1 << 20, 0), (128, 128, 128), (8192 + 512,) * 3, ((16 << 20),) * 3]:
        us, p = run(pads)
        print(f"pads {pads}  base {p & 0xFFFFFFF:#x}: {us:.1f} us/step  frac {16 * n / (us * 1e-6) / 8e12:.3f}", flush=True)
