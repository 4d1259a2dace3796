#!/usr/bin/env python3
"""Per-chunk step time over several seconds of back-to-back fwd+bwd: looks for clock/power modes."""
import os, sys, time, subprocess, torch
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
x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev); y = torch.empty_like(x); gx = torch.empty_like(x)
def step():
    lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(), xk.data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)
    lib.smx_backward(g.data_ptr(), xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), gx.data_ptr(), flat.data_ptr(), flat[D*F:].data_ptr(), flat[2*D*F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 3, st)
out = []
t_start = time.perf_counter()
for chunk in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
    out.append(us)
    if chunk in (20, 40):
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            print("\n".join(l for l in r.splitlines() if "clk" in l or "Power" in l or "Temp" in l), flush=True)
        except Exception as e: print("smi failed", e)
print(" ".join(f"{u:.0f}" for u in out))
