#!/usr/bin/env python3
"""HIP-event timing of the non-transform pieces of the twin blocks (smx_time.hip) at the FrequencyNativeBlock benchmark
shape: planar multiply, SpectralLayerNorm, planar fold / split, the gate chain -- with the bytes each moves."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import functional as fn

B, Fq, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 513, 512)
dev = torch.device("cuda:0")
torch.manual_seed(0)
H = 2 * C


def timeit(name, f, nbytes, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:34s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


h = torch.randn(2, B, Fq, H, device=dev)
fr, fi = torch.randn(Fq, H, device=dev), torch.randn(Fq, H, device=dev)
g = torch.randn_like(h)
pb = h.numel() * 4
timeit("planar_cmul fwd (H = 2C)", lambda: fn._PlanarCmul.apply(h, fr, fi), 2 * pb)
hr, frr, fir = h.clone().requires_grad_(True), fr.clone().requires_grad_(True), fi.clone().requires_grad_(True)
y = fn.planar_cmul(hr, frr, fir)
timeit("planar_cmul bwd", lambda: torch.autograd.grad(y, (hr, frr, fir), g, retain_graph=True), 3 * pb)

z = torch.randn(B, Fq, C, dtype=torch.complex64, device=dev)
gz = torch.randn_like(z)
gamma, beta = torch.randn(Fq, C, device=dev), torch.randn(Fq, C, device=dev)
zb = z.numel() * 8
for planar in (False, True):
    zr, gr_, br_ = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    timeit(f"spectral_ln fwd planar={int(planar)}", lambda: fn.spectral_layer_norm(z, gamma, beta, 1e-5, planar=planar), 2 * zb)
    o = fn.spectral_layer_norm(zr, gr_, br_, 1e-5, planar=planar)
    go = torch.randn_like(o)
    timeit(f"spectral_ln bwd planar={int(planar)}", lambda: torch.autograd.grad(o, (zr, gr_, br_), go, retain_graph=True), 3 * zb)
pl = torch.randn(2, B, Fq, C, device=dev)
timeit("add_planar", lambda: fn.add_planar(z, pl), 3 * zb)
a = torch.randn(Fq, dtype=torch.complex64, device=dev)
u, p, q = torch.randn(C, device=dev), torch.rand(Fq, device=dev), torch.rand(B, C, device=dev)
m = torch.ones(Fq, device=dev); m[300:] = 0
timeit("spectral_gate fwd", lambda: fn.spectral_gate(z, a, u, p, q, m), 2 * zb)
lv = [t.clone().requires_grad_(True) for t in (z, a, u, p, q)]
yg = fn.spectral_gate(*lv, m, reference_gain_grad=True)
timeit("spectral_gate bwd", lambda: torch.autograd.grad(yg, lv, gz, retain_graph=True), 3 * zb)
w = torch.randn(Fq, C, dtype=torch.complex64, device=dev)
timeit("torch copy of z (yardstick)", lambda: z.clone(), 2 * zb)
