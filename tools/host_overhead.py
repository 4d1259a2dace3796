#!/usr/bin/env python3
"""Host-side cost of one eager fwd+bwd of the layer (cProfile), to keep eager loops GPU-bound."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg

dev = torch.device("cuda:0")
B, N, D = (int(v) for v in (sys.argv[1:4] or (64, 4096, 256)))
layer = pkg.SpectralMixingLayer(D).to(dev)
x = torch.randn(B, N, D, device=dev, requires_grad=True); g = torch.randn(B, N, D, device=dev)
params = list(layer.parameters())
def step():
    y = layer(x); y.backward(g)
    x.grad = None
    for p in params: p.grad = None
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()          # host time to ISSUE 200 steps (queue may run ahead)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/200:.3f} ms/step, complete {1e3*(t2-t0)/200:.3f} ms/step")
# tiny problem: pure host cost
xs = torch.randn(2, 256, D, device=dev, requires_grad=True); gs = torch.randn(2, 256, D, device=dev)
def small():
    y = layer(xs); y.backward(gs); xs.grad = None
    for p in params: p.grad = None
for _ in range(20): small()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): small()
torch.cuda.synchronize()
print(f"small problem: {1e3*(time.perf_counter()-t0)/500:.3f} ms/step (host-bound)")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): small()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
