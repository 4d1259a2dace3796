#!/usr/bin/env python3
"""Eager fwd+bwd step time (HIP events) of the spectrum-domain twin blocks at (64,1024,512), K = 128 -- the in-situ
figure: every kernel meets its neighbours' cache state.  `SMX_LIB=<lib.so> python3 tools/twin_bench.py [blocks...]`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg

names = [a for a in sys.argv[1:] if not a.isdigit()] or ["freqnative", "bicameral"]
B, T, C, K = 64, 1024, 512, 128
dev = torch.device("cuda:0")
out = []
for which in names:
    torch.manual_seed(0)
    cls = {"freqnative": pkg.FrequencyNativeBlock, "bicameral": pkg.BicameralBlock, "fixed": pkg.FixedSpectralBlock}[which]
    blk = cls(C, seq_len=T, kernel_len=K, transition_bins=32, dropout=0.0).to(dev)
    x = torch.randn(B, T, C, device=dev, requires_grad=True)
    g = torch.randn(B, T, C, device=dev)

    def step():
        y = blk(x)
        y.backward(g)
        x.grad = None
        blk.zero_grad(set_to_none=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10)
    out.append(f"{which} {min(best):.3f} (" + " ".join(f"{b:.3f}" for b in best) + ")")
    del blk, x, g
print(os.path.basename(os.environ.get("SMX_LIB", "libsmx.so")), "ms/step:", " | ".join(out), flush=True)
