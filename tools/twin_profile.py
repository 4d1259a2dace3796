#!/usr/bin/env python3
"""fwd+bwd of ONE of the spectrum-domain twin blocks, for `rocprofv3 --kernel-trace --stats -- python3 tools/twin_profile.py
freqnative|bicameral|fixed [B T C K]`: where its time goes (GEMM / elementwise / native)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg

which = sys.argv[1] if len(sys.argv) > 1 else "freqnative"
B, T, C, K = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (64, 1024, 512, 128)
dev = torch.device("cuda:0")
torch.manual_seed(0)
cls = {"freqnative": pkg.FrequencyNativeBlock, "bicameral": pkg.BicameralBlock, "fixed": pkg.FixedSpectralBlock}[which]
blk = cls(C, seq_len=T, kernel_len=K, transition_bins=32, dropout=0.0).to(dev)
x = torch.randn(B, T, C, device=dev, requires_grad=True)
g = torch.randn(B, T, C, device=dev)
for _ in range(8):
    y = blk(x)
    y.backward(g)
    x.grad = None
    blk.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("done", which, B, T, C, K)
