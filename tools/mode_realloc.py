#!/usr/bin/env python3
"""Does the bimodal step time follow the ALLOCATION?  One process, several trials: fresh
x / g / layer output buffers per trial (earlier ones kept alive or freed), the fwd+bwd step captured in a hipGraph
and timed with HIP events.  Prints ms/step and the device addresses of x and g per trial."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=8)
ap.add_argument("--keep", type=int, default=1, help="1: keep earlier buffers alive (new addresses every trial)")
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N, D = 64, 4096, 256
layer = pkg.SpectralMixingLayer(D).to(dev)
with torch.no_grad():
    layer.weight_real.normal_(1, 0.5); layer.weight_imag.normal_(0, 0.5); layer.bias.normal_(0, 0.1)
hold = []
for trial in range(args.trials):
    x = torch.randn(B, N, D, device=dev, requires_grad=True)
    g = torch.randn(B, N, D, device=dev)

    def step():
        y = layer(x)
        y.backward(g)
        x.grad = None
        layer.zero_grad(set_to_none=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            step()
    for _ in range(30):
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps // 10):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / (args.steps // 10 * 10)
    print(json.dumps({"trial": trial, "ms_per_step": round(ms, 4), "x": hex(x.data_ptr()), "g": hex(g.data_ptr())}), flush=True)
    if args.keep:
        hold.append((x, g, gr))
    else:
        del gr, x, g
        torch.cuda.empty_cache()
