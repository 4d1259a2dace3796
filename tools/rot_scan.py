#!/usr/bin/env python3
"""Scan of the experimental residue-rotation lattice rot = (l2*a + dt*b) % lc (placement = 3 | a<<8 | b<<16)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib, functional as fn
from kbench import timeit

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64x4096x512x128")
ap.add_argument("--a", default="1,2,3,4,5,7,8")
ap.add_argument("--b", default="0,1,2,3,4,5,7,8")
args = ap.parse_args()
B, N, D, F = map(int, args.shape.split("x"))
dev = torch.device("cuda:0")
x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
res = []
def run(tag, pl):
    _lib.set_option("placement", pl)
    y, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True)
    f = timeit(lambda: fn.forward_raw(x, wr, wi, bias, save_spectrum=True), 12)
    b = timeit(lambda: fn.backward_raw(g, xk, wr, wi), 12)
    res.append((f[0] + b[0], tag, f[0], b[0]))
run("default(2)", 2)
for a in map(int, args.a.split(",")):
    for b in map(int, args.b.split(",")):
        run(f"a={a} b={b}", 3 | (a << 8) | (b << 16))
run("default(2) again", 2)
res.sort()
for t, tag, f, b in res[:12]:
    print(f"{args.shape} {tag:18s} fwd {f*1e3:6.1f} bwd {b*1e3:6.1f} sum {t*1e3:6.1f}")
for t, tag, f, b in res:
    if tag.startswith("default"):
        print(f"{args.shape} {tag:18s} fwd {f*1e3:6.1f} bwd {b*1e3:6.1f} sum {t*1e3:6.1f}")
