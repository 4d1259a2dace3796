#!/usr/bin/env python3
"""One arena (fixed physical memory), tensors carved at random 2 MiB-aligned offsets: is the
fast/slow step-time mode a function of the addresses?"""
import os, sys, time, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_cuda_fft_amd import _lib
dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 256, 128
n = B * N * D
GB = 1 << 30
arena = torch.empty(6 * GB, dtype=torch.uint8, device=dev)
wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
xk = torch.empty(B, F, D, dtype=torch.complex64, device=dev)
flat = torch.empty(2 * D * F + D, device=dev)
ws = torch.empty(_lib.workspace_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
base = arena.data_ptr()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def timed(px, py, pg, pgx, which=3, iters=150):
    def step():
        if which & 1: lib.smx_forward(px, wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), py, xk.data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)
        if which & 2: lib.smx_backward(pg, xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), pgx, flat.data_ptr(), flat[D*F:].data_ptr(), flat[2*D*F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F, 3, st)
    for _ in range(150): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
arena.zero_()
slots = list(range(0, 6 * GB - (n * 4), 2 << 20))
for trial in range(14):
    # 4 non-overlapping 256 MiB windows at random 2 MiB-aligned offsets
    while True:
        offs = sorted(random.sample(range(0, (6 * GB - n * 4) >> 21), 4))
        if all(b - a >= (n * 4) >> 21 for a, b in zip(offs, offs[1:])): break
    random.shuffle(offs)
    ps = [base + (o << 21) for o in offs]
    t = timed(*ps)
    tf = timed(*ps, which=1); tb = timed(*ps, which=2)
    print(f"trial {trial}: offs(MiB) x {offs[0]*2} y {offs[1]*2} g {offs[2]*2} gx {offs[3]*2}: step {t:.1f} us (fwd {tf:.1f} bwd {tb:.1f}) frac {16*n/(t*1e-6)/8e12:.3f}", flush=True)
