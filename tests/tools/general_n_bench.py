import os, sys, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib, functional as fn
import numpy as np
from oracle import spectral_oracle as so
dev = torch.device("cuda:0")
def t(f, it=5):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
import os
if os.environ.get("TILED"):
    _lib.set_option("tiled_dft", int(os.environ["TILED"]))
for (B, N, D, F) in [(64, 4000, 256, 128), (64, 1000, 256, 128), (8, 128, 256, 128), (16, 4000, 255, 128)]:
    x = torch.randn(B, N, D, device=dev); g = torch.randn(B, N, D, device=dev)
    wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); b = torch.randn(D, device=dev)
    y, xk = fn.forward_raw(x, wr, wi, b, save_spectrum=True)
    f_ms = t(lambda: fn.forward_raw(x, wr, wi, b, save_spectrum=True))
    b_ms = t(lambda: fn.backward_raw(g, xk, wr, wi))
    # parity on a slab
    sl = slice(0, 2)
    yr, _ = so.forward_closed(x[sl].cpu().numpy(), wr.cpu().numpy(), wi.cpu().numpy(), b.cpu().numpy())
    err = float(np.abs(y[sl].cpu().numpy() - yr).max() / np.abs(yr).max())
    print(json.dumps({"shape": [B, N, D, F], "path": _lib.plan(B, N, D, F).path, "fwd_ms": round(f_ms, 3), "bwd_ms": round(b_ms, 3), "err_y": err}), flush=True)
