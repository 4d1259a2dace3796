import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib, functional as fn
dev = torch.device("cuda:0")
B, N, D, F = 64, 4096, 256, 128
x = torch.randn(B, N, D, device=dev); wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
def t(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); ev=[]
    for _ in range(it):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record(); f(); b.record(); ev.append((a,b))
    torch.cuda.synchronize(); ts=sorted(p.elapsed_time(q) for p,q in ev); return ts[len(ts)//2]*1e3, ts[0]*1e3
s = t(lambda: pkg.pruned_rfft(x, F)); f = t(lambda: fn.forward_raw(x, wr, wi, bias, save_spectrum=True))
print(os.path.basename(os.environ.get("SMX_LIB","libsmx.so")), "spec %.1f/%.1f  fwd %.1f/%.1f us" % (s+f))
