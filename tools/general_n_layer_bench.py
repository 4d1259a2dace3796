import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib
dev = torch.device("cuda:0")
def t(f, it=10):
    f(); f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
import warnings
warnings.simplefilter("ignore")
for (B, N, D) in [(64, 1000, 256), (64, 1024, 256), (64, 3000, 256), (64, 3072, 256), (64, 4000, 256), (64, 4096, 256), (64, 4000, 512), (64, 4096, 512), (64, 4000, 255)]:
    for dec in (1, 0):
        if dec == 0 and N % 256 == 0: continue
        with _lib.options(decim16=dec):
            layer = pkg.SpectralMixingLayer(D).to(dev)
            x = torch.randn(B, N, D, device=dev, requires_grad=True); g = torch.randn(B, N, D, device=dev)
            def step():
                y = layer(x); y.backward(g); x.grad = None; layer.zero_grad(set_to_none=True)
            ms = t(step)
        print(json.dumps({"shape": [B, N, D], "decim16": dec, "fwd_bwd_ms": round(ms, 3), "GSamples_s": round(B*N*D/ms/1e6, 1)}), flush=True)
