#!/usr/bin/env python3
"""What crashed `hipStreamEndCapture` in round 1 (gpurun_out/t_test_hipgraph_replay_matches_eager.log,
`Fatal Python error: Segmentation fault` inside torch/cuda/graphs.py capture_end)?

Each case runs in its own child process (a crash must not take the others down; this parent never touches
the GPU) and prints its exit status.  Cases:
  linear_keep   plain nn.Linear, no libsmx: a warm-up activation produced on a SIDE stream is still alive
                when torch.cuda.graph() captures the same module on its own capture stream
  linear_drop   the same with the warm-up activation released before the capture
  smx_keep      the same lifetime pattern with SpectralMixingLayer forward + backward (libsmx)
  smx_drop      ... and released before the capture (what tests/test_parity_gpu.py does)
  smx_oldcache_* round 1's workspace cache put back (keyed by stream, caching whatever it allocated -- also a
                buffer allocated INSIDE the capture, i.e. in the graph's private pool), keep / drop as above
  smx_grow      workspace first sized by a SMALL shape eagerly, then a larger shape captured (the cached
                workspace is too small inside the capture: functional._workspace must not cache a buffer that
                lives in the graph's private pool)
"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["linear_keep", "linear_drop", "smx_keep", "smx_drop", "smx_grow", "smx_oldcache_keep",
         "smx_oldcache_drop"]


def child(case):
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    import torch
    dev = torch.device("cuda:0")
    if case.startswith("linear"):
        mod = torch.nn.Linear(64, 64).to(dev)
        x = torch.randn(4, 2048, 64, device=dev, requires_grad=True)

        def step():
            y = mod(x)
            y.sum().backward()
            return y
    else:
        import tensor_cuda_fft_amd as pkg
        if "oldcache" in case:
            from tensor_cuda_fft_amd import functional
            cache = {}

            def old_workspace(d, nbytes):
                if nbytes == 0:
                    return None
                key = (d.index, torch.cuda.current_stream(d).cuda_stream)
                ws = cache.get(key)
                if ws is None or ws.numel() < nbytes:
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
                    cache[key] = ws
                return ws
            functional._workspace = old_workspace
        mod = pkg.SpectralMixingLayer(64).to(dev)
        x = torch.randn(4, 2048, 64, device=dev, requires_grad=True)
        g = torch.randn(4, 2048, 64, device=dev)
        if case == "smx_grow":
            xs = torch.randn(1, 256, 64, device=dev, requires_grad=True)
            mod(xs).sum().backward()               # small shape first: small cached workspace

        def step():
            y = mod(x)
            y.backward(g)
            return y
    if case != "smx_grow":
        step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        keep = step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    if case.endswith("drop") or case == "smx_grow":
        del keep
        x.grad = None
        mod.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    graph.replay()
    torch.cuda.synchronize()
    if case == "smx_grow":                      # eager call after the capture must still be correct
        ref = out.clone()
        y2 = step()
        torch.cuda.synchronize()
        assert torch.equal(y2, ref)
    print(f"{case}: capture + replay OK", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for c in CASES:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True,
                               timeout=180)
            tail = (p.stdout + p.stderr).strip().splitlines()
            keyl = [l for l in tail if "Fatal" in l or "capture_end" in l or "OK" in l or "Error" in l][:4]
            print(f"[{c}] exit={p.returncode} :: " + " | ".join(keyl), flush=True)
        except subprocess.TimeoutExpired:
            print(f"[{c}] timeout", flush=True)
