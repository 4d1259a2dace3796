#!/usr/bin/env python3
"""Does the allocation that holds the WRITTEN tensors decide the step-time mode?
Child processes (fresh allocator each): four separate 256 MiB allocations vs one 1 GiB slab carved in four,
with / without a dummy allocated first, with / without expandable segments.  Times the fwd + bwd launch pair."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode):
    sys.path.insert(0, ROOT)
    import torch
    from tensor_cuda_fft_amd import _lib
    dev = torch.device("cuda:0")
    B, N, D, F = 64, 4096, 256, 128
    n = B * N * D
    if "dummy" in mode:
        dummy = torch.empty(n, device=dev)
    if "slab" in mode:
        slab = torch.empty(4 * n, device=dev)
        ts = [slab[i * n:(i + 1) * n].view(B, N, D) for i in range(4)]
    else:
        ts = [torch.empty(B, N, D, device=dev) for _ in range(4)]
    for t in ts:
        t.normal_()
    wr = torch.randn(D, F, device=dev); wi = torch.randn(D, F, device=dev); bias = torch.randn(D, device=dev)
    xk = torch.empty(B, F, D, dtype=torch.complex64, device=dev)
    flat = torch.empty(2 * D * F + D, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, N, D, F), dtype=torch.uint8, device=dev)
    lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
    lib.smx_prepare(N)

    def fwd(x, y):
        lib.smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), bias.data_ptr(), y.data_ptr(), xk.data_ptr(),
                        ws.data_ptr(), ws.numel(), B, N, D, F, 0, st)

    def bwd(g, gx):
        lib.smx_backward(g.data_ptr(), xk.data_ptr(), wr.data_ptr(), wi.data_ptr(), gx.data_ptr(), flat.data_ptr(),
                         flat[D * F:].data_ptr(), flat[2 * D * F:].data_ptr(), ws.data_ptr(), ws.numel(), B, N, D, F,
                         7, st)

    def timed(x, y, g, gx, iters=150):
        for _ in range(600):
            fwd(x, y); bwd(g, gx)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fwd(x, y); bwd(g, gx)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    out = []
    for perm in [(0, 1, 2, 3), (1, 0, 3, 2), (2, 3, 0, 1), (0, 2, 1, 3)]:
        x, y, g, gx = (ts[i] for i in perm)
        out.append(f"x{perm[0]}y{perm[1]}g{perm[2]}gx{perm[3]}={timed(x, y, g, gx):.1f}")
    print(f"[{mode}] VA(MiB) {[hex(t.data_ptr() >> 20) for t in ts]} us/step: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1]); sys.exit(0)
    for mode, env in [("separate", {}), ("separate_dummy", {}), ("slab", {}), ("slab_dummy", {}),
                      ("separate_expandable", {"PYTORCH_HIP_ALLOC_CONF": "expandable_segments:True"}),
                      ("separate", {}), ("slab", {})]:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        print((p.stdout.strip() or p.stderr.strip()[-300:]), flush=True)
