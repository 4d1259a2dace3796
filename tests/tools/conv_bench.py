#!/usr/bin/env python3
"""Timing of the SURVEY 8(f) rows built on the general fused transform (HIP events, hipGraph-free eager
calls, median of --iters): fft_lm causal convolution (FixedSpectralBlock's hot line), PhaseAware /
ComplexRoPE full-spectrum filters, MultiScale bands, fnet.  Beside each: the reference's own op sequence
on the same GPU through torch.fft (rocFFT) -- a comparison point only, never part of the product.

Algorithmic bytes: a transform direction reads x and writes y once = 8 B/sample; fwd+bwd of a
learnable filter = 16 B/sample.  roofline = bytes / time / 8 TB/s."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd import _lib, functional as fn
from oracle import spectral_oracle as so          # tests/ may use the oracle; tools/ never does


def timeit(f, iters, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def graphed(step, iters, n=10):
    """ms per step with n steps replayed from one hipGraph (no host overhead between the launches)"""
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        step(); step()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            step()
    med, _ = timeit(gr.replay, iters)
    return med / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--conv", default="8x1024x512x128,64x1024x512x128,16x4096x512x128,64x512x256x64")
    ap.add_argument("--full", default="64x1024x256,16x2048x512,64x4096x256")
    ap.add_argument("--pair", default="8x1024x512x128,64x1024x512x128,16x4096x256x128",
                    help="B x T x D x kernel_len: transform pair and the spectrum-domain twin blocks")
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--opts", default="", help='library options, e.g. "fourstep=2;full8=0"')
    args = ap.parse_args()
    for o in filter(None, args.opts.split(";")):
        k_, v_ = o.split("=")
        _lib.set_option(k_, int(v_))
    dev = torch.device("cuda:0")
    out = []
    for sh in filter(None, args.conv.split(",")):
        B, T, C, K = map(int, sh.split("x"))
        x = torch.randn(B, T, C, device=dev, requires_grad=True)
        g = torch.randn(B, T, C, device=dev)
        kern = (0.1 * torch.randn(K, device=dev)).requires_grad_(True)
        gain = torch.ones(C, device=dev, requires_grad=True)
        logits = torch.full((so.next_pow2(T + K - 1) // 2 + 1,), 2.0, device=dev, requires_grad=True)
        gctx = torch.rand(B, C, device=dev)

        def ours():
            y = pkg.causal_spectral_conv(x, kern, gain, logits, gctx, None, 32)
            y.backward(g)
            x.grad = kern.grad = gain.grad = logits.grad = None

        def ours_fwd():
            with torch.no_grad():
                pkg.causal_spectral_conv(x, kern, gain, logits, gctx, None, 32)

        n_fft = so.next_pow2(T + K - 1)
        p = _lib.plan_ex(_lib.smx_shape(B, T, C, n_fft // 2 + 1, n_fft, n_fft // 2 + 1))
        med, mn = timeit(ours, args.iters)
        fmed, fmn = timeit(ours_fwd, args.iters)
        try:
            gmed = graphed(ours, args.iters)
        except Exception as e:                                      # noqa: BLE001
            gmed = float("nan")
        rec = {"op": "causal_spectral_conv fwd+bwd", "shape": sh, "n_fft": n_fft, "graph_ms": round(gmed, 4),
               "roofline_fwd_bwd_graph": round(16 * B * T * C / (gmed * 1e-3) / 8e12, 4),
               "plan": {"bands": p.bands, "groups": p.groups, "nsplit": p.nsplit},
               "ms": round(med, 4), "min_ms": round(mn, 4), "fwd_ms": round(fmed, 4),
               "roofline_fwd_bwd": round(16 * B * T * C / (med * 1e-3) / 8e12, 4),
               "roofline_fwd": round(8 * B * T * C / (fmed * 1e-3) / 8e12, 4)}
        if not args.no_torch:
            def ref():
                xx = x.detach().requires_grad_(True)
                y = so_gpu_conv(xx, kern, gain, logits, gctx)
                y.backward(g)
                kern.grad = gain.grad = logits.grad = None
            rmed, _ = timeit(ref, max(5, args.iters // 3))
            rec["torch_fft_ms"] = round(rmed, 4)
        out.append(rec); print(json.dumps(rec), flush=True)
    if args.full == "none":
        args.full = ""
    for sh in filter(None, args.full.split(",")):
        B, T, D = map(int, sh.split("x"))
        x = torch.randn(B, T, D, device=dev, requires_grad=True)
        g = torch.randn(B, T, D, device=dev)
        m = pkg.PhaseAwareSpectralMixing(D).to(dev)

        def ours():
            y = m(x); y.backward(g); x.grad = None; m.zero_grad(set_to_none=True)
        med, mn = timeit(ours, args.iters)
        p = _lib.plan_ex(_lib.smx_shape(B, T, D, T // 2 + 1, T, T // 2 + 1))
        try:                                          # the same step from a hipGraph: the GPU's share of the eager figure
            gmed = graphed(ours, args.iters)
        except Exception:                             # noqa: BLE001
            gmed = float("nan")
        rec = {"op": "PhaseAwareSpectralMixing fwd+bwd", "shape": sh,
               "plan": {"bands": p.bands, "groups": p.groups, "nsplit": p.nsplit}, "ms": round(med, 4),
               "roofline_fwd_bwd": round(16 * B * T * D / (med * 1e-3) / 8e12, 4), "graph_ms": round(gmed, 4),
               "roofline_fwd_bwd_graph": round(16 * B * T * D / (gmed * 1e-3) / 8e12, 4)}
        if not args.no_torch:
            def ref():
                xx = x.detach().requires_grad_(True)
                y = so.phase_aware_port(xx, m.magnitude_filter, m.phase_filter); y.backward(g)
                m.zero_grad(set_to_none=True)
            rec["torch_fft_ms"] = round(timeit(ref, max(5, args.iters // 3))[0], 4)
        out.append(rec); print(json.dumps(rec), flush=True)
        ms = pkg.MultiScaleSpectralFeatures(D).to(dev)
        with torch.no_grad():
            med, _ = timeit(lambda: ms.bands(x), args.iters)
        rec = {"op": "MultiScale bands fwd", "shape": sh, "ms": round(med, 4)}
        if not args.no_torch:
            with torch.no_grad():
                rec["torch_fft_ms"] = round(timeit(lambda: so.multiscale_bands_port(x), max(5, args.iters // 3))[0], 4)
        out.append(rec); print(json.dumps(rec), flush=True)
        z = torch.randn(B, T, D // 2, device=dev, dtype=torch.complex64)
        med, _ = timeit(lambda: fn.seq_fft_raw(z), args.iters)
        rec = {"op": "fnet_attention fwd", "shape": f"{B}x{T}x{D // 2} c64", "ms": round(med, 4)}
        if not args.no_torch:
            rec["torch_fft_ms"] = round(timeit(lambda: torch.fft.fft(z, dim=1), max(5, args.iters // 3))[0], 4)
        out.append(rec); print(json.dumps(rec), flush=True)
    bench_pair(args, dev, out)


def bench_pair(args, dev, out):
    """The transform pair (functional.rfft / irfft) and the two blocks built on it, fwd+bwd, beside the same op
    sequence through torch.fft (the oracle's restatement of the reference modules, run on the GPU)."""
    for sh in filter(None, args.pair.split(",")):
        B, T, D, K = map(int, sh.split("x"))
        n_fft = so.next_pow2(T + K - 1)
        x = torch.randn(B, T, D, device=dev, requires_grad=True)
        g = torch.randn(B, T, D, device=dev)

        def ours():
            y = fn.irfft(fn.rfft(x, n_fft), n_fft, T); y.backward(g); x.grad = None
        med, mn = timeit(ours, args.iters)
        # four transforms, each reads or writes the (B, T, D) tensor once and the (B, n_fft/2+1, D) spectrum once
        nbytes = 4 * (4 * B * T * D + 8 * B * (n_fft // 2 + 1) * D)
        rec = {"op": "rfft -> irfft fwd+bwd", "shape": f"{B}x{T}x{D} n_fft {n_fft}", "ms": round(med, 4),
               "min_ms": round(mn, 4), "achieved_TBps": round(nbytes / (med * 1e-3) / 1e12, 3)}
        if not args.no_torch:
            import torch.nn.functional as Fnn

            def ref():
                xx = x.detach().requires_grad_(True)
                y = torch.fft.irfft(torch.fft.rfft(Fnn.pad(xx, (0, 0, 0, n_fft - T)), dim=1), n=n_fft, dim=1)[:, :T]
                y.backward(g)
            rec["torch_fft_ms"] = round(timeit(ref, max(5, args.iters // 3))[0], 4)
        out.append(rec); print(json.dumps(rec), flush=True)
        for cls, port in ((pkg.FrequencyNativeBlock, so.freq_native_block_port),
                          (pkg.BicameralBlock, so.bicameral_block_port)):
            blk = cls(D, seq_len=T, kernel_len=K, transition_bins=32, dropout=0.0).to(dev)
            cutoff = (n_fft // 2 + 1) // 2

            def step():
                y = blk(x, cutoff=cutoff); y.backward(g); x.grad = None; blk.zero_grad(set_to_none=True)
            med, mn = timeit(step, args.iters)
            rec = {"op": cls.__name__ + " fwd+bwd", "shape": f"{B}x{T}x{D} kernel {K}", "ms": round(med, 4),
                   "min_ms": round(mn, 4)}
            if not args.no_torch:
                sd = dict(blk.named_parameters())

                def ref():
                    xx = x.detach().requires_grad_(True)
                    y = port(sd, xx, cutoff, 32); y.backward(g); blk.zero_grad(set_to_none=True)
                rec["torch_fft_ms"] = round(timeit(ref, max(5, args.iters // 3))[0], 4)
            out.append(rec); print(json.dumps(rec), flush=True)


def so_gpu_conv(x, kernel, gain, logits, g_ctx):
    """reference train_fixed_full.py:507-555 with torch.fft on the GPU (cutoff None)."""
    import torch.nn.functional as Fn
    B, T, C = x.shape
    K = kernel.shape[0]
    n_fft = so.next_pow2(T + K - 1)
    k = torch.zeros(n_fft, device=x.device, dtype=x.dtype)
    k[:K] = kernel
    k_freq = torch.fft.rfft(k)
    x_freq = torch.fft.rfft(Fn.pad(x, (0, 0, 0, n_fft - T)), dim=1)
    y_freq = x_freq * k_freq.unsqueeze(0).unsqueeze(-1) * gain.unsqueeze(0).unsqueeze(0)
    y_freq = y_freq * torch.sigmoid(logits[:y_freq.size(1)]).unsqueeze(0).unsqueeze(-1) * g_ctx.unsqueeze(1)
    return torch.fft.irfft(y_freq, n=n_fft, dim=1)[:, :T, :]


if __name__ == "__main__":
    main()
