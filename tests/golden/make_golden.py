#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Each case stores inputs, parameters, upstream gradient and the reference's fp32 outputs/gradients
(plus an fp64 evaluation of the same reference module for error budgeting where that is small).
Case list = SURVEY.md section 8(c), G1..G13.
"""
import os
import sys
import warnings
import io
import contextlib

import numpy as np
import torch

REF = os.environ.get("SMX_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
with contextlib.redirect_stdout(io.StringIO()):      # the package prints a banner on import
    from fft_tensor.spectral_layers import SpectralMixingLayer
    from fft_tensor.wirtinger_ops import WirtingerSpectralFilter, WirtingerGradient

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 1234


def layer_case(name, B, N, D, num_filters=None, init="random", learnable=True, g_kind="randn",
               store64=True):
    torch.manual_seed(SEED)
    layer = SpectralMixingLayer(D, num_filters=num_filters, learnable=learnable)
    if learnable and init == "random":
        with torch.no_grad():
            layer.weight_real.copy_(1.0 + 0.5 * torch.randn_like(layer.weight_real))
            layer.weight_imag.copy_(0.5 * torch.randn_like(layer.weight_imag))
            layer.bias.copy_(0.1 * torch.randn_like(layer.bias))
    x = torch.randn(B, N, D)
    g = torch.randn(B, N, D) if g_kind == "randn" else torch.ones(B, N, D)

    def run(dtype):
        l = SpectralMixingLayer(D, num_filters=num_filters, learnable=learnable).to(dtype)
        if learnable:
            l.load_state_dict({k: v.to(dtype) for k, v in layer.state_dict().items()})
        xx = x.to(dtype).clone().requires_grad_(True)
        y = l(xx)
        y.backward(g.to(dtype))
        out = {"y": y.detach(), "grad_x": xx.grad}
        if learnable:
            out.update(grad_w_real=l.weight_real.grad, grad_w_imag=l.weight_imag.grad,
                       grad_bias=l.bias.grad)
        return out

    o32 = run(torch.float32)
    rec = {"x": x, "g": g, "num_filters": np.int64(layer.num_filters),
           "learnable": np.int64(learnable)}
    if learnable:
        rec.update(weight_real=layer.weight_real.detach(), weight_imag=layer.weight_imag.detach(),
                   bias=layer.bias.detach())
    rec.update(o32)
    if store64 and B * N * D <= 40000:
        rec.update({k + "_f64": v for k, v in run(torch.float64).items()})
    save(name, rec)


def wirtinger_case(name, B, N, D, F):
    torch.manual_seed(SEED)
    filt = WirtingerSpectralFilter(D, F)
    with torch.no_grad():
        filt.weight.real.copy_(1.0 + 0.5 * torch.randn(D, F))
        filt.weight.imag.copy_(0.5 * torch.randn(D, F))
    xf = torch.complex(torch.randn(B, N, D), torch.randn(B, N, D)).requires_grad_(True)
    gf = torch.complex(torch.randn(B, N, D), torch.randn(B, N, D))
    out = filt(xf)
    out.backward(gf)
    # raw WirtingerGradient.apply on (B,k,D) x (1,k,D)
    k = min(F, N // 2)
    xs = xf.detach()[:, :k, :].clone().requires_grad_(True)
    ws = filt.weight().detach()[:, :k].T.unsqueeze(0).clone().requires_grad_(True)
    o2 = WirtingerGradient.apply(xs, ws)
    o2.backward(gf[:, :k, :])
    save(name, {
        "x_freq": xf.detach(), "g_freq": gf, "w_real": filt.weight.real.detach(),
        "w_imag": filt.weight.imag.detach(), "out": out.detach(), "grad_x_freq": xf.grad,
        "grad_w_real": filt.weight.real.grad, "grad_w_imag": filt.weight.imag.grad,
        "mul_out": o2.detach(), "mul_grad_x": xs.grad, "mul_grad_w": ws.grad,
    })


def wfused_case(name, B, N, D, F):
    """BASELINE config 5's unit: ifft(WirtingerSpectralFilter(fft(x))).real forward + backward
    (reference wirtinger_ops.py:170-203 wrapped in torch.fft, as ARCHITECTURE.md:74-86 does)."""
    torch.manual_seed(SEED)
    filt = WirtingerSpectralFilter(D, F)
    with torch.no_grad():
        filt.weight.real.copy_(1.0 + 0.5 * torch.randn(D, F))
        filt.weight.imag.copy_(0.5 * torch.randn(D, F))
    x = torch.randn(B, N, D, requires_grad=True)
    g = torch.randn(B, N, D)
    y = torch.fft.ifft(filt(torch.fft.fft(x, dim=1)), dim=1).real
    y.backward(g)
    save(name, {"x": x.detach(), "g": g, "w_real": filt.weight.real.detach(),
                "w_imag": filt.weight.imag.detach(), "y": y.detach(), "grad_x": x.grad,
                "grad_w_real": filt.weight.real.grad, "grad_w_imag": filt.weight.imag.grad})


def _randomize(mod, scale=0.3):
    with torch.no_grad():
        for p in mod.parameters():
            if p.is_complex():
                p.add_(scale * torch.complex(torch.randn(p.shape), torch.randn(p.shape)))
            else:
                p.add_(scale * torch.randn_like(p))


def _module_case(name, mod, x, g, fwd=None, extra=None):
    """Forward + backward of a reference module; stores its state_dict, output and every gradient."""
    x = x.clone().requires_grad_(True)
    y = fwd(mod, x) if fwd else mod(x)
    y.backward(g)
    rec = {"x": x.detach(), "g": g, "y": y.detach(), "grad_x": x.grad}
    for k, v in mod.state_dict().items():
        rec["sd." + k] = v
    for k, p in mod.named_parameters():
        rec["grad." + k] = p.grad
    rec.update(extra or {})
    save(name, rec)


def fixed_block_case(name, B, T, C, seq_len, kernel_len, trans, cutoff):
    """fft_lm.train_fixed_full.FixedSpectralBlock (reference fft_lm/train_fixed_full.py:427-563), dropout 0."""
    with contextlib.redirect_stdout(io.StringIO()):
        from fft_lm.train_fixed_full import FixedSpectralBlock
    torch.manual_seed(SEED)
    blk = FixedSpectralBlock(C, seq_len=seq_len, kernel_len=kernel_len, transition_bins=trans, dropout=0.0)
    _randomize(blk)
    x, g = torch.randn(B, T, C), torch.randn(B, T, C)
    _module_case(name, blk, x, g, fwd=lambda m, xx: m(xx, cutoff=cutoff),
                 extra={"cutoff": np.int64(-1 if cutoff is None else cutoff), "seq_len": np.int64(seq_len),
                        "kernel_len": np.int64(kernel_len), "transition_bins": np.int64(trans)})


def _twin_block_case(name, cls_path, B, T, C, seq_len, kernel_len, trans, cutoff):
    """The twins of FixedSpectralBlock that work on the spectrum between the two transforms:
    fft_lm.frequency_native.FrequencyNativeBlock (reference fft_lm/frequency_native.py:242-362) and
    fft_lm.bicameral.BicameralBlock (reference fft_lm/bicameral.py:26-278), dropout 0."""
    import importlib
    with contextlib.redirect_stdout(io.StringIO()):
        mod_name, cls_name = cls_path.rsplit(".", 1)
        cls = getattr(importlib.import_module(mod_name), cls_name)
    torch.manual_seed(SEED)
    blk = cls(C, seq_len=seq_len, kernel_len=kernel_len, transition_bins=trans, dropout=0.0)
    _randomize(blk)
    x, g = torch.randn(B, T, C), torch.randn(B, T, C)
    _module_case(name, blk, x, g, fwd=lambda m, xx: m(xx, cutoff=cutoff),
                 extra={"cutoff": np.int64(-1 if cutoff is None else cutoff), "seq_len": np.int64(seq_len),
                        "kernel_len": np.int64(kernel_len), "transition_bins": np.int64(trans)})


def freq_native_case(name, *a):
    _twin_block_case(name, "fft_lm.frequency_native.FrequencyNativeBlock", *a)


def bicameral_case(name, *a):
    _twin_block_case(name, "fft_lm.bicameral.BicameralBlock", *a)


def phase_aware_case(name, B, T, D):
    from fft_tensor.spectral_enhancements import PhaseAwareSpectralMixing
    torch.manual_seed(SEED)
    m = PhaseAwareSpectralMixing(D)
    _randomize(m, 0.5)
    _module_case(name, m, torch.randn(B, T, D), torch.randn(B, T, D))


def multiscale_case(name, B, T, D):
    from fft_tensor.spectral_enhancements import MultiScaleSpectralFeatures
    torch.manual_seed(SEED)
    m = MultiScaleSpectralFeatures(D)
    _module_case(name, m, torch.randn(B, T, D), torch.randn(B, T, D))


def rope_layer_case(name, B, T, D):
    from fft_tensor.complex_rope import ComplexRoPESpectralLayer
    torch.manual_seed(SEED)
    m = ComplexRoPESpectralLayer(D, dropout=0.0)
    _randomize(m)
    _module_case(name, m, torch.randn(B, T, D), torch.randn(B, T, D))


def fnet_case(name, B, N, D):
    from fft_tensor.frequency_ops import FrequencyAttention
    torch.manual_seed(SEED)
    z = torch.complex(torch.randn(B, N, D), torch.randn(B, N, D)).requires_grad_(True)
    gz = torch.complex(torch.randn(B, N, D), torch.randn(B, N, D))
    out = FrequencyAttention.fnet_attention(z)
    out.backward(gz)
    save(name, {"z": z.detach(), "gz": gz, "out": out.detach(), "grad_z": z.grad})


def freqconv_case(name, B, Fb, C):
    with contextlib.redirect_stdout(io.StringIO()):
        from fft_lm.frequency_native import FrequencyConvFunc
    torch.manual_seed(SEED)
    x = torch.complex(torch.randn(B, Fb, C), torch.randn(B, Fb, C)).requires_grad_(True)
    kf = torch.complex(torch.randn(Fb), torch.randn(Fb)).requires_grad_(True)
    gain = (1 + 0.3 * torch.randn(C)).requires_grad_(True)
    go = torch.complex(torch.randn(B, Fb, C), torch.randn(B, Fb, C))
    out = FrequencyConvFunc.apply(x, kf, gain)
    out.backward(go)
    save(name, {"x_freq": x.detach(), "kernel_freq": kf.detach(), "gain": gain.detach(), "g": go,
                "out": out.detach(), "grad_x": x.grad, "grad_kernel": kf.grad, "grad_gain": gain.grad})


def save(name, rec):
    arrs = {}
    for k, v in rec.items():
        arrs[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def block_case(name, B, N, D):
    """SpectralMLPBlock (reference spectral_layers.py:135-190), dropout 0: the immediate caller of the
    hot path.  Stores the reference state_dict so the drop-in block can load it unchanged."""
    from fft_tensor.spectral_layers import SpectralMLPBlock
    torch.manual_seed(SEED)
    blk = SpectralMLPBlock(D, mlp_ratio=2, dropout=0.0)
    with torch.no_grad():
        blk.spectral_mix.weight_real.copy_(1.0 + 0.5 * torch.randn_like(blk.spectral_mix.weight_real))
        blk.spectral_mix.weight_imag.copy_(0.5 * torch.randn_like(blk.spectral_mix.weight_imag))
        blk.spectral_mix.bias.copy_(0.1 * torch.randn_like(blk.spectral_mix.bias))
        blk.norm1.weight.copy_(1.0 + 0.1 * torch.randn(D)); blk.norm1.bias.copy_(0.1 * torch.randn(D))
    x = torch.randn(B, N, D, requires_grad=True)
    g = torch.randn(B, N, D)
    y = blk(x)
    y.backward(g)
    rec = {"x": x.detach(), "g": g, "y": y.detach(), "grad_x": x.grad}
    for k, v in blk.state_dict().items():
        rec["sd." + k] = v
    for k, p in blk.named_parameters():
        rec["grad." + k] = p.grad
    save(name, rec)


def hybrid_case(name, B, N, D, heads):
    """HybridSpectralAttention (reference spectral_layers.py:193-256), dropout 0: third public class of the
    module this package mirrors.  Stores the reference state_dict, output and all gradients."""
    from fft_tensor.spectral_layers import HybridSpectralAttention
    torch.manual_seed(SEED)
    m = HybridSpectralAttention(D, num_heads=heads, dropout=0.0)
    with torch.no_grad():
        m.spectral.weight_real.copy_(1.0 + 0.5 * torch.randn_like(m.spectral.weight_real))
        m.spectral.weight_imag.copy_(0.5 * torch.randn_like(m.spectral.weight_imag))
        m.spectral.bias.copy_(0.1 * torch.randn_like(m.spectral.bias))
    x = torch.randn(B, N, D, requires_grad=True)
    g = torch.randn(B, N, D)
    y = m(x)
    y.backward(g)
    rec = {"x": x.detach(), "g": g, "y": y.detach(), "grad_x": x.grad, "heads": np.int64(heads)}
    for k, v in m.state_dict().items():
        rec["sd." + k] = v
    for k, p in m.named_parameters():
        rec["grad." + k] = p.grad
    save(name, rec)


def mixhalf_case(name, B, N, D, num_filters=None, offset=0.0):
    """First residual line of SpectralMLPBlock.forward (reference spectral_layers.py:185):
    y = x + spectral_mix(norm1(x)), built from the reference block's own members, dropout 0."""
    from fft_tensor.spectral_layers import SpectralMLPBlock
    torch.manual_seed(SEED)
    blk = SpectralMLPBlock(D, mlp_ratio=1, dropout=0.0)
    if num_filters is not None:
        blk.spectral_mix = SpectralMixingLayer(D, num_filters=num_filters)
    sm, n1 = blk.spectral_mix, blk.norm1
    with torch.no_grad():
        sm.weight_real.copy_(1.0 + 0.5 * torch.randn_like(sm.weight_real))
        sm.weight_imag.copy_(0.5 * torch.randn_like(sm.weight_imag))
        sm.bias.copy_(0.1 * torch.randn_like(sm.bias))
        n1.weight.copy_(1.0 + 0.3 * torch.randn(D)); n1.bias.copy_(0.2 * torch.randn(D))
    x = (offset + (1.0 + torch.rand(B, N, 1)) * torch.randn(B, N, D)).requires_grad_(True)
    g = torch.randn(B, N, D)
    y = x + sm(n1(x))
    y.backward(g)
    save(name, {"x": x.detach(), "g": g, "y": y.detach(), "grad_x": x.grad,
                "num_filters": np.int64(sm.num_filters), "eps": np.float64(n1.eps),
                "ln_weight": n1.weight.detach(), "ln_bias": n1.bias.detach(),
                "weight_real": sm.weight_real.detach(), "weight_imag": sm.weight_imag.detach(),
                "bias": sm.bias.detach(), "grad_ln_weight": n1.weight.grad,
                "grad_ln_bias": n1.bias.grad, "grad_w_real": sm.weight_real.grad,
                "grad_w_imag": sm.weight_imag.grad, "grad_bias": sm.bias.grad})


if __name__ == "__main__":
    only = sys.argv[1:]                     # optional: name prefixes to (re)generate
    if only:
        _save = save
        def save(name, rec, _s=_save):      # noqa: E306
            if any(name.startswith(p) for p in only):
                _s(name, rec)
    layer_case("G01_default_2x128x256", 2, 128, 256, init="default")           # k=64 < F
    layer_case("G02_c1class_1x512x256", 1, 512, 256)                            # k=128
    layer_case("G03_small_3x64x32", 3, 64, 32, num_filters=16)
    layer_case("G04_nonpow2_2x20x16", 2, 20, 16)
    layer_case("G05_oddN_2x21x8", 2, 21, 8)
    layer_case("G06_dconly_1x2x4", 1, 2, 4)                                     # k=1
    layer_case("G07_k0_1x1x4", 1, 1, 4)                                         # k=0 -> y=bias
    layer_case("G08_c2bins_1x4096x8", 1, 4096, 8, num_filters=128)
    layer_case("G09_c3bins_1x65536x2", 1, 65536, 2, num_filters=128, store64=False)
    layer_case("G10_fgtn2_2x256x16", 2, 256, 16, num_filters=200)               # F > N//2
    layer_case("G11_nolearn_2x128x32", 2, 128, 32, learnable=False)
    wirtinger_case("G12_wirtinger_2x32x16", 2, 32, 16, 8)
    layer_case("G13_ysum_2x128x256", 2, 128, 256, init="default", g_kind="ones")
    # extra shapes that exercise the decimated kernel's tails
    layer_case("G14_L3_2x768x12", 2, 768, 12, num_filters=100)                  # N=256*3, ragged D
    layer_case("G15_k256_1x1024x6", 1, 1024, 6, num_filters=256)                # two bands
    layer_case("G16_k200_2x512x34", 2, 512, 34, num_filters=200)                # two bands, ragged
    layer_case("G17_k400_1x1024x6", 1, 1024, 6, num_filters=400)                # four bands
    layer_case("G18_k512_2x2048x4", 2, 2048, 4, num_filters=512)                # four bands, k = 512
    # more than 512 kept bins: band groups of the four-band kernels + edge bins (multiples of 512)
    layer_case("G19_k700_1x2048x4", 1, 2048, 4, num_filters=700, store64=False)         # two groups
    layer_case("G20_k1500_2x4096x2", 2, 4096, 2, num_filters=1500, store64=False)       # three groups
    layer_case("G21_kfull_1x2048x6", 1, 2048, 6, num_filters=1024, store64=False)       # k = N/2: every bin
    # lengths that are not multiples of 256, large enough for the tiled (matrix-core) direct kernels
    layer_case("G22_n1000_2x1000x64", 2, 1000, 64, num_filters=128, store64=False)
    layer_case("G23_n4000_2x4000x32", 2, 4000, 32, num_filters=64, store64=False)
    # C5's unit through the Wirtinger filter API, one and two bands
    wfused_case("W01_wfused_2x512x64", 2, 512, 64, 48)
    wfused_case("W02_wfused_2x1024x12", 2, 1024, 12, 200)
    # ---- the causal FFT convolution block the reference trains with -------------------
    fixed_block_case("F01_fixed_2x192x32", 2, 192, 32, 192, 64, 8, None)       # n_fft 256: one band + Nyquist, padded rows
    fixed_block_case("F02_fixed_2x512x16", 2, 512, 16, 512, 128, 32, 200)      # n_fft 1024: four bands + Nyquist, cutoff
    fixed_block_case("F03_fixed_1x1024x8", 1, 1024, 8, 1024, 128, 32, 128)     # n_fft 2048: the reference's default lengths
    fixed_block_case("F04_fixed_2x100x16", 2, 100, 16, 100, 16, 4, 40)         # n_fft 128: direct plan
    fixed_block_case("F05_fixed_2x100x16", 2, 100, 16, 256, 32, 4, None)       # T shorter than seq_len
    fixed_block_case("F06_fixed_2x300x9", 2, 300, 9, 300, 20, 4, 500)          # odd channel count, cutoff beyond the bins
    fixed_block_case("F07_fixed_1x8000x4", 1, 8000, 4, 8000, 128, 32, 3000)     # n_fft 8192: two-level rank-one kernels, native kernel response
    freqconv_case("FC1_freqconv_2x33x8", 2, 33, 8)
    freq_native_case("T01_freqnative_2x192x16", 2, 192, 16, 192, 64, 8, None)   # n_fft 256: one band + Nyquist
    freq_native_case("T02_freqnative_1x1024x8", 1, 1024, 8, 1024, 128, 32, 300) # n_fft 2048: four-step, cutoff
    freq_native_case("T03_freqnative_2x100x6", 2, 100, 6, 128, 16, 4, 40)       # n_fft 128: direct plan, T < seq_len
    bicameral_case("T11_bicameral_2x192x16", 2, 192, 16, 192, 64, 8, None)
    bicameral_case("T12_bicameral_1x1024x8", 1, 1024, 8, 1024, 128, 32, 300)
    bicameral_case("T13_bicameral_2x300x10", 2, 300, 10, 300, 20, 4, 100)       # n_fft 512: two bands + Nyquist
    # ---- sequence mixers on the full one-sided spectrum ------------------------------------
    phase_aware_case("P01_phase_2x512x32", 2, 512, 32)
    phase_aware_case("P02_phase_2x33x6", 2, 33, 6)
    phase_aware_case("P03_phase_1x2048x4", 1, 2048, 4)
    multiscale_case("M01_multi_2x1024x16", 2, 1024, 16)
    multiscale_case("M02_multi_2x50x8", 2, 50, 8)
    rope_layer_case("R01_rope_2x256x16", 2, 256, 16)
    rope_layer_case("R02_rope_2x40x8", 2, 40, 8)
    rope_layer_case("R03_rope_1x768x6", 1, 768, 6)
    # ---- complex-input sequence FFT -------------------------------------------------------
    fnet_case("N01_fnet_2x256x8", 2, 256, 8)
    fnet_case("N02_fnet_2x30x5", 2, 30, 5)
    fnet_case("N03_fnet_1x1024x3", 1, 1024, 3)
    fnet_case("N04_fnet_2x2048x5", 2, 2048, 5)                                 # four-step plan: packed spectrum out
    block_case("B01_mlpblock_2x512x64", 2, 512, 64)
    hybrid_case("A01_hybrid_2x256x64", 2, 256, 64, heads=4)
    # first half of the block (LayerNorm + mix + residual), one case per transform plan / row kernel
    mixhalf_case("H01_half_2x512x64", 2, 512, 64, offset=3.0)                    # decimated, one band
    mixhalf_case("H02_half_1x4096x8", 1, 4096, 8, num_filters=128)              # split plan
    mixhalf_case("H03_half_2x20x16", 2, 20, 16)                                 # direct plan
    mixhalf_case("H04_half_2x21x9", 2, 21, 9, num_filters=4)                    # odd D: scalar rows
    mixhalf_case("H05_half_1x1024x12", 1, 1024, 12, num_filters=400)            # four bands
    mixhalf_case("H06_half_2x512x34", 2, 512, 34, num_filters=200, offset=-1.0) # two bands, ragged D
