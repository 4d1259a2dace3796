"""CPU oracle for the spectral-mixing hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product (``tensor-cuda-fft-_amd/``) never does: it has no CPU path
and raises when the HIP library is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the reference
(`/root/reference/fft_tensor/spectral_layers.py`, `wirtinger_ops.py`) in the build container
with ``tests/golden/make_golden.py``; ``tests/test_oracle.py`` checks every function below
against them.

Two restatements are kept side by side:

* ``*_port`` -- the reference's *operation sequence* restated on CPU torch in fp32
  (full C2C fft -> slice * W -> zero fill -> ifft -> .real -> + bias, autograd backward).
  This is what ``bench.py`` times as ``cpu_baseline.kind == "port"``.
* ``*_closed`` -- the closed forms of SURVEY.md section 0.3/0.4 in numpy float64
  (pruned rfft / irfft), used for error budgeting and for sizes where the port is slow.

Reference lines restated (relative to /root/reference/):
  fft_tensor/spectral_layers.py:83-118   forward
  fft_tensor/spectral_layers.py:122-132  verify_energy_preservation
  fft_tensor/wirtinger_ops.py:45-50,67-82   complex multiply fwd / Wirtinger bwd
  fft_tensor/wirtinger_ops.py:170-203    WirtingerSpectralFilter.forward
"""
from __future__ import annotations

import math

import numpy as np

try:  # torch is only needed by the *_port functions
    import torch
except Exception:  # pragma: no cover
    torch = None


def num_bins(N: int, F: int) -> int:
    """k = min(num_filters, T // 2)   -- spectral_layers.py:94 / wirtinger_ops.py:187."""
    return min(int(F), int(N) // 2)


# --------------------------------------------------------------------------------------
# fp32 torch port of the reference op sequence
# --------------------------------------------------------------------------------------
def forward_port(x, w_re, w_im, bias):
    """spectral_layers.py:88-116 with dropout p=0.  x (B,N,D) f32; w_* (D,F); bias (D) or None."""
    B, N, D = x.shape
    spec = torch.fft.fft(x, dim=1)                                  # :88
    if w_re is not None:
        k = num_bins(N, w_re.shape[1])                              # :94
        w = torch.complex(w_re, w_im)                               # :97
        out = torch.zeros_like(spec)                                # :104
        out[:, :k, :] = spec[:, :k, :] * w[:, :k].T.unsqueeze(0)    # :101,105
        spec = out
    y = torch.fft.ifft(spec, dim=1).real                            # :112
    if bias is not None:
        y = y + bias                                                # :115-116
    return y


def fwd_bwd_port(x, w_re, w_im, bias, g):
    """Forward + autograd backward with upstream gradient g (random g, not y.sum())."""
    x = x.detach().clone().requires_grad_(True)
    w_re = w_re.detach().clone().requires_grad_(True)
    w_im = w_im.detach().clone().requires_grad_(True)
    bias = bias.detach().clone().requires_grad_(True)
    y = forward_port(x, w_re, w_im, bias)
    y.backward(g)
    return (y.detach(), x.grad.detach(), w_re.grad.detach(), w_im.grad.detach(),
            bias.grad.detach())


def block_half_port(x, ln_w, ln_b, eps, w_re, w_im, bias, g):
    """First residual line of SpectralMLPBlock.forward, spectral_layers.py:185 with :154-162:
    y = x + SpectralMixingLayer(LayerNorm(x)) (dropout inactive), and its autograd backward.
    Returns (y, grad_x, grad_ln_w, grad_ln_b, grad_w_re, grad_w_im, grad_bias)."""
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    x, ln_w, ln_b, w_re, w_im, bias = map(leaf, (x, ln_w, ln_b, w_re, w_im, bias))
    h = torch.nn.functional.layer_norm(x, (x.shape[-1],), ln_w, ln_b, eps)
    y = x + forward_port(h, w_re, w_im, bias)
    y.backward(g)
    return tuple(t.detach() for t in (y, x.grad, ln_w.grad, ln_b.grad, w_re.grad, w_im.grad,
                                      bias.grad))


def wirtinger_filter_port(x_freq, w_re, w_im):
    """wirtinger_ops.py:170-203: zero everything but bins [0,k), multiply those by W^T."""
    B, N, D = x_freq.shape
    k = num_bins(N, w_re.shape[1])
    w = torch.complex(w_re, w_im)
    out = torch.zeros_like(x_freq)
    out[:, :k, :] = x_freq[:, :k, :] * w[:, :k].T.unsqueeze(0)
    return out


def wirtinger_mul_backward(x, w, grad_out):
    """wirtinger_ops.py:67-82: grad_x = g*conj(w); grad_w = sum_b g*conj(x) (keepdim)."""
    gx = grad_out * np.conj(w)
    gw = (grad_out * np.conj(x)).sum(axis=0, keepdims=True)
    return gx, gw


def energy_ratio(x, y) -> float:
    """spectral_layers.py:122-132."""
    return float((np.asarray(y, np.float64) ** 2).sum() /
                 ((np.asarray(x, np.float64) ** 2).sum() + 1e-8))


# --------------------------------------------------------------------------------------
# float64 closed forms
# --------------------------------------------------------------------------------------
def _wT(w_re, w_im, k):
    w = np.asarray(w_re, np.float64) + 1j * np.asarray(w_im, np.float64)   # (D,F)
    return w[:, :k].T                                                      # (k,D)


def spectrum_closed(x, k):
    """First k bins of the DFT along axis 1: X[b,f,d] = sum_n x[b,n,d] e^{-2 pi i f n/N}."""
    x = np.asarray(x, np.float64)
    if k == 0:
        return np.zeros((x.shape[0], 0, x.shape[2]), np.complex128)
    return np.fft.rfft(x, axis=1)[:, :k, :]


def synth_closed(Y, N):
    """real(ifft) of a spectrum that is Y on bins [0,k) and zero elsewhere (incl. negative bins)."""
    B, k, D = Y.shape
    full = np.zeros((B, N, D), np.complex128)
    full[:, :k, :] = Y
    return np.fft.ifft(full, axis=1).real


def forward_closed(x, w_re, w_im, bias):
    B, N, D = x.shape
    k = num_bins(N, np.asarray(w_re).shape[1])
    X = spectrum_closed(x, k)
    y = synth_closed(X * _wT(w_re, w_im, k)[None], N)
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y, X


def backward_closed(x, w_re, w_im, g):
    """Returns grad_x (B,N,D), grad_w_re (D,F), grad_w_im (D,F), grad_bias (D) in float64."""
    B, N, D = x.shape
    F = np.asarray(w_re).shape[1]
    k = num_bins(N, F)
    X = spectrum_closed(x, k)
    G = spectrum_closed(g, k)
    gx = synth_closed(G * np.conj(_wT(w_re, w_im, k))[None], N)
    # d/dW of sum(g * real(ifft(pad(W X)))) : every kept bin contributes X*conj(G)/N
    P = (X * np.conj(G)).sum(axis=0) / N                  # (k,D)
    gw_re = np.zeros((D, F)); gw_im = np.zeros((D, F))
    gw_re[:, :k] = P.real.T
    gw_im[:, :k] = -P.imag.T
    gb = np.asarray(g, np.float64).sum(axis=(0, 1))
    return gx, gw_re, gw_im, gb


# --------------------------------------------------------------------------------------
# SURVEY 8(f) "next" rows: the callers either side of the layer.  Ports = the reference's op sequence
# in fp32 torch (bit-exact with the reference modules on every fixture, tests/test_oracle_next.py);
# *_closed_ex = fp64 closed forms of the general transform the native op computes.
# Reference lines restated:
#   fft_lm/train_fixed_full.py:507-555        causal FFT convolution of FixedSpectralBlock
#   fft_lm/frequency_native.py:95-121         FrequencyConvFunc forward / hand-written backward
#   fft_tensor/spectral_enhancements.py:147-164, :237-262   PhaseAware / MultiScale bands
#   fft_tensor/complex_rope.py:77-93, :207-216               ComplexRoPE table, layer transform
#   fft_tensor/frequency_ops.py:201           fnet_attention
# --------------------------------------------------------------------------------------
def next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


def causal_conv_port(x, kernel, gain, gate_freq_logits, g_ctx, cutoff, transition_bins):
    """train_fixed_full.py:507-555 on the already layer-normed x (B,T,C); g_ctx = sigmoid(gate_ctx(pooled))."""
    import torch.nn.functional as Fn
    B, T, C = x.shape
    K = kernel.shape[0]
    n_fft = next_pow2(T + K - 1)                                            # :507-509
    k = torch.zeros(n_fft, dtype=x.dtype)
    k[:K] = kernel
    k_freq = torch.fft.rfft(k)                                              # :513
    x_pad = Fn.pad(x, (0, 0, 0, n_fft - T))                                 # :516
    x_freq = torch.fft.rfft(x_pad, dim=1)                                   # :517
    y_freq = x_freq * k_freq.unsqueeze(0).unsqueeze(-1) * gain.unsqueeze(0).unsqueeze(0)      # :520
    Fbins = y_freq.size(1)
    g_freq = torch.sigmoid(gate_freq_logits[:Fbins])                        # :529
    y_freq = y_freq * g_freq.unsqueeze(0).unsqueeze(-1) * g_ctx.unsqueeze(1)                   # :536
    if cutoff is not None:                                                  # :540-551
        cutoff_idx = min(int(cutoff), Fbins)
        if cutoff_idx < Fbins:
            trans = min(transition_bins, cutoff_idx)
            mask = torch.ones(Fbins, dtype=x.dtype)
            start = cutoff_idx - trans
            if trans > 0:
                t = torch.linspace(0, 1, steps=trans, dtype=mask.dtype)
                mask[start:cutoff_idx] = 0.5 * (1.0 + torch.cos(torch.pi * t))
            mask[cutoff_idx:] = 0.0
            y_freq = y_freq * mask.unsqueeze(0).unsqueeze(-1)
    y_pad = torch.fft.irfft(y_freq, n=n_fft, dim=1)                         # :553
    return y_pad[:, :T, :]                                                  # :555


def freqconv_port(x_freq, kernel_freq, gain, grad_output):
    """frequency_native.py:95-121: forward product and the three hand-written gradients."""
    y = x_freq * kernel_freq.unsqueeze(0).unsqueeze(-1) * gain.unsqueeze(0).unsqueeze(0)
    gx = grad_output * kernel_freq.conj().unsqueeze(0).unsqueeze(-1) * gain.unsqueeze(0).unsqueeze(0)
    gk = (grad_output * x_freq.conj() * gain.unsqueeze(0).unsqueeze(0)).sum(dim=(0, 2))
    gg = (grad_output * x_freq * kernel_freq.unsqueeze(0).unsqueeze(-1)).real.sum(dim=(0, 1))
    return y, gx, gk, gg


def _cutoff_mask_port(y_freq, cutoff, transition_bins):
    """The cosine roll-off shared by the three fft_lm blocks (train_fixed_full.py:540-551 =
    frequency_native.py:341-351 = bicameral.py:193-203)."""
    Fbins = y_freq.size(1)
    if cutoff is None or min(int(cutoff), Fbins) >= Fbins:
        return y_freq
    cutoff_idx = min(int(cutoff), Fbins)
    trans = min(transition_bins, cutoff_idx)
    mask = torch.ones(Fbins, dtype=y_freq.real.dtype, device=y_freq.device)
    if trans > 0:
        t = torch.linspace(0, 1, steps=trans, dtype=mask.dtype, device=mask.device)
        mask[cutoff_idx - trans:cutoff_idx] = 0.5 * (1.0 + torch.cos(torch.pi * t))
    mask[cutoff_idx:] = 0.0
    return y_freq * mask.unsqueeze(0).unsqueeze(-1)


def phase_shift_port(z, phase_weights, magnitude_logits):
    """frequency_native.py:58-77: polar split, learned rotation, near-unity magnitude factor."""
    Fb = z.size(1)
    new_phase = z.angle() + (torch.tanh(phase_weights[:Fb]) * math.pi).unsqueeze(0)
    new_mag = z.abs() * (1.0 + 0.1 * torch.tanh(magnitude_logits[:Fb])).unsqueeze(0)
    return new_mag * torch.exp(1j * new_phase)


def spectral_layernorm_port(z, gamma, beta, eps=1e-5):
    """frequency_native.py:219-239."""
    mag, phase = z.abs(), z.angle()
    mean = mag.mean(dim=-1, keepdim=True)
    var = mag.var(dim=-1, keepdim=True, unbiased=False)
    Fb = z.size(1)
    scaled = (mag - mean) / torch.sqrt(var + eps) * gamma[:Fb].unsqueeze(0) + beta[:Fb].unsqueeze(0)
    return scaled * torch.exp(1j * phase)


def spectral_ffn_port(z, sd, prefix="ffn."):
    """frequency_native.py:152-200 in eval mode (no dropout)."""
    import torch.nn.functional as Fn
    z = spectral_layernorm_port(z, sd[prefix + "ln.gamma"], sd[prefix + "ln.beta"])
    w1, b1, w2, b2 = (sd[prefix + n] for n in ("w1.weight", "w1.bias", "w2.weight", "w2.bias"))
    h = torch.complex(Fn.linear(z.real, w1, b1), Fn.linear(z.imag, w1, b1))
    h = phase_shift_port(h, sd[prefix + "activation.phase_weights"], sd[prefix + "activation.magnitude_logits"])
    return torch.complex(Fn.linear(h.real, w2, b2), Fn.linear(h.imag, w2, b2))


def _kernel_freq_port(kernel, n_fft):
    k = torch.zeros(n_fft, dtype=kernel.dtype, device=kernel.device)
    k[:kernel.shape[0]] = kernel
    return torch.fft.rfft(k)


def freq_native_block_port(sd, x, cutoff, transition_bins):
    """FrequencyNativeBlock.forward, frequency_native.py:296-362, dropout off.  sd: the module's state_dict."""
    import torch.nn.functional as Fn
    residual = x
    x = Fn.layer_norm(x, x.shape[-1:], sd["ln.weight"], sd["ln.bias"])
    B, T, C = x.shape
    n_fft = next_pow2(T + sd["kernel"].shape[0] - 1)
    x_freq = torch.fft.rfft(Fn.pad(x, (0, 0, 0, n_fft - T)), dim=1)                                   # :314-315
    y_freq = x_freq * _kernel_freq_port(sd["kernel"], n_fft).view(1, -1, 1) * sd["gain"].view(1, 1, -1)  # :325
    Fb = y_freq.size(1)
    g_freq = torch.sigmoid(sd["gate_freq_logits"][:Fb])
    g_ctx = torch.sigmoid(Fn.linear(x.mean(dim=1), sd["gate_ctx.weight"], sd["gate_ctx.bias"]))
    y_freq = y_freq * g_freq.view(1, -1, 1) * g_ctx.unsqueeze(1)                                      # :338
    y_freq = _cutoff_mask_port(y_freq, cutoff, transition_bins)
    y_freq = y_freq + spectral_ffn_port(y_freq, sd)                                                   # :355-356
    return residual + torch.fft.irfft(y_freq, n=n_fft, dim=1)[:, :T, :]                               # :359-362


def bicameral_block_port(sd, x, cutoff, transition_bins):
    """BicameralBlock.forward, bicameral.py:134-275, dropout off."""
    import torch.nn.functional as Fn
    residual = x
    C = x.shape[-1]
    x = Fn.layer_norm(x, (C,), sd["ln.weight"], sd["ln.bias"])
    B, T, _ = x.shape
    pooled = x.mean(dim=1)
    n_fft = next_pow2(T + sd["kernel_freq"].shape[0] - 1)
    x_freq = torch.fft.rfft(Fn.pad(x, (0, 0, 0, n_fft - T)), dim=1)                                   # :170-171
    y_freq = x_freq * _kernel_freq_port(sd["kernel_freq"], n_fft).view(1, -1, 1) * sd["gain_freq"].view(1, 1, -1)
    Fb = y_freq.size(1)
    g_freq = torch.sigmoid(sd["gate_freq_logits"][:Fb])
    g_ctx = torch.sigmoid(Fn.linear(pooled, sd["gate_ctx_freq.weight"], sd["gate_ctx_freq.bias"]))
    y_freq = y_freq * g_freq.view(1, -1, 1) * g_ctx.unsqueeze(1)                                      # :186
    y_freq = phase_shift_port(y_freq, sd["phase_activation.phase_weights"], sd["phase_activation.magnitude_logits"])
    y_freq = _cutoff_mask_port(y_freq, cutoff, transition_bins)
    y_spectral = torch.fft.irfft(y_freq, n=n_fft, dim=1)[:, :T, :]                                    # :206-207
    shifted = Fn.pad(x.transpose(1, 2)[:, :, :-1], (1, 0))                                            # :221
    y_time = Fn.conv1d(shifted, sd["conv1d.weight"], sd["conv1d.bias"], padding=1, groups=C).transpose(1, 2)
    y_time = y_time * torch.sigmoid(Fn.linear(pooled, sd["gate_time.weight"], sd["gate_time.bias"])).unsqueeze(1)
    a_f, a_t = torch.sigmoid(sd["alpha_freq"]), torch.sigmoid(sd["alpha_time"])
    total = a_f + a_t + 1e-8
    y_cross = Fn.linear(torch.cat([y_spectral, y_time], dim=-1), sd["cross_interact.weight"], sd["cross_interact.bias"])
    out = residual + (a_f / total) * y_spectral + (a_t / total) * y_time + 0.1 * y_cross              # :261-269
    ff = Fn.layer_norm(out, (C,), sd["ffn_ln.weight"], sd["ffn_ln.bias"])
    ff = Fn.linear(Fn.gelu(Fn.linear(ff, sd["ffn.0.weight"], sd["ffn.0.bias"])), sd["ffn.3.weight"], sd["ffn.3.bias"])
    return out + ff                                                                                   # :272-273


def phase_aware_port(x, magnitude_filter, phase_filter):
    """spectral_enhancements.py:147-164."""
    x_freq = torch.fft.rfft(x, dim=1)
    magnitude, phase = torch.abs(x_freq), torch.angle(x_freq)
    fm = magnitude * magnitude_filter[:x_freq.size(-1)]
    fp = phase + phase_filter[:x_freq.size(-1)]
    return torch.fft.irfft(torch.polar(fm, fp), n=x.size(1), dim=1)


def multiscale_bands_port(x):
    """spectral_enhancements.py:237-262: the three band-limited reconstructions (before the Linears)."""
    x_freq = torch.fft.rfft(x, dim=1)
    K = x_freq.size(1)
    low_k, mid_k = K // 4, K // 2
    out = []
    for lo, hi in ((0, low_k), (low_k, mid_k), (mid_k, K)):
        band = torch.zeros_like(x_freq)
        band[:, lo:hi] = x_freq[:, lo:hi]
        out.append(torch.fft.irfft(band, n=x.size(1), dim=1))
    return out


def complex_rope_mix_port(x, rotation, freq_filter):
    """complex_rope.py:207-216 on the normed x: fft -> ComplexRoPE (:77-93) -> filter -> ifft.real."""
    B, T, D = x.shape
    x_freq = torch.fft.fft(x, dim=1)
    rot = rotation[:T]
    pairs = x_freq.reshape(B, T, D // 2, 2)
    x0, x1 = pairs[..., 0] * rot.unsqueeze(0), pairs[..., 1] * rot.unsqueeze(0)
    x_freq = torch.stack([x0, x1], dim=-1).reshape(B, T, D)
    x_freq = x_freq * freq_filter.unsqueeze(0).unsqueeze(0)
    return torch.fft.ifft(x_freq, dim=1).real


def fnet_port(x_freq):
    """frequency_ops.py:201."""
    return torch.fft.fft(x_freq, dim=1)


def forward_closed_ex(x, w_re, w_im, bias, n_fft, k):
    """y[:, :R] = real(ifft_n(pad_k(W * rfft_n(zero-pad(x))[:k])))[:, :R] + bias in float64; also returns
    the kept spectrum.  k may be n_fft // 2 + 1 (Nyquist included)."""
    x = np.asarray(x, np.float64)
    B, R, D = x.shape
    X = np.fft.rfft(x, n=n_fft, axis=1)[:, :k, :]
    full = np.zeros((B, n_fft, D), np.complex128)
    full[:, :k, :] = X * _wT(w_re, w_im, k)[None]
    y = np.fft.ifft(full, axis=1).real[:, :R, :]
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y, X


def backward_closed_ex(x, w_re, w_im, g, n_fft, k):
    """Gradients of forward_closed_ex for upstream g (B,R,D): grad_x, grad_w_re, grad_w_im, grad_bias."""
    x = np.asarray(x, np.float64); g = np.asarray(g, np.float64)
    B, R, D = x.shape
    F = np.asarray(w_re).shape[1]
    X = np.fft.rfft(x, n=n_fft, axis=1)[:, :k, :]
    G = np.fft.rfft(g, n=n_fft, axis=1)[:, :k, :]
    full = np.zeros((B, n_fft, D), np.complex128)
    full[:, :k, :] = G * np.conj(_wT(w_re, w_im, k))[None]
    gx = np.fft.ifft(full, axis=1).real[:, :R, :]
    P = (X * np.conj(G)).sum(axis=0) / n_fft
    gw_re = np.zeros((D, F)); gw_im = np.zeros((D, F))
    gw_re[:, :k] = P.real.T
    gw_im[:, :k] = -P.imag.T
    return gx, gw_re, gw_im, g.sum(axis=(0, 1))
