"""Drop-in mirror of the causal FFT convolution block the reference actually trains with:
`fft_lm.train_fixed_full.FixedSpectralBlock` (reference fft_lm/train_fixed_full.py:427-563) and the
frequency-domain multiply of its twin, `fft_lm.frequency_native.FrequencyConvFunc` (:80-121).
Same constructor, attribute names and state_dict keys (`kernel`, `gain`, `gate_freq_logits`, `ln.*`,
`gate_ctx.*`, `ffn_ln.*`, `ffn.*`), so reference checkpoints load unchanged.

The hot loop of that model is
    y = irfft( rfft(pad(x)) * k_freq * gain * sigmoid(gate_freq) * g_ctx * cutoff_mask, n )[:, :T]
Everything that multiplies the spectrum is either per frequency (shared by all channels) or per
(batch, channel), so the whole line is ONE fused native transform -- zero-padded load, full one-sided
spectrum incl. Nyquist, cropped store (functional.spectral_filter, smx_forward_ex) -- with
W[c, f] = H[f] * gain[c],  H = k_freq * sigmoid(gate_freq) * mask,  and the context gate applied to the
cropped output.  Parameter gradients follow from the native grad_W through that construction.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _lib
from .functional import _stream, conv_response, conv_supported, hermitian_scale, rank_one_conv, spectral_filter


def next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


_dft_cache: dict = {}


def _kernel_dft(n_fft: int, kernel_len: int, device) -> tuple:
    """(cos, -sin) matrices (n_fft//2+1, kernel_len), generated in fp64: the spectrum of the zero-padded
    causal kernel -- reference :511-513, rfft of n_fft points of which kernel_len are non-zero -- is a
    (1025 x 128) matrix-vector product at the default sizes, differentiable in the taps.  (The transform
    of the ACTIVATIONS is the native one; this is the filter's own 128-tap response.)"""
    key = (n_fft, kernel_len, str(device))
    m = _dft_cache.get(key)
    if m is None:
        f = torch.arange(n_fft // 2 + 1, dtype=torch.float64).unsqueeze(1)
        t = torch.arange(kernel_len, dtype=torch.float64).unsqueeze(0)
        ang = 2.0 * math.pi * ((f * t) % n_fft) / n_fft
        m = _dft_cache[key] = (torch.cos(ang).float().to(device), (-torch.sin(ang)).float().to(device))
    return m


def _kernel_response(kernel: torch.Tensor, n_fft: int) -> tuple:
    """(Re, Im) of rfft(zero-pad(kernel), n_fft) (reference :511-513), differentiable in the taps.  Short transforms:
    the cached DFT matrix product above.  From n_fft = 8192 on that matrix is 4 MiB and more per part and its two
    matrix-vector products cost more than the convolution's own launches (measured 2 x 352 us at n_fft = 65536):
    there the taps go through the native transform as a (1, K, 2) tensor, zero-padded load."""
    K = kernel.shape[0]
    if n_fft >= 8192 and kernel.is_cuda and kernel.dtype == torch.float32:
        from .functional import rfft
        spec = rfft(kernel.view(1, K, 1).expand(1, K, 2).contiguous(), n_fft)[0, :, 0]
        return spec.real, spec.imag
    cm, sm = _kernel_dft(n_fft, K, kernel.device)
    return cm @ kernel, sm @ kernel


def cutoff_mask(cutoff, fbins: int, transition_bins: int, device) -> "torch.Tensor | None":
    """Progressive frequency horizon of reference :540-551: 1 up to cutoff - trans, cosine roll-off, 0
    from cutoff on.  None when nothing is cut."""
    if cutoff is None:
        return None
    cutoff_idx = min(int(cutoff), fbins)
    if cutoff_idx >= fbins:
        return None
    trans = min(transition_bins, cutoff_idx)
    mask = torch.ones(fbins, device=device, dtype=torch.float32)
    start = cutoff_idx - trans
    if trans > 0:
        t = torch.linspace(0, 1, steps=trans, device=device, dtype=mask.dtype)
        mask[start:cutoff_idx] = 0.5 * (1.0 + torch.cos(torch.pi * t))
    mask[cutoff_idx:] = 0.0
    return mask


def causal_spectral_conv(x: torch.Tensor, kernel: torch.Tensor, gain: torch.Tensor,
                         gate_freq_logits: "torch.Tensor | None" = None,
                         g_ctx: "torch.Tensor | None" = None, cutoff=None,
                         transition_bins: int = 1) -> torch.Tensor:
    """Lines :507-555 of the reference block for x (B, T, C): causal linear convolution with the
    time-domain `kernel` via a zero-padded transform of length next_pow2(T + K - 1), per-channel `gain`,
    per-frequency sigmoid gate, per-(batch, channel) context gate `g_ctx` (already in [0, 1]) and the
    cosine cutoff mask; returns the first T samples."""
    B, T, C = x.shape
    K = kernel.shape[0]
    n_fft = next_pow2(T + K - 1)                                           # :507-509
    fbins = n_fft // 2 + 1
    if conv_supported(B, T, C, n_fft) and x.is_cuda and x.dtype == torch.float32:
        # n_fft 512 ... 65536: the convolution's own kernels -- the packed
        # spectrum times the Hermitian extension of H, gain x context gate at the store (smx_conv_*).
        # H = k_freq (:511-513) x sigmoid(gate) (:529) x mask (:551): one native launch (and one for its gradients)
        mask = cutoff_mask(cutoff, fbins, transition_bins, x.device)
        if kernel.dtype == torch.float32 and (gate_freq_logits is None or gate_freq_logits.dtype == torch.float32):
            h_re, h_im = conv_response(kernel, gate_freq_logits, mask, n_fft)
        else:
            h_re, h_im = _kernel_response(kernel, n_fft)
            per_f = None
            if gate_freq_logits is not None:
                per_f = torch.sigmoid(gate_freq_logits[:fbins])
            if mask is not None:
                per_f = mask if per_f is None else per_f * mask
            if per_f is not None:
                h_re, h_im = h_re * per_f, h_im * per_f
        s = gain.unsqueeze(0).expand(B, C) if g_ctx is None else gain.unsqueeze(0) * g_ctx   # :522, :533-536
        return rank_one_conv(x, h_re, h_im, s.contiguous(), n_fft)
    h_re, h_im = _kernel_response(kernel, n_fft)                           # k_freq, :511-513
    scale = hermitian_scale(n_fft, fbins, x.device)                        # irfft semantics, :553
    if gate_freq_logits is not None:
        scale = scale * torch.sigmoid(gate_freq_logits[:fbins])            # :529
    mask = cutoff_mask(cutoff, fbins, transition_bins, x.device)
    if mask is not None:
        scale = scale * mask                                               # :551
    h_re, h_im = h_re * scale, h_im * scale
    w_re = gain.unsqueeze(1) * h_re.unsqueeze(0)                           # (C, fbins), :522
    w_im = gain.unsqueeze(1) * h_im.unsqueeze(0)
    # :515-519, :553-555; the context gate (:533-536) rides along as the per-(batch, channel) factor
    return spectral_filter(x, w_re, w_im, None, n_fft=n_fft, k=fbins, row_scale=g_ctx)


class FixedSpectralBlock(nn.Module):
    """Pre-norm causal spectral mixing + gated valve + FFN residual (reference :427-563)."""

    def __init__(self, d_model: int, seq_len: int, kernel_len: int, transition_bins: int, dropout: float = 0.1):
        super().__init__()
        self.ln = nn.LayerNorm(d_model)
        self.drop = nn.Dropout(dropout)
        self.seq_len = seq_len
        self.kernel_len = kernel_len
        self.transition_bins = int(max(1, transition_bins))
        self.kernel = nn.Parameter(torch.zeros(kernel_len))
        nn.init.normal_(self.kernel, mean=0.0, std=0.001)                  # :447-448
        self.gain = nn.Parameter(torch.ones(d_model))                      # :451
        self.max_freq_bins = next_pow2(int(seq_len + kernel_len - 1)) // 2 + 1      # :468-472
        self.gate_freq_logits = nn.Parameter(torch.ones(self.max_freq_bins) * 2.0)  # :475
        self.gate_ctx = nn.Linear(d_model, d_model)
        nn.init.zeros_(self.gate_ctx.weight)
        nn.init.constant_(self.gate_ctx.bias, 2.0)                         # :478-480
        hidden = d_model * 2
        self.ffn_ln = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(nn.Linear(d_model, hidden), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden, d_model))
        for m in self.ffn:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor, cutoff: "int | None" = None) -> torch.Tensor:
        residual = x
        x = self.ln(x)
        g_ctx = torch.sigmoid(self.gate_ctx(x.mean(dim=1)))                # :532-533
        y = causal_spectral_conv(x, self.kernel, self.gain, self.gate_freq_logits, g_ctx, cutoff,
                                 self.transition_bins)
        x = residual + self.drop(y)                                        # :557-558
        return x + self.drop(self.ffn(self.ffn_ln(x)))                     # :561-562


class FrequencyConvFunc(torch.autograd.Function):
    """y_freq = x_freq * kernel_freq[None, :, None] * gain[None, None, :] with the reference's hand-written
    backward (fft_lm/frequency_native.py:80-121), on the native complex-multiply kernels
    (smx_cmul / smx_cmul_grad_w).  x_freq (B, F, C) complex64, kernel_freq (F) complex64, gain (C) fp32."""

    @staticmethod
    def forward(ctx, x_freq, kernel_freq, gain):
        if not x_freq.is_cuda or x_freq.dtype != torch.complex64:
            raise TypeError("x_freq must be a complex64 tensor on a ROCm device")
        x = x_freq.resolve_conj().contiguous()
        w = (kernel_freq.to(torch.complex64).unsqueeze(1) * gain.unsqueeze(0)).contiguous()    # (F, C)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().smx_cmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.shape[0],
                                           w.numel(), 0, _stream(x.device)))
        ctx.save_for_backward(x, kernel_freq, gain, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        x, kernel_freq, gain, w = ctx.saved_tensors
        g = grad_output.resolve_conj().contiguous()
        B, inner = x.shape[0], w.numel()
        grad_x = torch.empty_like(x)
        s_conj = torch.empty_like(w)          # sum_b g * conj(x)
        s_plain = torch.empty_like(w)         # sum_b g * x
        xc = x.conj().resolve_conj()
        with torch.cuda.device(x.device):
            s = _stream(x.device)
            lib = _lib.lib()
            _lib.check(lib.smx_cmul(g.data_ptr(), w.data_ptr(), grad_x.data_ptr(), B, inner, 1, s))     # :111
            _lib.check(lib.smx_cmul_grad_w(x.data_ptr(), g.data_ptr(), s_conj.data_ptr(), B, inner, s))
            _lib.check(lib.smx_cmul_grad_w(xc.data_ptr(), g.data_ptr(), s_plain.data_ptr(), B, inner, s))
        grad_kernel = (s_conj * gain.unsqueeze(0)).sum(dim=1)                                 # :114
        grad_gain = (s_plain * kernel_freq.unsqueeze(1)).real.sum(dim=0)                      # :117
        return grad_x, grad_kernel, grad_gain
