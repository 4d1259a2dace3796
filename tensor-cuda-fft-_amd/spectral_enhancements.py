"""Drop-in mirrors of the sequence mixers of `fft_tensor.spectral_enhancements` that share the layer's
transform: `PhaseAwareSpectralMixing` (reference fft_tensor/spectral_enhancements.py:118-166)
and `MultiScaleSpectralFeatures` (:214-275).  Same constructors, attribute names and state_dict keys; the
rfft -> filter -> irfft of each runs as the fused HIP transform (functional.spectral_filter), not torch.fft.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .functional import hermitian_scale, phase_filter, spectral_filter


class PhaseAwareSpectralMixing(nn.Module):
    """irfft(polar(|X| * m, angle(X) + p)) with X = rfft(x, dim=1) and per-CHANNEL m, p (the reference
    indexes its (dim,) filters with `x_freq.size(-1)`, the channel count, :154-158).

    |X| m e^{i (angle X + p)} = X * (m e^{i p}): one complex constant per channel on every bin, so the
    whole layer is the fused transform with W[d, f] = m[d] e^{i p[d]} on all T//2 + 1 bins (irfft keeps
    only the real part of the DC / Nyquist products, exactly like the kernel).  The gradients of
    magnitude_filter / phase_filter follow from the native grad_W through this construction."""

    def __init__(self, dim, learnable=True):
        super().__init__()
        self.dim = dim
        if learnable:
            self.magnitude_filter = nn.Parameter(torch.ones(dim))       # reference :132-133
            self.phase_filter = nn.Parameter(torch.zeros(dim))
        else:
            self.register_buffer("magnitude_filter", torch.ones(dim))
            self.register_buffer("phase_filter", torch.zeros(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        K = T // 2 + 1
        m, p = self.magnitude_filter[:D], self.phase_filter[:D]         # :154, :157
        if x.is_cuda and m.dtype == torch.float32 and p.dtype == torch.float32:
            w_re, w_im = phase_filter(m, p, K, T)                        # one native launch (and one for its gradients)
        else:
            c = hermitian_scale(T, K, x.device)                          # irfft semantics (:164)
            w_re = (m * torch.cos(p)).unsqueeze(1) * c.unsqueeze(0)      # (D, K)
            w_im = (m * torch.sin(p)).unsqueeze(1) * c.unsqueeze(0)
        return spectral_filter(x, w_re, w_im, None, n_fft=T, k=K)


class MultiScaleSpectralFeatures(nn.Module):
    """Three band-limited copies of x -- rfft bins [0, K/4), [K/4, K/2), [K/2, K) -- each through its own
    Linear, then fused (reference :214-275).  The bands partition the spectrum, so with
    P_k = "keep the first k bins" (one fused native transform each):
        low = P_{K//4} x,   mid = P_{K//2} x - P_{K//4} x,   high = x - P_{K//2} x
    two pruned transforms instead of one full rfft and three full irffts."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.low_freq = nn.Linear(dim, dim)
        self.mid_freq = nn.Linear(dim, dim)
        self.high_freq = nn.Linear(dim, dim)
        self.fusion = nn.Linear(dim * 3, dim)
        self._w_cache = {}

    def _lowpass(self, x: torch.Tensor, k: int) -> torch.Tensor:
        if k <= 0:
            return torch.zeros_like(x)
        B, T, D = x.shape
        key = (T, D, k, x.device)
        w = self._w_cache.get(key)
        if w is None:
            w_re = hermitian_scale(T, k, x.device).unsqueeze(0).expand(D, k).contiguous()
            w = self._w_cache[key] = (w_re, torch.zeros_like(w_re))
        return spectral_filter(x, w[0], w[1], None, n_fft=T, k=k)

    def bands(self, x: torch.Tensor):
        """(low, mid, high) = the three irfft(...) of reference :246-262."""
        K = x.shape[1] // 2 + 1
        low_k, mid_k = K // 4, K // 2                                    # :242-243
        p_low = self._lowpass(x, low_k)
        p_mid = self._lowpass(x, mid_k)
        return p_low, p_mid - p_low, x - p_mid

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        low, mid, high = self.bands(x)
        combined = torch.cat([self.low_freq(low), self.mid_freq(mid), self.high_freq(high)], dim=-1)
        return self.fusion(combined)                                     # :265-272
