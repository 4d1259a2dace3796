"""Drop-in mirror of `fft_tensor.complex_rope` (reference fft_tensor/complex_rope.py): `ComplexRoPE`,
`GatedLinearUnit`, `ComplexRoPESpectralLayer` with the reference's constructors, buffers and state_dict
keys.  The fft -> rotate -> per-channel complex filter -> ifft(.).real of the layer (:207-216) runs as ONE
fused native transform (functional.spectral_filter); the GLU and the norms are GEMM / row work on torch.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .functional import spectral_filter


class ComplexRoPE(nn.Module):
    """Multiply element (t, d) of a complex (B, T, D) tensor by e^{i t theta_{d // 2}} (reference :15-98)."""

    def __init__(self, dim, max_seq_len=4096, base=10000):
        super().__init__()
        self.dim = dim
        self.max_seq_len = max_seq_len
        self.base = base
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))       # :35
        self.register_buffer("inv_freq", inv_freq)
        self._cache_rotations(max_seq_len)

    def _cache_rotations(self, max_len):
        t = torch.arange(max_len, dtype=torch.float32)
        freqs = torch.outer(t, self.inv_freq)                                     # :46
        self.register_buffer("rotation", torch.complex(torch.cos(freqs), torch.sin(freqs)))

    def table(self, T: int) -> torch.Tensor:
        """(T, D) complex: the factor of element (t, d), i.e. rotation[t, d // 2] (:77-93)."""
        return self.rotation[:T].repeat_interleave(2, dim=1)

    def forward(self, x_freq: torch.Tensor) -> torch.Tensor:
        B, T, D = x_freq.shape
        if not torch.is_complex(x_freq):
            raise ValueError("ComplexRoPE requires complex input from FFT")      # :71-72
        return x_freq * self.table(T).unsqueeze(0)

    def apply_to_fft(self, x: torch.Tensor) -> torch.Tensor:
        """ifft(rope(fft(x))).real (:100-118) as one fused pass."""
        B, T, D = x.shape
        w = _one_sided(self.table(T), T)
        return spectral_filter(x, w.real.T.contiguous(), w.imag.T.contiguous(), None, n_fft=T, k=T // 2 + 1)


def _one_sided(r: torch.Tensor, T: int) -> torch.Tensor:
    """Two-sided per-bin factors R[f, d], f = 0..T-1, applied to the spectrum of a REAL signal before
    ifft(.).real, folded onto the T//2 + 1 one-sided bins:  Re sum_f X_f R_f e^{..} with X_{T-f} = conj X_f
    equals the one-sided sum with W_f = R_f + conj(R_{T-f}) on the bins that have a mirror image."""
    K = T // 2 + 1
    m = (T - 1) // 2
    w = r[:K].clone()
    if m > 0:
        w[1:m + 1] = r[1:m + 1] + torch.flip(r[T - m:], dims=(0,)).conj()
    return w


class GatedLinearUnit(nn.Module):
    """sigmoid(gate_proj(x)) * value_proj(x) -> out_proj (reference :121-159)."""

    def __init__(self, dim):
        super().__init__()
        self.gate_proj = nn.Linear(dim, dim)
        self.value_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.out_proj(torch.sigmoid(self.gate_proj(x)) * self.value_proj(x))


class ComplexRoPESpectralLayer(nn.Module):
    """norm1 -> fft -> ComplexRoPE -> freq_filter -> ifft.real -> residual; norm2 -> GLU -> residual
    (reference :162-226)."""

    def __init__(self, dim, dropout=0.1):
        super().__init__()
        self.dim = dim
        self.rope = ComplexRoPE(dim)
        self.freq_filter = nn.Parameter(torch.ones(dim, dtype=torch.complex64))   # :181
        self.glu = GatedLinearUnit(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        residual = x
        x = self.norm1(x)
        B, T, D = x.shape
        r = self.rope.table(T) * self.freq_filter.unsqueeze(0)                    # :210-213, (T, D) complex
        w = _one_sided(r, T)                                                      # (T//2+1, D)
        x = spectral_filter(x, w.real.T.contiguous(), w.imag.T.contiguous(), None, n_fft=T,
                            k=T // 2 + 1)                                         # :207-216
        x = residual + self.dropout(x)
        residual = x
        x = self.glu(self.norm2(x))
        return residual + self.dropout(x)
