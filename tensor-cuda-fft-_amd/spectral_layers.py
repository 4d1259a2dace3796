"""Drop-in mirror of the reference module `fft_tensor.spectral_layers` for the hot path.

Same constructor, attributes, state_dict keys (`weight_real`, `weight_imag`, `bias`), assertion
text and method names as the reference `SpectralMixingLayer`
(reference fft_tensor/spectral_layers.py:19-132), so reference checkpoints load unchanged and
callers (`SpectralMLPBlock`, reference :135-190) keep working.  The arithmetic is not torch.fft:
it is the fused HIP path in csrc/ reached through the C ABI of include/smx.h.

Documented deviations from the reference:
  * inputs must be float32 on a ROCm device (the reference would run anywhere torch runs and
    promote float64); anything else raises instead of silently taking a slower path;
  * `learnable=False` returns the input itself (the reference computes ifft(fft(x)).real, which is
    x to 1.2e-7);
  * training-mode dropout (p > 0) is drawn inside the transform's launches from the library's
    counter-based generator (keyed per call from torch's device generator): same distribution and
    scaling as nn.Dropout, different random bits, p quantised to 1/65536.  `layer.fuse_dropout = False` puts
    nn.Dropout back as a separate pass.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .functional import DropoutState, block_supported, spectral_block_mix, spectral_mix


class SpectralMixingLayer(nn.Module):
    """FFT along the sequence axis -> learnable complex low-pass filter -> inverse FFT -> bias.

    y = real(ifft(pad(W[:, :k].T * fft(x, dim=1)[:, :k, :]), dim=1)) + bias,  k = min(num_filters, T//2)
    """

    def __init__(self, embed_dim: int, num_filters: Optional[int] = None, dropout: float = 0.0,
                 learnable: bool = True):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_filters = num_filters or (embed_dim // 2)           # reference :51
        self.learnable = learnable
        if learnable:
            self.weight_real = nn.Parameter(torch.ones(embed_dim, self.num_filters))    # :57
            self.weight_imag = nn.Parameter(torch.zeros(embed_dim, self.num_filters))   # :58
            self.bias = nn.Parameter(torch.zeros(embed_dim))                            # :61
        else:
            self.register_parameter("weight_real", None)
            self.register_parameter("weight_imag", None)
            self.register_parameter("bias", None)
        self.dropout = nn.Dropout(dropout)
        self._verify_gradients = True                                                    # :71
        # set by distributed.attach_grad_sync(): overlaps the filter-gradient all-reduce with grad_x
        self._grad_sync = None
        # training-mode dropout runs inside the transform's launches (own counter-based generator, seeded
        # from torch's); False keeps nn.Dropout as a separate pass with torch's generator
        self.fuse_dropout = True
        self._drop_state = None

    def _fused_dropout_p(self) -> float:
        """Drop probability to hand to the native op, 0.0 when nn.Dropout (or nothing) applies instead."""
        p = float(self.dropout.p)
        return p if (self.training and self.fuse_dropout and 0.0 < p < 1.0) else 0.0

    def _eight_band_split(self, B: int, T: int) -> bool:
        """The one case the native dropout does not serve: the eight-band plan (2048 rows, more than 512 kept bins, library
        option fourstep = 0) under a phase-split backward (an attached gradient sync).  Everything else -- every plan,
        whatever the bin count -- takes the native dropout and the native block line (round 4: on the plans for more than
        512 bins the mask is one more native pass, the block's LayerNorm a separate pass)."""
        if self._grad_sync is None or T != 2048 or min(self.num_filters, T // 2) <= 512:
            return False
        from . import _lib
        return _lib.plan(B, T, self.embed_dim, self.num_filters).bands == 8

    def _dropout_state(self, device: torch.device) -> DropoutState:
        if self._drop_state is None or self._drop_state.device != device:
            self._drop_state = DropoutState(device)
        return self._drop_state

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        assert D == self.embed_dim, f"Expected embed_dim={self.embed_dim}, got {D}"     # :84
        if self.learnable and self.weight_real is not None:
            p = self._fused_dropout_p() if x.is_cuda and not self._eight_band_split(B, T) else 0.0
            if p > 0.0:                                                                 # :118 fused
                return spectral_mix(x, self.weight_real, self.weight_imag, self.bias, self._grad_sync,
                                    dropout_p=p, drop_state=self._dropout_state(x.device))
            y = spectral_mix(x, self.weight_real, self.weight_imag, self.bias, self._grad_sync)
        else:
            y = x.clone()
        return self.dropout(y)

    def verify_energy_preservation(self, x: torch.Tensor, y: torch.Tensor) -> float:
        """sum(y^2) / (sum(x^2) + 1e-8)  (reference :122-132)."""
        energy_in = torch.sum(x ** 2).item()
        energy_out = torch.sum(y ** 2).item()
        return energy_out / (energy_in + 1e-8)

    def extra_repr(self) -> str:
        return (f"embed_dim={self.embed_dim}, num_filters={self.num_filters}, "
                f"learnable={self.learnable}")


class SpectralMLPBlock(nn.Module):
    """Immediate caller of the hot path (reference fft_tensor/spectral_layers.py:135-190):
    x + spectral_mix(norm1(x)), then x + mlp(norm2(x)).  The first line runs as one fused native op
    (LayerNorm inside the transform's load, dropout and residual inside its store, smx_block_forward);
    norm2 and the MLP stay on torch/hipBLASLt.  Attribute
    names match the reference so `spectral_mix.weight_real`, `norm1.weight` etc. load from
    reference checkpoints.  `fuse_norm=False` keeps the three ops separate."""

    def __init__(self, embed_dim: int, mlp_ratio: int = 4, dropout: float = 0.1):
        super().__init__()
        self.spectral_mix = SpectralMixingLayer(embed_dim=embed_dim, dropout=dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        mlp_dim = embed_dim * mlp_ratio
        self.mlp = nn.Sequential(nn.Linear(embed_dim, mlp_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(mlp_dim, embed_dim), nn.Dropout(dropout))

        self.fuse_norm = True

    def _fusable(self, x: torch.Tensor) -> bool:
        sm = self.spectral_mix
        active = self.training and sm.dropout.p > 0.0          # a dropout the native op cannot take over
        return (self.fuse_norm and sm.learnable and x.dim() == 3 and x.shape[-1] == sm.embed_dim
                and x.is_cuda and x.dtype == torch.float32 and not sm._eight_band_split(x.shape[0], x.shape[1])
                and not (active and sm._fused_dropout_p() == 0.0)
                and block_supported(sm.embed_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fusable(x):
            sm, n1 = self.spectral_mix, self.norm1
            p = sm._fused_dropout_p()
            x = spectral_block_mix(x, n1.weight, n1.bias, n1.eps, sm.weight_real, sm.weight_imag,
                                   sm.bias, sm._grad_sync, dropout_p=p,
                                   drop_state=sm._dropout_state(x.device) if p > 0.0 else None)  # :185
        else:
            x = x + self.spectral_mix(self.norm1(x))
        x = x + self.mlp(self.norm2(x))                                                 # :188
        return x


class HybridSpectralAttention(nn.Module):
    """Third public class of the reference module (fft_tensor/spectral_layers.py:193-256): global context
    from the spectral mix, then dense multi-head attention over `norm(x + context)`, residual to x.
    Attribute names and state_dict keys are the reference's (`spectral.*`, `qkv`, `proj`, `norm`).

    Only the spectral mix is this package's own kernel; the attention is GEMM work and goes through
    torch's fused scaled-dot-product attention (same arithmetic as the reference's explicit
    softmax(q k^T / sqrt(d)) v with dropout on the attention weights, without materialising the (T, T)
    matrix the reference builds -- 34 GB at (64, 4096, 256) with 8 heads)."""

    def __init__(self, embed_dim: int, num_heads: int = 8, window_size: int = 64, dropout: float = 0.1):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.window_size = window_size                       # kept for API parity; the reference ignores it too
        self.spectral = SpectralMixingLayer(embed_dim, dropout=dropout)
        self.qkv = nn.Linear(embed_dim, 3 * embed_dim)
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        global_context = self.spectral(x)                                                # :236
        qkv = self.qkv(self.norm(x + global_context))                                    # :240
        qkv = qkv.reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        p = self.dropout.p if self.training else 0.0
        out = torch.nn.functional.scaled_dot_product_attention(q, k, v, dropout_p=p)      # :246-251
        out = out.transpose(1, 2).reshape(B, T, D)
        return x + self.dropout(self.proj(out))                                          # :252-255
