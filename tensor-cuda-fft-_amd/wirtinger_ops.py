"""Mirror of the reference module `fft_tensor.wirtinger_ops` (reference fft_tensor/wirtinger_ops.py).

`WirtingerGradient`, `ComplexParameter` and `WirtingerSpectralFilter` keep the reference's names,
constructor arguments, parameter names (`weight.real`, `weight.imag`) and error behaviour; the
complex multiply, its Wirtinger backward and the zero-filled filter run as HIP kernels through
include/smx.h.  `spectral_mix_with_filter` is the fused form of `ifft(filter(fft(x))).real`, the
unit BASELINE config 5 measures.
"""
from __future__ import annotations

import math

import torch
from torch.autograd.function import once_differentiable
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .functional import _require_gpu_f32, _stream, spectral_mix


def _require_gpu_c64(name: str, t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: this path has no CPU implementation")
    if t.dtype != torch.complex64:
        raise TypeError(f"{name} must be complex64, got {t.dtype}")


class WirtingerGradient(Function):
    """out = x_freq * weight_complex with the Wirtinger backward of reference :53-82:
    grad_x = g * conj(w); grad_w = (g * conj(x)).sum(dim=0, keepdim=True).
    The weight must broadcast over the leading dimension only (the reference's call site,
    :192-194, passes (1, k, D) against (B, k, D))."""

    @staticmethod
    def forward(ctx, x_freq: torch.Tensor, weight_complex: torch.Tensor) -> torch.Tensor:
        _require_gpu_c64("x_freq", x_freq)
        _require_gpu_c64("weight_complex", weight_complex)
        if weight_complex.dim() != x_freq.dim() or weight_complex.shape[0] != 1 \
                or weight_complex.shape[1:] != x_freq.shape[1:]:
            raise ValueError(f"weight of shape {tuple(weight_complex.shape)} must be "
                             f"(1, {', '.join(str(s) for s in x_freq.shape[1:])})")
        x = x_freq.resolve_conj().contiguous()
        w = weight_complex.resolve_conj().contiguous()
        out = torch.empty_like(x)
        batch, inner = x.shape[0], w.numel()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().smx_cmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), batch, inner,
                                           0, _stream(x.device)))
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output: torch.Tensor) -> tuple:
        x, w = ctx.saved_tensors
        g = grad_output.resolve_conj().contiguous()
        batch, inner = x.shape[0], w.numel()
        grad_x = torch.empty_like(x)
        grad_w = torch.empty_like(w)
        with torch.cuda.device(x.device):
            s = _stream(x.device)
            _lib.check(_lib.lib().smx_cmul(g.data_ptr(), w.data_ptr(), grad_x.data_ptr(), batch,
                                           inner, 1, s))
            _lib.check(_lib.lib().smx_cmul_grad_w(x.data_ptr(), g.data_ptr(), grad_w.data_ptr(),
                                                  batch, inner, s))
        return grad_x, grad_w


class ComplexParameter(nn.Module):
    """Learnable complex tensor stored as `.real` / `.imag` Parameters (reference :85-142)."""

    def __init__(self, shape: tuple, init_mode: str = "xavier"):
        super().__init__()
        if init_mode == "xavier":
            fan = (shape[0] + shape[1]) if len(shape) == 2 else shape[0]
            bound = math.sqrt(3.0 / fan)
            self.real = nn.Parameter(torch.empty(shape).uniform_(-bound, bound))
            self.imag = nn.Parameter(torch.empty(shape).uniform_(-bound, bound))
        elif init_mode == "kaiming":
            std = math.sqrt(2.0 / shape[0])
            self.real = nn.Parameter(torch.randn(shape) * std)
            self.imag = nn.Parameter(torch.randn(shape) * std)
        elif init_mode == "uniform":
            re = torch.empty(shape).uniform_(-1, 1)
            im = torch.empty(shape).uniform_(-1, 1)
            mag = torch.sqrt(re ** 2 + im ** 2)
            self.real = nn.Parameter(re / mag)
            self.imag = nn.Parameter(im / mag)
        elif init_mode == "ones":
            self.real = nn.Parameter(torch.ones(shape))
            self.imag = nn.Parameter(torch.zeros(shape))
        else:
            raise ValueError(f"Unknown init_mode: {init_mode}")

    def forward(self) -> torch.Tensor:
        return torch.complex(self.real, self.imag)

    def magnitude(self) -> torch.Tensor:
        return torch.sqrt(self.real ** 2 + self.imag ** 2)

    def phase(self) -> torch.Tensor:
        return torch.atan2(self.imag, self.real)


class _FilterFn(Function):
    """Zero-filled spectral filter on a complex (B, T, D) tensor, reference :170-203."""

    @staticmethod
    def forward(ctx, x_freq, w_real, w_imag):
        _require_gpu_c64("x_freq", x_freq)
        for name, w in (("weight.real", w_real), ("weight.imag", w_imag)):
            _require_gpu_f32(name, w)
            if w.device != x_freq.device:
                raise RuntimeError(f"{name} is on {w.device} but x_freq is on {x_freq.device}")
            if w.dim() != 2 or w.shape[0] != x_freq.shape[2] or w.shape != w_real.shape:
                raise ValueError(f"{name} must be (num_channels={x_freq.shape[2]}, num_frequencies), "
                                 f"got {tuple(w.shape)}")
        w_real, w_imag = w_real.contiguous(), w_imag.contiguous()
        x = x_freq.contiguous()
        B, N, D = x.shape
        F = w_real.shape[1]
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().smx_wfilter_forward(
                x.data_ptr(), w_real.data_ptr(), w_imag.data_ptr(), out.data_ptr(), B, N, D, F, 0,
                _stream(x.device)))
        ctx.save_for_backward(x, w_real, w_imag)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w_real, w_imag = ctx.saved_tensors
        _require_gpu_c64("grad_output", g)
        if g.shape != x.shape or g.device != x.device:
            raise ValueError(f"grad_output must be {tuple(x.shape)} on {x.device}")
        g = g.contiguous()
        B, N, D = x.shape
        F = w_real.shape[1]
        gx = gwr = gwi = None
        with torch.cuda.device(x.device):
            s = _stream(x.device)
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _lib.check(_lib.lib().smx_wfilter_forward(
                    g.data_ptr(), w_real.data_ptr(), w_imag.data_ptr(), gx.data_ptr(), B, N, D, F, 1,
                    s))
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                gwr = torch.empty_like(w_real)
                gwi = torch.empty_like(w_imag)
                _lib.check(_lib.lib().smx_wfilter_grad_w(
                    x.data_ptr(), g.data_ptr(), gwr.data_ptr(), gwi.data_ptr(), B, N, D, F, s))
        return gx, gwr, gwi


class WirtingerSpectralFilter(nn.Module):
    """Complex-in / complex-out learnable low-pass filter (reference :145-203)."""

    def __init__(self, num_channels: int, num_frequencies: int):
        super().__init__()
        self.num_channels = num_channels
        self.num_frequencies = num_frequencies
        self.weight = ComplexParameter(shape=(num_channels, num_frequencies), init_mode="ones")

    def forward(self, x_freq: torch.Tensor) -> torch.Tensor:
        B, T, D = x_freq.shape
        assert D == self.num_channels                                   # reference :181
        return _FilterFn.apply(x_freq, self.weight.real, self.weight.imag)


def spectral_mix_with_filter(x: torch.Tensor, filt: WirtingerSpectralFilter) -> torch.Tensor:
    """`torch.fft.ifft(filt(torch.fft.fft(x, dim=1)), dim=1).real` as ONE fused pass over x and one
    over the output (identical to SpectralMixingLayer with zero bias)."""
    return spectral_mix(x, filt.weight.real, filt.weight.imag, None)
