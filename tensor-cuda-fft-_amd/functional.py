"""Autograd glue between torch tensors and the C ABI (include/smx.h).

PyTorch is plumbing here: device memory, streams, autograd bookkeeping.  All arithmetic of the
hot path runs in libsmx.so's HIP kernels on torch's current stream.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _require_gpu_f32(name: str, t: torch.Tensor) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the MI355X spectral-mixing path has no CPU implementation "
            f"(move the module and its input to a ROCm device)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype} (the kernels compute in fp32)")


# One workspace per (device, stream): calls on the same stream are ordered, so reuse is safe.
_ws_cache: dict = {}


def _workspace(dev: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    if nbytes == 0:
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


_ws_bytes_cache: dict = {}


def _ws_bytes(B: int, N: int, D: int, F: int) -> int:
    """smx_workspace_bytes, memoised per shape (the tuning options are process-wide and fixed)."""
    key = (B, N, D, F)
    v = _ws_bytes_cache.get(key)
    if v is None:
        v = _ws_bytes_cache[key] = _lib.workspace_bytes(B, N, D, F)
    return v


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (it is the slow part of a call)."""

    def __init__(self, dev: torch.device):
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def num_bins(N: int, F: int) -> int:
    return min(int(F), int(N) // 2)


def forward_raw(x, w_re, w_im, bias, *, conj_w=False, save_spectrum=False):
    """y, xk = smx_forward(...).  x (B,N,D) contiguous f32 on GPU; returns xk (B,k,D) c64 or None."""
    B, N, D = x.shape
    F = w_re.shape[1]
    k = num_bins(N, F)
    y = torch.empty_like(x)
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device) if save_spectrum else None
    ws = _workspace(x.device, _ws_bytes(B, N, D, F))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_forward(
            x.data_ptr(), w_re.data_ptr(), w_im.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(xk),
            _ptr(ws), 0 if ws is None else ws.numel(), B, N, D, F, int(conj_w), _stream(x.device)))
    return y, xk


def backward_raw(g, xk, w_re, w_im, *, want_x=True, want_w=True, phases=3, grad_x=None, flat=None):
    """Runs smx_backward.  Returns (grad_x, flat) where flat = [gw_re | gw_im | gbias] fp32."""
    B, N, D = g.shape
    F = w_re.shape[1]
    if want_x and grad_x is None:
        grad_x = torch.empty_like(g)
    if want_w and flat is None:
        flat = torch.empty(2 * D * F + D, dtype=torch.float32, device=g.device)
    gw_re = gw_im = gb = None
    if want_w:
        gw_re, gw_im, gb = flat[:D * F], flat[D * F:2 * D * F], flat[2 * D * F:]
    ws = _workspace(g.device, _ws_bytes(B, N, D, F))
    if not want_x:
        phases &= 1
    with _on_device(g.device):
        _lib.check(_lib.lib().smx_backward(
            g.data_ptr(), _ptr(xk), w_re.data_ptr(), w_im.data_ptr(), _ptr(grad_x), _ptr(gw_re),
            _ptr(gw_im), _ptr(gb), _ptr(ws), 0 if ws is None else ws.numel(), B, N, D, F, phases,
            _stream(g.device)))
    return grad_x, flat


class _SpectralMix(torch.autograd.Function):
    """y = real(ifft(pad_k(W * fft(x)[:k]))) + bias, reference fft_tensor/spectral_layers.py:88-116.

    `sync` is None or an object with `.all_reduce(flat)` -> handle-with-wait(); when given, the
    parameter gradients are produced first (smx_backward phase 1), handed to the collective on
    a side stream, and the grad_x inverse transform (phase 2) runs underneath it.
    """

    @staticmethod
    def forward(ctx, x, w_re, w_im, bias, sync):
        needs = any(ctx.needs_input_grad[:4])
        y, xk = forward_raw(x, w_re, w_im, bias, save_spectrum=needs)
        ctx.sync = sync
        ctx.has_bias = bias is not None
        if needs:
            ctx.save_for_backward(xk, w_re, w_im)
        return y

    @staticmethod
    def backward(ctx, g):
        xk, w_re, w_im = ctx.saved_tensors
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        D, F = w_re.shape
        want_x = ctx.needs_input_grad[0]
        want_w = any(ctx.needs_input_grad[1:4])
        sync = ctx.sync if (want_w and ctx.sync is not None and ctx.sync.active()) else None
        if sync is None:
            gx, flat = backward_raw(g, xk, w_re, w_im, want_x=want_x, want_w=want_w)
        else:
            gx, flat = backward_raw(g, xk, w_re, w_im, want_x=want_x, want_w=True, phases=1)
            handle = sync.all_reduce(flat)               # side stream; overlaps the inverse below
            if want_x:
                backward_raw(g, xk, w_re, w_im, want_x=True, want_w=True, phases=2, grad_x=gx,
                             flat=flat)
            handle.wait()
        gwr = gwi = gb = None
        if want_w:
            gwr = flat[:D * F].view(D, F)
            gwi = flat[D * F:2 * D * F].view(D, F)
            gb = flat[2 * D * F:] if ctx.has_bias else None
        return gx, gwr, gwi, gb, None


def spectral_mix(x: torch.Tensor, weight_real: torch.Tensor, weight_imag: torch.Tensor,
                 bias: Optional[torch.Tensor] = None, sync=None) -> torch.Tensor:
    """Functional form of SpectralMixingLayer.forward (learnable branch, dropout excluded)."""
    _require_gpu_f32("x", x)
    _require_gpu_f32("weight_real", weight_real)
    _require_gpu_f32("weight_imag", weight_imag)
    if bias is not None:
        _require_gpu_f32("bias", bias)
    if x.dim() != 3:
        raise ValueError(f"expected x of shape (B, T, D), got {tuple(x.shape)}")
    if weight_real.shape != weight_imag.shape or weight_real.dim() != 2 \
            or weight_real.shape[0] != x.shape[2]:
        raise ValueError("weights must both be (D, num_filters)")
    if x.numel() == 0:
        return torch.empty_like(x)
    return _SpectralMix.apply(x.contiguous(), weight_real.contiguous(), weight_imag.contiguous(),
                              None if bias is None else bias.contiguous(), sync)


def pruned_rfft(x: torch.Tensor, num_filters: int) -> torch.Tensor:
    """fft(x, dim=1)[:, :k, :] with k = min(num_filters, T//2), without forming the other bins."""
    _require_gpu_f32("x", x)
    x = x.contiguous()
    B, N, D = x.shape
    k = num_bins(N, num_filters)
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device)
    if k == 0 or x.numel() == 0:
        return xk
    ws = _workspace(x.device, _ws_bytes(B, N, D, num_filters))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_spectrum(x.data_ptr(), xk.data_ptr(), _ptr(ws),
                                           0 if ws is None else ws.numel(), B, N, D, num_filters,
                                           _stream(x.device)))
    return xk
