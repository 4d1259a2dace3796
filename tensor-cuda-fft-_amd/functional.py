"""Autograd glue between torch tensors and the C ABI (include/smx.h).

PyTorch is plumbing here: device memory, streams, autograd bookkeeping.  All arithmetic of the
hot path runs in libsmx.so's HIP kernels on torch's current stream.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _under_forward_options(backward):
    """Autograd runs backward on ITS OWN thread, where the scoped plan options of the forward call
    (`with _lib.options(...)`, thread-local) are not in force: forward and backward of one node would plan
    differently -- other kernels, another workspace layout for the workspace forward hands over.  Every Function
    below notes the options its forward ran under (ctx.smx_opts) and re-applies them around its backward."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *args):
        o = getattr(ctx, "smx_opts", None)
        if o is None:
            return backward(ctx, *args)
        with _lib.options(**o):
            return backward(ctx, *args)
    return wrapped


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
# A captured hipGraph replays with the addresses it was captured with.  Workspaces handed out WHILE A STREAM IS
# CAPTURING therefore never come from _ws_cache: they are allocated inside the capture, i.e. from that graph's
# private memory pool, and live exactly as long as the graph does.  Eager workspaces are never referenced by a
# graph, so a buffer that is outgrown simply goes back to the allocator (stream-ordered, same stream).
_WS_CACHE_MAX = 16                      # (device, stream) pairs kept; least recently used goes first


def _raw_stream(dev: torch.device) -> int:
    """The hipStream_t of torch's current stream on `dev` as an integer (the raw C query: torch.cuda.current_stream
    builds a Stream object, ~8 us a call, four calls per eager fwd+bwd: host cost of a small step 0.18-0.22 -> 0.145-0.19 ms,
    A/B in one gpurun call, the spread being the shared host CPUs)."""
    return torch._C._cuda_getCurrentRawStream(dev.index)


def _workspace(dev: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    if nbytes == 0:
        return None
    if torch._C._cuda_isCurrentStreamCapturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    key = (dev.index, _raw_stream(dev))
    ws = _ws_cache.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _ws_cache[key] = ws                 # (re)inserted last = most recently used
    while len(_ws_cache) > _WS_CACHE_MAX:
        _ws_cache.pop(next(iter(_ws_cache)))
    return ws


def release_workspaces() -> None:
    """Drop every cached workspace (they return to torch's allocator once the work queued on them is done)."""
    _ws_cache.clear()


class _Memo:
    """Small bounded memo for per-shape answers of the library (plan, workspace size ...).  Keys carry
    _lib.opts_key(), so an entry is only ever found under the plan options it was computed with; the bound keeps
    a variable-length workload from growing it without limit (oldest entries go first)."""

    def __init__(self, cap: int = 512):
        self.cap = cap
        self.d: dict = {}

    def get(self, key, make):
        k = (key, _lib.opts_key())
        v = self.d.get(k)
        if v is None:
            v = self.d[k] = make()
            while len(self.d) > self.cap:
                self.d.pop(next(iter(self.d)))
        return v

    def peek(self, key):
        """the cached value or None, never computing it"""
        return self.d.get((key, _lib.opts_key()))

    def __contains__(self, key):
        return (key, _lib.opts_key()) in self.d

    def __getitem__(self, key):
        return self.d[(key, _lib.opts_key())]

    def clear(self):
        self.d.clear()


_prepared: set = set()
_prepared_epoch = [0]


def _prepare(dev: torch.device, N: int) -> None:
    """smx_prepare(N) once per (device, N): the twiddle tables are uploaded with a blocking copy, which
    must not happen inside a stream capture (the library refuses it there with a clear error).  The set is
    dropped whenever the library evicted tables (smx_tables_epoch), so it never claims more than is on the
    device, and it cannot outgrow the library's own table cache."""
    ep = int(_lib.lib().smx_tables_epoch())
    if ep != _prepared_epoch[0]:
        _prepared.clear()
        _prepared_epoch[0] = ep
    key = (dev.index, int(N))
    if key not in _prepared:
        with _on_device(dev):
            _lib.check(_lib.lib().smx_prepare(int(N)))
        _prepared.add(key)


_ws_bytes_cache = _Memo()


_slow_plan_warned: set = set()


def _note_plan(p, B: int, R: int, D: int, n_fft: int) -> None:
    """One warning per process and reason when a LARGE problem lands on a plan that is far from the streaming
    kernels: DFT products (n_fft % 256 != 0 or odd channel count: ~10x the cost of the neighbouring multiple of
    256) or band groups (more than 512 bins at a tile count the four-step path does not take: x is re-read once
    per 512 bins).  Results are the same on every plan; the reference runs any length at O(N log N)."""
    if B * R * D < (1 << 22):
        return
    if p.path == _lib.SMX_PATH_DIRECT:
        why = ("direct", f"n_fft = {n_fft} is not a multiple of 256 -- nor of 16 with at most 256 bins below the Nyquist "
               f"bin, which streams at about a third of the rate -- (or an odd channel count {D} reached the native op "
               f"directly: spectral_mix pads it): this shape runs "
               f"DFT matrix products, roughly 10x the cost of the neighbouring multiple of 256")
    elif p.groups > 1:
        why = ("groups", f"{p.k} bins at n_fft = {n_fft} (256 x {n_fft // 256} tiles) run as {p.groups} band groups, "
               f"each re-reading the input; tile counts 5..32, 36..64 step 4, 72..128 step 8, 144..256 step 16 "
               f"stream it once")
    else:
        return
    if why[0] not in _slow_plan_warned:
        _slow_plan_warned.add(why[0])
        import warnings
        warnings.warn("tensor_cuda_fft_amd: " + why[1], RuntimeWarning, stacklevel=4)


def _ws_bytes(B: int, N: int, D: int, F: int) -> int:
    """smx_workspace_bytes, memoised per (shape, plan options)."""
    def make():
        _note_plan(_lib.plan(B, N, D, F), B, N, D, N)
        return _lib.workspace_bytes(B, N, D, F)
    return _ws_bytes_cache.get((B, N, D, F), make)


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


def _dense(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Contiguous and 16-byte aligned (the kernels use 8/16-byte vector accesses): a view that starts at
    an odd element of a larger buffer is copied, everything else is passed through."""
    if t is None:
        return None
    if t.is_complex():
        t = t.resolve_conj()
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(dev: torch.device) -> int:
    return _raw_stream(dev)


def num_bins(N: int, F: int) -> int:
    return min(int(F), int(N) // 2)


class DropoutState:
    """Source of the two 64-bit words (seed, counter) a fused-dropout forward/backward pair shares.

    They are drawn from torch's generator of the device (`torch.randint` on the GPU), so the fused dropout
    follows torch's RNG discipline: `torch.manual_seed` reproduces the masks, a captured hipGraph draws
    fresh words at every replay (torch registers the generator with the graph), and activation
    checkpointing -- which saves and restores the device RNG state around the recomputed forward --
    regenerates the mask of the original forward."""

    def __init__(self, device: torch.device):
        self.device = torch.device(device)

    def next(self) -> torch.Tensor:
        return torch.randint(-(1 << 62), 1 << 62, (2,), dtype=torch.int64, device=self.device)


def _check_p(p: float) -> float:
    p = float(p)
    if not 0.0 <= p < 1.0:
        raise ValueError(f"fused dropout needs 0 <= p < 1, got {p}")
    return p


def forward_raw(x, w_re, w_im, bias, *, conj_w=False, save_spectrum=False, dropout_p=0.0, rng=None,
                pack=None, pack_ready=False, ws=None):
    """y, xk = smx_forward[_dropout](...).  x (B,N,D) contiguous f32 on GPU; returns xk (B,k,D) c64 or
    None.  rng: the int64[2] device tensor from DropoutState.next() when dropout_p > 0.  pack: (k,D)
    complex64 tensor that receives the packed filter (hand it to backward_raw to skip its packing launch);
    pack_ready=True: `pack` was filled by an earlier call with the same weights, do not pack again."""
    B, N, D = x.shape
    F = w_re.shape[1]
    k = num_bins(N, F)
    y = torch.empty_like(x)
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device) if save_spectrum else None
    _prepare(x.device, N)
    if ws is None:
        ws = _workspace(x.device, _ws_bytes(B, N, D, F))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_forward_dropout(
            x.data_ptr(), w_re.data_ptr(), w_im.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(xk),
            _ptr(ws), 0 if ws is None else ws.numel(), B, N, D, F,
            int(bool(conj_w)) | (2 if pack_ready else 0), float(dropout_p), _ptr(rng), _ptr(pack),
            _stream(x.device)))
    return y, xk


PHASE_SPECTRUM, PHASE_INVERSE, PHASE_PARAMS, PHASE_ALL = 1, 2, 4, 7     # include/smx.h
PHASE_SYNC_CLEAN = 8


def backward_raw(g, xk, w_re, w_im, *, want_x=True, want_w=True, phases=PHASE_ALL, grad_x=None,
                 flat=None, ws=None, dropout_p=0.0, rng=None, pack=None, sync_clean=False):
    """Runs smx_backward.  Returns (grad_x, flat) where flat = [gw_re | gw_im | gbias] fp32.
    `ws`: workspace of an earlier phase (a call made on another stream must not pick that stream's), or the
    workspace the forward call of the same autograd node used -- then `sync_clean=True` tells the library that
    its flag words are zero (include/smx.h, SMX_PHASE_SYNC_CLEAN) and it skips clearing them."""
    B, N, D = g.shape
    F = w_re.shape[1]
    # with dropout the direct plan stages g * mask in grad_x during the SPECTRUM phase
    if (want_x or (dropout_p > 0.0 and phases & PHASE_SPECTRUM)) and grad_x is None:
        grad_x = torch.empty_like(g)
    if want_w and flat is None:
        flat = torch.empty(2 * D * F + D, dtype=torch.float32, device=g.device)
    gw_re = gw_im = gb = None
    if want_w:
        gw_re, gw_im, gb = flat[:D * F], flat[D * F:2 * D * F], flat[2 * D * F:]
    if ws is None:
        ws = _workspace(g.device, _ws_bytes(B, N, D, F))
    if not want_x:
        phases &= ~PHASE_INVERSE
    if sync_clean:
        phases |= PHASE_SYNC_CLEAN
    _prepare(g.device, N)
    with _on_device(g.device):
        _lib.check(_lib.lib().smx_backward_dropout(
            g.data_ptr(), _ptr(xk), w_re.data_ptr(), w_im.data_ptr(), _ptr(grad_x), _ptr(gw_re),
            _ptr(gw_im), _ptr(gb), _ptr(ws), 0 if ws is None else ws.numel(), B, N, D, F, phases,
            float(dropout_p), _ptr(rng), _ptr(pack), _stream(g.device)))
    return grad_x, flat


_pack_used_cache = _Memo()


def _new_pack(x: torch.Tensor, w_re: torch.Tensor) -> Optional[torch.Tensor]:
    """(k, D) complex64 buffer for the filter in the kernels' layout (include/smx.h, filter_pack), or None
    where the library would not touch it: small problems, and one band (k <= 128: every workgroup stages its own
    slice of (D, F) through LDS)."""
    B, N, D = x.shape
    F = w_re.shape[1]
    def make():
        p = _lib.plan(B, N, D, F)
        return (p.path == _lib.SMX_PATH_DECIMATED and B * N * D >= 8 * (1 << 20)
                and not (p.bands == 1 and p.groups == 1))
    if not _pack_used_cache.get((B, N, D, F), make):
        return None
    return torch.empty((num_bins(N, F), D), dtype=torch.complex64, device=x.device)


class _SpectralMix(torch.autograd.Function):
    """y = real(ifft(pad_k(W * fft(x)[:k]))) + bias, reference fft_tensor/spectral_layers.py:88-116.

    `sync` is None or an object with `.all_reduce(flat, pre)` -> handle-with-wait(); when given,
    backward runs SMX_PHASE_SPECTRUM on the main stream, then the parameter-gradient reduction
    (SMX_PHASE_PARAMS, passed as `pre`) and the collective on a side stream while the grad_x inverse
    transform (SMX_PHASE_INVERSE) runs on the main one.
    """

    @staticmethod
    def forward(ctx, x, w_re, w_im, bias, sync, dropout_p=0.0, drop_state=None, grad_mode=True):
        ctx.smx_opts = _lib.effective_options()
        # needs_input_grad ignores torch.no_grad(); grad_mode is the caller's torch.is_grad_enabled()
        # (inside forward() it is always off), so inference does not write the spectrum or pack the filter
        needs = grad_mode and any(ctx.needs_input_grad[:4])
        rng = drop_state.next() if dropout_p > 0.0 else None
        pack = _new_pack(x, w_re) if needs else None         # packed filter, reused by backward
        B, N, D = x.shape
        # backward runs on forward's workspace: the forward launch leaves its flag words zero (SYNC_CLEAN)
        ws = _workspace(x.device, _ws_bytes(B, N, D, w_re.shape[1]))
        y, xk = forward_raw(x, w_re, w_im, bias, save_spectrum=needs, dropout_p=dropout_p, rng=rng,
                            pack=pack, ws=ws)
        ctx.ws = ws if needs else None
        ctx.sync = sync
        ctx.has_bias = bias is not None
        ctx.drop = (dropout_p, rng)
        ctx.pack = pack
        if needs:
            ctx.save_for_backward(xk, w_re, w_im)
        return y

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        xk, w_re, w_im = ctx.saved_tensors
        if g.dtype != torch.float32:
            g = g.float()
        g = _dense(g)
        D, F = w_re.shape
        want_x = ctx.needs_input_grad[0]
        want_w = any(ctx.needs_input_grad[1:4])
        sync = ctx.sync if (want_w and ctx.sync is not None and ctx.sync.active()) else None
        dkw = dict(dropout_p=ctx.drop[0], rng=ctx.drop[1], pack=ctx.pack)
        if sync is None:
            gx, flat = backward_raw(g, xk, w_re, w_im, want_x=want_x, want_w=want_w, ws=ctx.ws,
                                    sync_clean=ctx.ws is not None, **dkw)
            if not want_x:
                gx = None
        else:
            ws = ctx.ws
            kw = dict(want_w=True, ws=ws, **dkw)
            fused = getattr(sync, "mode", "overlap") == "fused" and want_x
            first = (PHASE_SPECTRUM | PHASE_INVERSE) if fused else PHASE_SPECTRUM
            gx, flat = backward_raw(g, xk, w_re, w_im, want_x=want_x, phases=first, **kw)
            handle = sync.all_reduce(flat, pre=lambda: backward_raw(
                g, xk, w_re, w_im, want_x=False, phases=PHASE_PARAMS, flat=flat, **kw))
            if want_x and not fused:
                backward_raw(g, xk, w_re, w_im, want_x=True, phases=PHASE_INVERSE, grad_x=gx, flat=flat,
                             **kw)
            handle.wait()
        gwr = gwi = gb = None
        if want_w:
            gwr = flat[:D * F].view(D, F)
            gwi = flat[D * F:2 * D * F].view(D, F)
            gb = flat[2 * D * F:] if ctx.has_bias else None
        return gx, gwr, gwi, gb, None, None, None, None


def spectral_mix(x: torch.Tensor, weight_real: torch.Tensor, weight_imag: torch.Tensor,
                 bias: Optional[torch.Tensor] = None, sync=None, dropout_p: float = 0.0,
                 drop_state: Optional[DropoutState] = None) -> torch.Tensor:
    """Functional form of SpectralMixingLayer.forward (learnable branch).  dropout_p > 0 applies the
    training-mode dropout of the reference (:118) inside the same launches, masks from `drop_state`."""
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
    dropout_p = _check_p(dropout_p)
    if dropout_p > 0.0 and drop_state is None:
        raise ValueError("dropout_p > 0 needs a DropoutState")
    B, N, D = x.shape
    if D % 2 == 1 and x.numel() >= _ODD_D_PAD_MIN and (
            N % 256 == 0 or (N % 8 == 0 and num_bins(N, weight_real.shape[1]) <= (256 if N % 16 == 0 else 128))):
        # An odd channel count cannot be read as packed float2 pairs, which alone would send the shape to the
        # O(N k) DFT products (~10x).  One zero channel more (its weights and bias zero as well) runs the streaming
        # kernels instead; pad and slice are ordinary differentiable torch ops, so every gradient comes back in
        # the caller's shapes.  The reference handles any D (spectral_layers.py:88); costs one extra copy of x.
        Fp = torch.nn.functional.pad
        y = spectral_mix(Fp(x, (0, 1)), Fp(weight_real, (0, 0, 0, 1)), Fp(weight_imag, (0, 0, 0, 1)),
                         None if bias is None else Fp(bias, (0, 1)), sync, dropout_p, drop_state)
        return y[..., :D]
    F = weight_real.shape[1]
    k = num_bins(N, F)
    if (N % 16 == 8 and D % 2 == 0 and 1 <= k <= 128 and x.numel() >= _ODD_D_PAD_MIN
            and (sync is None or not sync.active())):
        # N = 8 (odd): no sub-transform of the decimated kernels divides it, but the N-point bins ARE the even bins of
        # the 2N-point transform of the zero-padded sequence (w_N^{f n} = w_2N^{2 f n}), and 2N is a multiple of 16:
        # the sixteen-row decimation runs it with rows N .. 2N-1 never read or written.  The filter is expanded to
        # the even bins (odd bins zero; doubled, because the 2N-point inverse divides by 2N) with ordinary
        # differentiable torch ops, so the parameter gradients come back through them in the caller's (D, F) shape.
        def even_bins(w):
            z = torch.zeros_like(w[:, :k])
            return torch.stack((2.0 * w[:, :k], z), dim=2).reshape(D, 2 * k)[:, :2 * k - 1]
        y = spectral_filter(x, even_bins(weight_real), even_bins(weight_imag), bias, n_fft=2 * N, k=2 * k - 1)
        # (training-mode dropout as a separate pass here: the general entry point has no fused one)
        return torch.nn.functional.dropout(y, dropout_p, True) if dropout_p > 0.0 else y
    return _SpectralMix.apply(_dense(x), _dense(weight_real), _dense(weight_imag), _dense(bias), sync,
                              dropout_p, drop_state, torch.is_grad_enabled())


_ODD_D_PAD_MIN = 1 << 18        # below this the literal kernels of the direct plan are launch-bound anyway


def block_forward_raw(x, ln_w, ln_b, eps, w_re, w_im, bias, *, save=True, dropout_p=0.0, rng=None,
                      pack=None, ws=None):
    """y, xk, stats = smx_block_forward(...): y = x + mix(LayerNorm(x)); stats (B,N,2) = (mean, rstd)."""
    B, N, D = x.shape
    F = w_re.shape[1]
    k = num_bins(N, F)
    y = torch.empty_like(x)
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device) if save else None
    stats = torch.empty((B, N, 2), dtype=torch.float32, device=x.device)
    _prepare(x.device, N)
    if ws is None:
        ws = _workspace(x.device, _ws_bytes(B, N, D, F))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_block_forward_dropout(
            x.data_ptr(), _ptr(ln_w), _ptr(ln_b), float(eps), w_re.data_ptr(), w_im.data_ptr(),
            _ptr(bias), y.data_ptr(), _ptr(xk), stats.data_ptr(), _ptr(ws),
            0 if ws is None else ws.numel(), B, N, D, F, float(dropout_p), _ptr(rng), _ptr(pack),
            _stream(x.device)))
    return y, xk, stats


def block_backward_raw(g, x, stats, ln_w, xk, w_re, w_im, *, phases=PHASE_ALL, grad_x=None,
                       flat=None, ln_flat=None, ws=None, dropout_p=0.0, rng=None, pack=None, sync_clean=False):
    """Runs smx_block_backward.  Returns (grad_x, flat, ln_flat): flat = [gw_re | gw_im | gbias],
    ln_flat = [g_ln_w | g_ln_b]."""
    B, N, D = g.shape
    F = w_re.shape[1]
    if grad_x is None:
        grad_x = torch.empty_like(g)
    if flat is None:
        flat = torch.empty(2 * D * F + D, dtype=torch.float32, device=g.device)
    if ln_flat is None:
        ln_flat = torch.empty(2 * D, dtype=torch.float32, device=g.device)
    if ws is None:
        ws = _workspace(g.device, _ws_bytes(B, N, D, F))
    if sync_clean:
        phases |= PHASE_SYNC_CLEAN
    _prepare(g.device, N)
    with _on_device(g.device):
        _lib.check(_lib.lib().smx_block_backward_dropout(
            g.data_ptr(), x.data_ptr(), stats.data_ptr(), _ptr(ln_w), _ptr(xk), w_re.data_ptr(),
            w_im.data_ptr(), grad_x.data_ptr(), ln_flat[:D].data_ptr(), ln_flat[D:].data_ptr(),
            flat[:D * F].data_ptr(), flat[D * F:2 * D * F].data_ptr(), flat[2 * D * F:].data_ptr(),
            _ptr(ws), 0 if ws is None else ws.numel(), B, N, D, F, phases, float(dropout_p),
            _ptr(rng), _ptr(pack), _stream(g.device)))
    return grad_x, flat, ln_flat


class _SpectralBlockMix(torch.autograd.Function):
    """y = x + mix(LayerNorm(x)): first line of SpectralMLPBlock.forward with inactive dropout,
    reference fft_tensor/spectral_layers.py:185.  LayerNorm is applied inside the transform's load
    and the residual inside its store; backward is smx_backward + one LayerNorm-backward pass."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w_re, w_im, bias, sync, dropout_p=0.0, drop_state=None,
                grad_mode=True):
        ctx.smx_opts = _lib.effective_options()
        needs = grad_mode and any(ctx.needs_input_grad)
        rng = drop_state.next() if dropout_p > 0.0 else None
        pack = _new_pack(x, w_re) if needs else None
        B, N, D = x.shape
        ws = _workspace(x.device, _ws_bytes(B, N, D, w_re.shape[1]))     # shared with backward, see _SpectralMix
        y, xk, stats = block_forward_raw(x, ln_w, ln_b, eps, w_re, w_im, bias, save=needs,
                                         dropout_p=dropout_p, rng=rng, pack=pack, ws=ws)
        ctx.ws = ws if needs else None
        ctx.sync = sync
        ctx.drop = (dropout_p, rng)
        ctx.pack = pack
        ctx.flags = (ln_w is not None, ln_b is not None, bias is not None)
        if needs:
            ctx.save_for_backward(x, stats, xk, w_re, w_im,
                                  ln_w if ln_w is not None else x.new_empty(0))
        return y

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        x, stats, xk, w_re, w_im, ln_w = ctx.saved_tensors
        has_w, has_b, has_bias = ctx.flags
        if not has_w:
            ln_w = None
        if g.dtype != torch.float32:
            g = g.float()
        g = _dense(g)
        D, F = w_re.shape
        sync = ctx.sync if (ctx.sync is not None and ctx.sync.active()) else None
        dkw = dict(dropout_p=ctx.drop[0], rng=ctx.drop[1], pack=ctx.pack)
        if sync is None:
            gx, flat, lnf = block_backward_raw(g, x, stats, ln_w, xk, w_re, w_im, ws=ctx.ws,
                                               sync_clean=ctx.ws is not None, **dkw)
        else:
            args = (g, x, stats, ln_w, xk, w_re, w_im)
            ws = ctx.ws
            fused = getattr(sync, "mode", "overlap") == "fused"
            first = (PHASE_SPECTRUM | PHASE_INVERSE) if fused else PHASE_SPECTRUM
            gx, flat, lnf = block_backward_raw(*args, phases=first, ws=ws, **dkw)
            kw = dict(grad_x=gx, flat=flat, ln_flat=lnf, ws=ws, **dkw)
            handle = sync.all_reduce(flat, pre=lambda: block_backward_raw(*args, phases=PHASE_PARAMS, **kw))
            if not fused:
                block_backward_raw(*args, phases=PHASE_INVERSE, **kw)
            # norm1's weight / bias gradients come out of the LayerNorm backward at the end of the INVERSE
            # phase: a second, tiny (2 D floats) collective, so that every gradient this op returns is
            # already summed over the ranks
            handle2 = sync.all_reduce(lnf) if (has_w or has_b) else None
            handle.wait()
            if handle2 is not None:
                handle2.wait()
        return (gx, lnf[:D] if has_w else None, lnf[D:] if has_b else None, None,
                flat[:D * F].view(D, F), flat[D * F:2 * D * F].view(D, F),
                flat[2 * D * F:2 * D * F + D] if has_bias else None, None, None, None, None)


def block_supported(D: int) -> bool:
    return bool(_lib.lib().smx_block_supported(int(D)))


def spectral_block_mix(x: torch.Tensor, ln_weight: Optional[torch.Tensor],
                       ln_bias: Optional[torch.Tensor], eps: float, weight_real: torch.Tensor,
                       weight_imag: torch.Tensor, bias: Optional[torch.Tensor] = None,
                       sync=None, dropout_p: float = 0.0,
                       drop_state: Optional[DropoutState] = None) -> torch.Tensor:
    """x + dropout(spectral_mix(layer_norm(x, (D,), ln_weight, ln_bias, eps), weight_real, weight_imag,
    bias), dropout_p)."""
    _require_gpu_f32("x", x)
    for name, t in (("ln_weight", ln_weight), ("ln_bias", ln_bias), ("weight_real", weight_real),
                    ("weight_imag", weight_imag), ("bias", bias)):
        if t is not None:
            _require_gpu_f32(name, t)
    if x.dim() != 3:
        raise ValueError(f"expected x of shape (B, T, D), got {tuple(x.shape)}")
    D = x.shape[2]
    if weight_real.shape != weight_imag.shape or weight_real.dim() != 2 or weight_real.shape[0] != D:
        raise ValueError("weights must both be (D, num_filters)")
    for name, t in (("ln_weight", ln_weight), ("ln_bias", ln_bias), ("bias", bias)):
        if t is not None and tuple(t.shape) != (D,):
            raise ValueError(f"{name} must have shape ({D},)")
    if x.numel() == 0:
        return torch.empty_like(x)
    dropout_p = _check_p(dropout_p)
    if dropout_p > 0.0 and drop_state is None:
        raise ValueError("dropout_p > 0 needs a DropoutState")
    N = x.shape[1]
    if x.numel() >= _ODD_D_PAD_MIN and ((D % 2 == 1 and N % 8 == 0) or (D % 2 == 0 and N % 16 == 8)):
        # shapes that only stream through spectral_mix's own routes (one zero channel more; the even bins of twice
        # the length): the block line as the composition it is -- the native block call would run DFT products
        h = torch.nn.functional.layer_norm(x, (D,), ln_weight, ln_bias, eps)
        return x + spectral_mix(h, weight_real, weight_imag, bias, sync, dropout_p, drop_state)
    return _SpectralBlockMix.apply(_dense(x), _dense(ln_weight), _dense(ln_bias), float(eps),
                                   _dense(weight_real), _dense(weight_imag), _dense(bias), sync,
                                   dropout_p, drop_state, torch.is_grad_enabled())


def pruned_rfft(x: torch.Tensor, num_filters: int) -> torch.Tensor:
    """fft(x, dim=1)[:, :k, :] with k = min(num_filters, T//2), without forming the other bins."""
    _require_gpu_f32("x", x)
    x = _dense(x)
    B, N, D = x.shape
    k = num_bins(N, num_filters)
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device)
    if k == 0 or x.numel() == 0:
        return xk
    _prepare(x.device, N)
    ws = _workspace(x.device, _ws_bytes(B, N, D, num_filters))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_spectrum(x.data_ptr(), xk.data_ptr(), _ptr(ws),
                                           0 if ws is None else ws.numel(), B, N, D, num_filters,
                                           _stream(x.device)))
    return xk


# ---- general shapes: zero-padded rows, explicit bin count, Nyquist bin (include/smx.h, smx_*_ex) -------
def _shape(B, R, D, F, n_fft, k) -> "_lib.smx_shape":
    return _lib.smx_shape(int(B), int(R), int(D), int(F), int(n_fft), int(k))


_ws_ex_cache = _Memo()


def _ws_bytes_ex(key) -> int:
    def make():
        _note_plan(_lib.plan_ex(_shape(*key)), key[0], key[1], key[2], key[4])
        return _lib.workspace_bytes_ex(_shape(*key))
    return _ws_ex_cache.get(key, make)


def hermitian_scale(n_fft: int, k: int, device=None) -> torch.Tensor:
    """(k,) factors that turn the layer's one-sided convention -- y = real(ifft(pad(W X))) -- into
    torch.fft.irfft's: 2 on the bins that have a mirror image, 1 on DC and (even n_fft) on the Nyquist bin."""
    # a constant per (n_fft, k, device): built once on the host and kept (bounded) -- three launches and a
    # host-to-device scalar copy per call otherwise, and that copy is what made PhaseAware / ComplexRoPE steps
    # impossible to capture into a hipGraph.  Callers only read it (out-of-place products).
    dev = torch.device(device) if device is not None else torch.device("cpu")
    key = (int(n_fft), int(k), dev.type, dev.index)

    def make():
        # an ordinary tensor whatever the caller's mode: built under inference_mode it would be an inference tensor,
        # and the training step that multiplies it with a requires-grad filter later could not save it for backward
        with torch.inference_mode(False), torch.no_grad():
            c = torch.full((k,), 2.0, dtype=torch.float32)
            if k > 0:
                c[0] = 1.0
            if n_fft % 2 == 0 and k > n_fft // 2:
                c[n_fft // 2] = 1.0
            return c.to(dev)
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        hit = _herm_cache.peek(key)
        if hit is None:           # the host-to-device copy would break the capture, and the tensor would live in
            raise RuntimeError(   # that graph's private pool: build it eagerly first
                f"hermitian_scale({n_fft}, {k}) is not cached on {dev} yet and the stream is being captured: "
                f"run one eager call of this shape before the capture")
        return hit
    return _herm_cache.get(key, make)


_herm_cache = _Memo(64)


_row_scale_ok = _Memo()


def row_scale_supported(key) -> bool:
    """Can the plan of this shape apply a per-(batch row, channel) factor inside its filter stage?"""
    return _row_scale_ok.get(key, lambda: bool(_lib.lib().smx_row_scale_supported(_shape(*key))))


class _SpectralFilter(torch.autograd.Function):
    """y[:, :R] = real(ifft_n(pad_k(W * s * fft_n(zero-pad(x))[:k])))[:, :R] + bias through smx_forward_ex /
    smx_backward_ex: the fused transform for the layer's relatives -- zero-padded causal
    convolution (reference fft_lm/train_fixed_full.py:507-555), full one-sided spectra incl. Nyquist
    (spectral_enhancements.py:147-164, complex_rope.py:207-216).  s = row_scale (B, D) or None."""

    @staticmethod
    def forward(ctx, x, w_re, w_im, bias, row_scale, n_fft, k, grad_mode):
        ctx.smx_opts = _lib.effective_options()
        B, R, D = x.shape
        F = w_re.shape[1]
        key = (B, R, D, F, n_fft, k)
        needs = grad_mode and any(ctx.needs_input_grad[:5])
        y = torch.empty_like(x)
        xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device) if needs else None
        _prepare(x.device, n_fft)
        ws = _workspace(x.device, _ws_bytes_ex(key))
        sh = _shape(*key)
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_forward_ex(
                sh, x.data_ptr(), w_re.data_ptr(), w_im.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(xk),
                _ptr(ws), 0 if ws is None else ws.numel(), 0, None, _ptr(row_scale), _stream(x.device)))
        ctx.key = key
        ctx.has_bias = bias is not None
        ctx.has_scale = row_scale is not None
        if needs:
            ctx.save_for_backward(xk, w_re, w_im, row_scale if row_scale is not None else x.new_empty(0))
        return y

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        xk, w_re, w_im, row_scale = ctx.saved_tensors
        if not ctx.has_scale:
            row_scale = None
        B, R, D, F, n_fft, k = ctx.key
        if g.dtype != torch.float32:
            g = g.float()
        g = _dense(g)
        want_x = ctx.needs_input_grad[0]
        want_w = any(ctx.needs_input_grad[1:4])
        want_s = ctx.has_scale and ctx.needs_input_grad[4]
        gx = torch.empty_like(g) if want_x else None
        flat = torch.empty(2 * D * F + D, dtype=torch.float32, device=g.device) if want_w else None
        gs = torch.empty((B, D), dtype=torch.float32, device=g.device) if want_s else None
        ptrs = (None, None, None) if flat is None else \
            (flat[:D * F].data_ptr(), flat[D * F:2 * D * F].data_ptr(), flat[2 * D * F:].data_ptr())
        ws = _workspace(g.device, _ws_bytes_ex(ctx.key))
        phases = PHASE_ALL if want_x else (PHASE_SPECTRUM | PHASE_PARAMS)
        with _on_device(g.device):
            _lib.check(_lib.lib().smx_backward_ex(
                _shape(*ctx.key), g.data_ptr(), _ptr(xk), w_re.data_ptr(), w_im.data_ptr(), _ptr(gx),
                ptrs[0], ptrs[1], ptrs[2], _ptr(ws), 0 if ws is None else ws.numel(), phases, None,
                _ptr(row_scale), _ptr(gs), _stream(g.device)))
        gwr = gwi = gb = None
        if want_w:
            gwr = flat[:D * F].view(D, F)
            gwi = flat[D * F:2 * D * F].view(D, F)
            gb = flat[2 * D * F:] if ctx.has_bias else None
        return gx, gwr, gwi, gb, gs, None, None, None


def spectral_filter(x: torch.Tensor, weight_real: torch.Tensor, weight_imag: torch.Tensor,
                    bias: Optional[torch.Tensor] = None, *, n_fft: Optional[int] = None,
                    k: Optional[int] = None, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """General form of spectral_mix: x (B, rows, D) is zero-padded to n_fft (default rows), the first k
    bins (default min(F, n_fft // 2), at most n_fft // 2 + 1) are multiplied by W = weight_real + i
    weight_imag (D, F) and the real part of the inverse transform is cropped back to `rows`.  One-sided
    convention as in the layer; multiply W by hermitian_scale() for torch.fft.irfft semantics.
    row_scale (B, D): an extra real factor per (batch row, channel) on the filter -- a gate that depends on
    the batch row -- applied inside the native filter stage where the plan allows (row_scale_supported),
    as a multiply of the output otherwise; differentiable either way."""
    _require_gpu_f32("x", x)
    _require_gpu_f32("weight_real", weight_real)
    _require_gpu_f32("weight_imag", weight_imag)
    if bias is not None:
        _require_gpu_f32("bias", bias)
    if x.dim() != 3:
        raise ValueError(f"expected x of shape (B, T, D), got {tuple(x.shape)}")
    B, R, D = x.shape
    if weight_real.shape != weight_imag.shape or weight_real.dim() != 2 or weight_real.shape[0] != D:
        raise ValueError("weights must both be (D, num_filters)")
    F = weight_real.shape[1]
    n_fft = R if n_fft is None else int(n_fft)
    if n_fft < R:
        raise ValueError(f"n_fft={n_fft} is shorter than the sequence ({R})")
    k = min(F, n_fft // 2) if k is None else int(k)
    if not 0 <= k <= min(F, n_fft // 2 + 1):
        raise ValueError(f"k={k} must be in [0, min(F, n_fft // 2 + 1)] = [0, {min(F, n_fft // 2 + 1)}]")
    if row_scale is not None:
        _require_gpu_f32("row_scale", row_scale)
        if tuple(row_scale.shape) != (B, D):
            raise ValueError(f"row_scale must be (B, D) = ({B}, {D}), got {tuple(row_scale.shape)}")
        if bias is not None:
            raise ValueError("row_scale and bias together are not defined (the factor is on the filter)")
    if x.numel() == 0:
        return torch.empty_like(x)
    post = None
    if row_scale is not None and not row_scale_supported((B, R, D, F, n_fft, k)):
        post, row_scale = row_scale, None
    y = _SpectralFilter.apply(_dense(x), _dense(weight_real), _dense(weight_imag), _dense(bias),
                              _dense(row_scale), n_fft, k, torch.is_grad_enabled())
    return y if post is None else y * post.unsqueeze(1)


def _rfft_raw(x: torch.Tensor, n_fft: int, k: int, scale: float = 1.0, hermitian: bool = False) -> torch.Tensor:
    """scale * c_f * rfft(x, n_fft, dim=1)[:, :k] (include/smx.h, smx_rfft_ex); no autograd."""
    B, R, D = x.shape
    xk = torch.empty((B, k, D), dtype=torch.complex64, device=x.device)
    if k == 0 or x.numel() == 0:
        return xk
    key = (B, R, D, max(k, 1), n_fft, k)
    _prepare(x.device, n_fft)
    ws = _workspace(x.device, _ws_bytes_ex(key))
    with _on_device(x.device):
        _lib.check(_lib.lib().smx_rfft_ex(_shape(*key), x.data_ptr(), xk.data_ptr(), float(scale), int(hermitian),
                                          _ptr(ws), 0 if ws is None else ws.numel(), _stream(x.device)))
    return xk


def _irfft_raw(spec: torch.Tensor, n_fft: int, rows: int, scale: float, hermitian: bool) -> torch.Tensor:
    """scale * sum_f c_f Re(spec[:, f] e^{+2 pi i f n / n_fft}), n < rows (smx_irfft_ex); no autograd."""
    B, k, D = spec.shape
    y = torch.empty((B, rows, D), dtype=torch.float32, device=spec.device)
    if y.numel() == 0:
        return y
    key = (B, rows, D, max(k, 1), n_fft, k)
    _prepare(spec.device, n_fft)
    ws = _workspace(spec.device, _ws_bytes_ex(key))
    with _on_device(spec.device):
        _lib.check(_lib.lib().smx_irfft_ex(_shape(*key), spec.data_ptr(), y.data_ptr(), float(scale),
                                           int(hermitian), _ptr(ws), 0 if ws is None else ws.numel(),
                                           _stream(spec.device)))
    return y


def _require_gpu_c64(name: str, t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X path has no CPU implementation")
    if t.dtype != torch.complex64:
        raise TypeError(f"{name} must be complex64, got {t.dtype}")
    if t.dim() != 3:
        raise ValueError(f"{name}: expected (B, bins, D), got {tuple(t.shape)}")


def _check_bins(R: int, n_fft: int, k: int) -> None:
    if n_fft < R or not 0 <= k <= n_fft // 2 + 1:
        raise ValueError(f"need rows <= n_fft and 0 <= k <= n_fft // 2 + 1 (rows={R}, n_fft={n_fft}, k={k})")


def rfft_bins(x: torch.Tensor, k: Optional[int] = None, n_fft: Optional[int] = None) -> torch.Tensor:
    """torch.fft.rfft(x, n=n_fft, dim=1)[:, :k, :] (default: every bin, k = n_fft // 2 + 1) without forming
    the others; no autograd (functional.rfft is the differentiable form)."""
    _require_gpu_f32("x", x)
    x = _dense(x.detach())
    R = x.shape[1]
    n_fft = R if n_fft is None else int(n_fft)
    k = n_fft // 2 + 1 if k is None else int(k)
    _check_bins(R, n_fft, k)
    return _rfft_raw(x, n_fft, k)


class _RFFT(torch.autograd.Function):
    """The transform pair of the blocks that work on the spectrum between the transforms (reference
    fft_lm/frequency_native.py:316-317 and :355-356, fft_lm/bicameral.py:170-171 and :203-204).
    y_f = sum_n x_n e^{-i theta}  =>  grad_x[n] = Re sum_f grad_y[f] e^{+i theta}: a synthesis with weight one
    on every bin (torch's convention: grad of a complex tensor = dL/dRe + i dL/dIm)."""

    @staticmethod
    def forward(ctx, x, n_fft, k):
        ctx.smx_opts = _lib.effective_options()
        ctx.dims = (x.shape[1], n_fft)
        return _rfft_raw(x, n_fft, k)

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        R, n_fft = ctx.dims
        return _irfft_raw(_dense(g), n_fft, R, 1.0, False), None, None


class _IRFFT(torch.autograd.Function):
    """y[n] = (1/N) sum_f c_f Re(Y_f e^{+i theta})  =>  grad_Y[f] = (c_f / N) rfft(grad_y)[f] (zero-padded rows)."""

    @staticmethod
    def forward(ctx, spec, n_fft, rows):
        ctx.smx_opts = _lib.effective_options()
        ctx.dims = (spec.shape[1], n_fft)
        return _irfft_raw(spec, n_fft, rows, 1.0 / n_fft, True)

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        k, n_fft = ctx.dims
        return _rfft_raw(_dense(g), n_fft, k, 1.0 / n_fft, True), None, None


def rfft(x: torch.Tensor, n: Optional[int] = None, k: Optional[int] = None) -> torch.Tensor:
    """torch.fft.rfft(x, n=n, dim=1)[:, :k] of a real (B, rows, D) tensor, rows <= n (zero-padded), with autograd.
    Default k = n // 2 + 1 (every bin)."""
    _require_gpu_f32("x", x)
    R = x.shape[1]
    n = R if n is None else int(n)
    k = n // 2 + 1 if k is None else int(k)
    _check_bins(R, n, k)
    return _RFFT.apply(_dense(x), n, k)


def irfft(spec: torch.Tensor, n: Optional[int] = None, rows: Optional[int] = None) -> torch.Tensor:
    """torch.fft.irfft(spec, n=n, dim=1)[:, :rows] of a (B, k, D) complex64 spectrum (bins >= k read as zero,
    as torch pads), with autograd.  Default n = 2 (k - 1), rows = n."""
    _require_gpu_c64("spec", spec)
    k = spec.shape[1]
    n = 2 * (k - 1) if n is None else int(n)
    rows = n if rows is None else int(rows)
    if n < 1:
        raise ValueError(f"n must be positive, got {n}")
    if k > n // 2 + 1:                       # torch.fft.irfft trims a longer input
        spec, k = spec[:, :n // 2 + 1], n // 2 + 1
    _check_bins(rows, n, k)
    return _IRFFT.apply(_dense(spec), n, rows)


class _SeqFFT(torch.autograd.Function):
    """torch.fft.fft(z, dim=1) of a COMPLEX (B, N, D) tensor (reference frequency_ops.py:188-204,
    FrequencyAttention.fnet_attention).  A complex channel IS the packed pair the kernels transform: the
    tensor is handed over as real (B, N, 2 D), the one-sided spectra of the real and imaginary parts come
    back from one native pass and are recombined, Z[f] = A[f] + i B[f], Z[N-f] = conj(A[f]) + i conj(B[f]).
    Backward of an unnormalised DFT is the same transform: grad_z = conj(fft(conj(grad_Z)))."""

    @staticmethod
    def forward(ctx, z):
        ctx.smx_opts = _lib.effective_options()
        return seq_fft_raw(z)

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        return seq_fft_raw(_dense(g).conj().resolve_conj()).conj().resolve_conj()


_cfft_native = _Memo()


def seq_fft_raw(z: torch.Tensor) -> torch.Tensor:
    if not z.is_cuda:
        raise RuntimeError(f"z is on {z.device}: the MI355X path has no CPU implementation")
    if z.dtype != torch.complex64:
        raise TypeError(f"z must be complex64, got {z.dtype}")
    if z.dim() != 3:
        raise ValueError(f"expected (B, N, D), got {tuple(z.shape)}")
    B, N, D = z.shape
    if z.numel() == 0:
        return torch.empty_like(z)
    # _dense: a lazily conjugated input (seq_fft(x.conj())) is materialised -- torch.fft.fft accepts such
    # tensors, view_as_real does not -- and a misaligned view is copied
    xr = torch.view_as_real(_dense(z)).reshape(B, N, 2 * D)
    key = (B, N, 2 * D, N // 2 + 1, N, N // 2 + 1)

    def native_bytes():
        import ctypes
        nb = ctypes.c_size_t()
        rc = _lib.lib().smx_cfft_workspace_bytes(_shape(*key), ctypes.byref(nb))
        return int(nb.value) if rc == 0 else 0
    fs = _cfft_native.get(key, native_bytes)
    if fs:               # the packed spectrum goes straight out: tile spectra + one column pass
        out = torch.empty((B, N, D), dtype=torch.complex64, device=z.device)
        _prepare(z.device, N)
        ws = _workspace(z.device, fs)
        with _on_device(z.device):
            _lib.check(_lib.lib().smx_cfft_ex(_shape(*key), xr.data_ptr(), out.data_ptr(), _ptr(ws),
                                              0 if ws is None else ws.numel(), _stream(z.device)))
        return out
    half = rfft_bins(xr, N // 2 + 1, N)                          # (B, N//2+1, 2D): spectra of re / im parts
    A, Bc = half[..., 0::2], half[..., 1::2]
    out = torch.empty((B, N, D), dtype=torch.complex64, device=z.device)
    out[:, :N // 2 + 1] = A + 1j * Bc
    m = (N - 1) // 2                                             # bins 1..m have a distinct mirror image
    if m > 0:
        out[:, N - m:] = torch.flip(A[:, 1:m + 1].conj() + 1j * Bc[:, 1:m + 1].conj(), dims=(1,))
    return out


def seq_fft(z: torch.Tensor) -> torch.Tensor:
    return _SeqFFT.apply(z)


# ---- rank-one filter: fft_lm's causal FFT convolution with its own kernels (include/smx.h, smx_conv_*) -------
_conv_info = _Memo()


def _conv_plan(B: int, R: int, D: int, n_fft: int):
    """(workspace bytes, saved-spectra bytes) when smx_conv_* takes this shape, else None."""
    def make():
        sh = _shape(B, R, D, n_fft // 2 + 1, n_fft, n_fft // 2 + 1)
        if D % 2 == 0 and n_fft % 256 == 0 and _lib.lib().smx_conv_supported(sh):
            import ctypes
            a, b = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(_lib.lib().smx_conv_workspace_bytes(sh, ctypes.byref(a), ctypes.byref(b)))
            return (int(a.value), int(b.value))
        return ()
    v = _conv_info.get((B, R, D, n_fft), make)
    return v if v else None


def conv_supported(B: int, R: int, D: int, n_fft: int) -> bool:
    return _conv_plan(B, R, D, n_fft) is not None


class _RankOneConv(torch.autograd.Function):
    """y[b, n, c] = s[b, c] * irfft(rfft(zero-pad(x[b, :, c]), n_fft) * (h_re + i h_im), n_fft)[n], n < rows
    (reference fft_lm/train_fixed_full.py:507-555) through smx_conv_forward / smx_conv_backward."""

    @staticmethod
    def forward(ctx, x, h_re, h_im, scale, n_fft, grad_mode):
        ctx.smx_opts = _lib.effective_options()
        B, R, D = x.shape
        wsb, saveb = _conv_plan(B, R, D, n_fft)
        needs = grad_mode and any(ctx.needs_input_grad[:4])
        y = torch.empty_like(x)
        xs = torch.empty(saveb, dtype=torch.uint8, device=x.device) if needs else None
        _prepare(x.device, n_fft)
        ws = _workspace(x.device, wsb)
        sh = _shape(B, R, D, n_fft // 2 + 1, n_fft, n_fft // 2 + 1)
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_conv_forward(sh, x.data_ptr(), h_re.data_ptr(), h_im.data_ptr(), _ptr(scale),
                                                   y.data_ptr(), _ptr(xs), ws.data_ptr(), ws.numel(),
                                                   _stream(x.device)))
        ctx.n_fft = n_fft
        ctx.has_scale = scale is not None
        if needs:
            ctx.save_for_backward(xs, h_re, h_im, scale if scale is not None else x.new_empty(0))
        return y

    @staticmethod
    @once_differentiable
    @_under_forward_options
    def backward(ctx, g):
        xs, h_re, h_im, scale = ctx.saved_tensors
        if not ctx.has_scale:
            scale = None
        if g.dtype != torch.float32:
            g = g.float()
        g = _dense(g)
        B, R, D = g.shape
        n = ctx.n_fft
        wsb, _ = _conv_plan(B, R, D, n)
        gx = torch.empty_like(g)
        fb = n // 2 + 1
        want_h = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gh = torch.empty((2, fb), dtype=torch.float32, device=g.device) if want_h else None
        gs = torch.empty((B, D), dtype=torch.float32, device=g.device) if ctx.has_scale else None
        ws = _workspace(g.device, wsb)
        sh = _shape(B, R, D, n // 2 + 1, n, n // 2 + 1)
        with _on_device(g.device):
            _lib.check(_lib.lib().smx_conv_backward(
                sh, g.data_ptr(), xs.data_ptr(), h_re.data_ptr(), h_im.data_ptr(), _ptr(scale), gx.data_ptr(),
                None if gh is None else gh[0].data_ptr(), None if gh is None else gh[1].data_ptr(), _ptr(gs),
                ws.data_ptr(), ws.numel(), _stream(g.device)))
        return gx, None if gh is None else gh[0], None if gh is None else gh[1], gs, None, None


class _CausalDwConv3(torch.autograd.Function):
    """y[b, t, c] = scale[b, c] (bias[c] + w[c,0] x[t-2] + w[c,1] x[t-1] + w[c,2] x[t] [t <= T-2]) on (B, T, C): the time
    path of BicameralBlock (reference fft_lm/bicameral.py:214-227) through smx_dwconv3_forward / _backward."""

    @staticmethod
    def forward(ctx, x, w, bias, scale):
        B, T, C = x.shape
        y = torch.empty_like(x)
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_dwconv3_forward(x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(scale), y.data_ptr(),
                                                      B, T, C, _stream(x.device)))
        ctx.has = (bias is not None, scale is not None)
        ctx.save_for_backward(x, w, bias if bias is not None else x.new_empty(0),
                              scale if scale is not None else x.new_empty(0))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        import ctypes
        x, w, bias, scale = ctx.saved_tensors
        bias = bias if ctx.has[0] else None
        scale = scale if ctx.has[1] else None
        g = _dense(g.float() if g.dtype != torch.float32 else g)
        B, T, C = x.shape
        nb = ctypes.c_size_t()
        _lib.check(_lib.lib().smx_dwconv3_workspace_bytes(B, T, C, ctypes.byref(nb)))
        ws = _workspace(x.device, int(nb.value))
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gb = torch.empty_like(bias) if bias is not None and ctx.needs_input_grad[2] else None
        gs = torch.empty_like(scale) if scale is not None and ctx.needs_input_grad[3] else None
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_dwconv3_backward(g.data_ptr(), x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(scale),
                                                       _ptr(gx), _ptr(gw), _ptr(gb), _ptr(gs), ws.data_ptr(), ws.numel(),
                                                       B, T, C, _stream(x.device)))
        return gx, gw, gb, gs


def causal_dwconv3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                   scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The time path of fft_lm's BicameralBlock (reference fft_lm/bicameral.py:214-227) on the block's own (B, T, C)
    layout: shift right by one and drop the last position, depthwise `nn.Conv1d(C, C, 3, padding=1, groups=C)` with
    `weight` (C, 1, 3) or (C, 3) and `bias` (C), times `scale` (B, C) (the time gate) -- no transposes, one launch;
    differentiable in x, weight, bias and scale (fixed-order sums)."""
    _require_gpu_f32("x", x)
    _require_gpu_f32("weight", weight)
    if x.dim() != 3:
        raise ValueError(f"expected (B, T, C), got {tuple(x.shape)}")
    B, T, C = x.shape
    if weight.numel() != 3 * C or weight.shape[0] != C:
        raise ValueError(f"weight must be (C, 1, 3) or (C, 3) with C = {C}, got {tuple(weight.shape)}")
    if bias is not None:
        _require_gpu_f32("bias", bias)
        if tuple(bias.shape) != (C,):
            raise ValueError(f"bias must be (C,) = ({C},)")
    if scale is not None:
        _require_gpu_f32("scale", scale)
        if tuple(scale.shape) != (B, C):
            raise ValueError(f"scale must be (B, C) = ({B}, {C})")
    if x.numel() == 0:
        return torch.empty_like(x)
    return _CausalDwConv3.apply(_dense(x), _dense(weight.reshape(C, 3)), _dense(bias), _dense(scale))


class _SpectralLN(torch.autograd.Function):
    """SpectralLayerNorm (reference fft_lm/frequency_native.py:203-239) through smx_spectral_ln_forward / _backward:
    z (B, F, C) complex64, gamma / beta (F, C).  planar: the result as (2, B, F, C) float32 planes (real, imaginary)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, planar):
        B, Fq, C = z.shape
        out = torch.empty((2, B, Fq, C), dtype=torch.float32, device=z.device) if planar else torch.empty_like(z)
        with _on_device(z.device):
            _lib.check(_lib.lib().smx_spectral_ln_forward(
                torch.view_as_real(z).data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                (out if planar else torch.view_as_real(out)).data_ptr(), int(planar), B, Fq, C, _stream(z.device)))
        ctx.eps, ctx.planar = float(eps), bool(planar)
        ctx.save_for_backward(z, gamma, beta)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        z, gamma, beta = ctx.saved_tensors
        B, Fq, C = z.shape
        g = _dense(g.float() if ctx.planar and g.dtype != torch.float32 else g) if ctx.planar else _dense(g.to(torch.complex64))
        gz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(gamma) if ctx.needs_input_grad[1] else None
        gb = torch.empty_like(beta) if ctx.needs_input_grad[2] else None
        with _on_device(z.device):
            _lib.check(_lib.lib().smx_spectral_ln_backward(
                (g if ctx.planar else torch.view_as_real(g)).data_ptr(), torch.view_as_real(z).data_ptr(),
                gamma.data_ptr(), beta.data_ptr(), ctx.eps, None if gz is None else torch.view_as_real(gz).data_ptr(),
                _ptr(gg), _ptr(gb), int(ctx.planar), B, Fq, C, _stream(z.device)))
        return gz, gg, gb, None, None


class _PlanarCmul(torch.autograd.Function):
    """out = h (f_re + i f_im)[f, c] on (2, B, F, C) planes: PhaseShift between SpectralFFN's Linear layers (reference
    fft_lm/frequency_native.py:62-77, :175) through smx_planar_cmul_*."""

    @staticmethod
    def forward(ctx, h, f_re, f_im):
        _, B, Fq, C = h.shape
        out = torch.empty_like(h)
        with _on_device(h.device):
            _lib.check(_lib.lib().smx_planar_cmul_forward(h.data_ptr(), f_re.data_ptr(), f_im.data_ptr(), out.data_ptr(),
                                                          B, Fq, C, _stream(h.device)))
        ctx.save_for_backward(h, f_re, f_im)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        h, f_re, f_im = ctx.saved_tensors
        _, B, Fq, C = h.shape
        g = _dense(g.float() if g.dtype != torch.float32 else g)
        gh = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(f_re) if ctx.needs_input_grad[1] else None
        gi = torch.empty_like(f_im) if ctx.needs_input_grad[2] else None
        with _on_device(h.device):
            _lib.check(_lib.lib().smx_planar_cmul_backward(g.data_ptr(), h.data_ptr(), f_re.data_ptr(), f_im.data_ptr(),
                                                           _ptr(gh), _ptr(gr), _ptr(gi), B, Fq, C, _stream(h.device)))
        return gh, gr, gi


class _AddPlanar(torch.autograd.Function):
    """y = a + (p[0] + i p[1]): a, y complex64 (...), p float32 (2, ...) -- the residual around SpectralFFN (reference
    fft_lm/frequency_native.py:355-356) with the planar result of its second Linear folded in."""

    @staticmethod
    def forward(ctx, a, p):
        y = torch.empty_like(a)
        with _on_device(a.device):
            _lib.check(_lib.lib().smx_planar_add(torch.view_as_real(a).data_ptr(), p.data_ptr(),
                                                 torch.view_as_real(y).data_ptr(), a.numel(), _stream(a.device)))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _dense(g.to(torch.complex64))
        gp = None
        if ctx.needs_input_grad[1]:
            gp = torch.empty((2,) + tuple(g.shape), dtype=torch.float32, device=g.device)
            with _on_device(g.device):
                _lib.check(_lib.lib().smx_planar_split(torch.view_as_real(g).data_ptr(), gp.data_ptr(), g.numel(),
                                                       _stream(g.device)))
        return (g if ctx.needs_input_grad[0] else None), gp


class _SpectralGate(torch.autograd.Function):
    """y = ((((x a[f]) u[c]) p[f]) q[b,c]) m[f] on a (B, F, C) complex64 spectrum: the chain between the two transforms of
    fft_lm's twin blocks (reference fft_lm/frequency_native.py:95, :338, :351; fft_lm/bicameral.py:179-186) through
    smx_spectral_gate_*: one launch forward, one backward (+ the tiny products that finish the parameter gradients)."""

    @staticmethod
    def forward(ctx, x, a, u, p, q, m, reference_gain_grad):
        B, Fq, C = x.shape
        y = torch.empty_like(x)
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_spectral_gate_forward(x.data_ptr(), a.data_ptr(), _ptr(u), _ptr(p), _ptr(q), _ptr(m),
                                                            y.data_ptr(), B, Fq, C, _stream(x.device)))
        ctx.has = tuple(t is not None for t in (u, p, q, m))
        ctx.reference_gain_grad = bool(reference_gain_grad)
        e = x.new_empty(0, dtype=torch.float32)
        ctx.save_for_backward(x, a, *(t if t is not None else e for t in (u, p, q, m)))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        import ctypes
        x, a, u, p, q, m = ctx.saved_tensors
        u, p, q, m = (t if h else None for t, h in zip((u, p, q, m), ctx.has))
        B, Fq, C = x.shape
        g = _dense(g.to(torch.complex64))
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        want_u = u is not None and need[2]
        s1 = torch.empty(Fq, dtype=torch.complex64, device=x.device) if need[1] or (p is not None and need[3]) else None
        plain = want_u and ctx.reference_gain_grad
        rc = (torch.empty(B, C, dtype=torch.float32, device=x.device)
              if (q is not None and need[4]) or (want_u and not plain) else None)
        rp = torch.empty(B, C, dtype=torch.float32, device=x.device) if plain else None
        nb = ctypes.c_size_t()
        _lib.check(_lib.lib().smx_spectral_gate_workspace_bytes(B, Fq, C, ctypes.byref(nb)))
        ws = _workspace(x.device, int(nb.value))
        with _on_device(x.device):
            _lib.check(_lib.lib().smx_spectral_gate_backward(g.data_ptr(), x.data_ptr(), a.data_ptr(), _ptr(u), _ptr(p),
                                                             _ptr(q), _ptr(m), _ptr(gx), _ptr(s1), _ptr(rc), _ptr(rp),
                                                             ws.data_ptr(), ws.numel(), B, Fq, C, _stream(x.device)))
        ga = gu = gp = gq = None
        if need[1]:
            ga = s1 if p is None and m is None else s1 * (p if m is None else m if p is None else p * m)      # :111
        if p is not None and need[3]:
            gp = (a.conj() * s1).real
            if m is not None:
                gp = gp * m
        if q is not None and need[4]:
            gq = rc if u is None else rc * u
        if want_u:
            r = rp if plain else rc                       # :115 sums Re(grad x k) -- no conjugate -- for the gain
            gu = (r if q is None else r * q).sum(dim=0)
        return gx, ga, gu, gp, gq, None, None


def spectral_gate(x: torch.Tensor, a: torch.Tensor, u: Optional[torch.Tensor] = None, p: Optional[torch.Tensor] = None,
                  q: Optional[torch.Tensor] = None, m: Optional[torch.Tensor] = None,
                  reference_gain_grad: bool = False) -> torch.Tensor:
    """`((((x * a[f]) * u[c]) * p[f]) * q[b,c]) * m[f]` for a complex64 spectrum x (B, F, C): a (F) complex64 -- the kernel
    spectrum --, u (C) the per-channel gain, p (F) the frequency gate, q (B, C) the context gate, m (F) the cutoff mask (a
    constant), each optional.  Differentiable in x, a, u, p and q.  `reference_gain_grad=True` gives u the gradient
    `FrequencyConvFunc.backward` writes by hand (reference fft_lm/frequency_native.py:115: `Re(grad * x * k)` summed, no
    conjugate) instead of the autograd one.  The channel count must be even."""
    if x.dtype != torch.complex64 or not x.is_cuda or x.dim() != 3:
        raise TypeError("x must be a (B, F, C) complex64 tensor on a ROCm device")
    B, Fq, C = x.shape
    if a.dtype != torch.complex64 or tuple(a.shape) != (Fq,):
        raise ValueError(f"a must be complex64 (F,) = ({Fq},), got {a.dtype} {tuple(a.shape)}")
    for name, t, shape in (("u", u, (C,)), ("p", p, (Fq,)), ("q", q, (B, C)), ("m", m, (Fq,))):
        if t is not None:
            _require_gpu_f32(name, t)
            if tuple(t.shape) != shape:
                raise ValueError(f"{name} must be {shape}, got {tuple(t.shape)}")
    if C % 2:
        raise ValueError(f"spectral_gate takes an even channel count, got {C}")
    if x.numel() == 0:
        return x.clone()
    d = lambda t: None if t is None else _dense(t)
    return _SpectralGate.apply(_dense(x), _dense(a), d(u), d(p), d(q), d(m.detach() if m is not None else None),
                               reference_gain_grad)


_MIX_WS_BYTES = None


class _MixPaths(torch.autograd.Function):
    """out = r + w[0] a + w[1] b + c3 c: BicameralBlock's fusion line and residual (reference fft_lm/bicameral.py
    :237-268) through smx_mix_*: one launch each way, the two scalar gradients summed in a fixed order."""

    @staticmethod
    def forward(ctx, r, a, b, c, w, c3):
        out = torch.empty_like(r)
        with _on_device(r.device):
            _lib.check(_lib.lib().smx_mix_forward(r.data_ptr(), a.data_ptr(), b.data_ptr(), _ptr(c), w.data_ptr(), float(c3),
                                                  out.data_ptr(), r.numel(), _stream(r.device)))
        ctx.c3 = float(c3)
        ctx.has_c = c is not None
        ctx.save_for_backward(a, b, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        import ctypes
        a, b, w = ctx.saved_tensors
        g = _dense(g.float() if g.dtype != torch.float32 else g)
        need = ctx.needs_input_grad
        ga = torch.empty_like(a) if need[1] else None
        gb = torch.empty_like(b) if need[2] else None
        gc = torch.empty_like(a) if ctx.has_c and need[3] else None
        gw = torch.empty_like(w) if need[4] else None
        global _MIX_WS_BYTES
        if _MIX_WS_BYTES is None:                        # a constant of the library
            nb = ctypes.c_size_t()
            _lib.check(_lib.lib().smx_mix_workspace_bytes(ctypes.byref(nb)))
            _MIX_WS_BYTES = int(nb.value)
        ws = _workspace(a.device, _MIX_WS_BYTES)
        with _on_device(a.device):
            _lib.check(_lib.lib().smx_mix_backward(g.data_ptr(), a.data_ptr(), b.data_ptr(), w.data_ptr(), ctx.c3, _ptr(ga),
                                                   _ptr(gb), _ptr(gc), _ptr(gw), ws.data_ptr(), ws.numel(), a.numel(),
                                                   _stream(a.device)))
        return (g if need[0] else None), ga, gb, gc, gw, None


def mix_paths(r: torch.Tensor, a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor], w: torch.Tensor,
              c3: float = 0.1) -> torch.Tensor:
    """`r + w[0] * a + w[1] * b + c3 * c` for float32 tensors of one shape (element count a multiple of 4) and two learned
    scalars `w` (2,) in device memory; differentiable in r, a, b, c and w."""
    for name, t in (("r", r), ("a", a), ("b", b), ("w", w)) + ((("c", c),) if c is not None else ()):
        _require_gpu_f32(name, t)
    if a.shape != r.shape or b.shape != r.shape or (c is not None and c.shape != r.shape) or tuple(w.shape) != (2,):
        raise ValueError("mix_paths: r, a, b, c must share one shape and w must be (2,)")
    if r.numel() == 0 or r.numel() % 4:
        raise ValueError(f"mix_paths takes a positive element count that is a multiple of 4, got {r.numel()}")
    return _MixPaths.apply(_dense(r), _dense(a), _dense(b), _dense(c), _dense(w), c3)


def planar_cmul(h: torch.Tensor, f_re: torch.Tensor, f_im: torch.Tensor) -> torch.Tensor:
    """(2, B, F, C) planes times the complex factor (f_re + i f_im)[f, c]."""
    _require_gpu_f32("h", h)
    _require_gpu_f32("f_re", f_re)
    _require_gpu_f32("f_im", f_im)
    if h.dim() != 4 or h.shape[0] != 2 or tuple(f_re.shape) != tuple(h.shape[2:]) or f_re.shape != f_im.shape:
        raise ValueError(f"expected h (2, B, F, C) and factors (F, C), got {tuple(h.shape)}, {tuple(f_re.shape)}")
    return _PlanarCmul.apply(_dense(h), _dense(f_re), _dense(f_im))


def add_planar(a: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """a + (p[0] + i p[1]) for a complex64 `a` and float32 planes p (2, *a.shape)."""
    if a.dtype != torch.complex64 or not a.is_cuda:
        raise TypeError("a must be a complex64 tensor on a ROCm device")
    _require_gpu_f32("p", p)
    if tuple(p.shape) != (2,) + tuple(a.shape):
        raise ValueError(f"p must be (2, {tuple(a.shape)}), got {tuple(p.shape)}")
    return _AddPlanar.apply(_dense(a), _dense(p))


def spectral_layer_norm_supported(C: int) -> bool:
    return bool(_lib.lib().smx_spectral_ln_supported(int(C)))


def spectral_layer_norm(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                        planar: bool = False) -> torch.Tensor:
    """Magnitudes of z (B, F, C) complex64 normalised across the channels per (batch row, bin), scaled by gamma (F, C),
    shifted by beta (F, C), phases kept (reference fft_lm/frequency_native.py:203-239); differentiable in all three.
    planar: return (2, B, F, C) float32 planes (real, imaginary) instead of a complex tensor."""
    if not z.is_cuda:
        raise RuntimeError(f"z is on {z.device}: the MI355X path has no CPU implementation")
    if z.dtype != torch.complex64 or z.dim() != 3:
        raise TypeError(f"z must be a (B, F, C) complex64 tensor, got {z.dtype} {tuple(z.shape)}")
    _require_gpu_f32("gamma", gamma)
    _require_gpu_f32("beta", beta)
    B, Fq, C = z.shape
    if tuple(gamma.shape) != (Fq, C) or tuple(beta.shape) != (Fq, C):
        raise ValueError(f"gamma and beta must be (F, C) = ({Fq}, {C})")
    if z.numel() == 0:
        return torch.empty((2,) + tuple(z.shape), device=z.device) if planar else torch.empty_like(z)
    return _SpectralLN.apply(_dense(z), _dense(gamma), _dense(beta), float(eps), bool(planar))


class _PhaseFilter(torch.autograd.Function):
    """(w_re, w_im)[d, f] = c_f m[d] (cos p[d], sin p[d]), f < k: PhaseAwareSpectralMixing's filter (reference
    fft_tensor/spectral_enhancements.py:147-164) in one native launch, its backward in one more."""

    @staticmethod
    def forward(ctx, m, p, k, n_fft):
        D = m.numel()
        w = torch.empty((2, D, k), dtype=torch.float32, device=m.device)
        with _on_device(m.device):
            _lib.check(_lib.lib().smx_phase_filter(m.data_ptr(), p.data_ptr(), D, k, n_fft, w[0].data_ptr(),
                                                   w[1].data_ptr(), _stream(m.device)))
        ctx.k, ctx.n_fft = k, n_fft
        ctx.save_for_backward(m, p)
        return w[0], w[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_re, g_im):
        m, p = ctx.saved_tensors
        D, k = m.numel(), ctx.k
        g_re = torch.zeros((D, k), device=m.device) if g_re is None else _dense(g_re.float())
        g_im = torch.zeros((D, k), device=m.device) if g_im is None else _dense(g_im.float())
        gm = torch.empty_like(m) if ctx.needs_input_grad[0] else None
        gp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        if gm is not None or gp is not None:
            with _on_device(m.device):
                _lib.check(_lib.lib().smx_phase_filter_backward(m.data_ptr(), p.data_ptr(), g_re.data_ptr(),
                                                                g_im.data_ptr(), D, k, ctx.n_fft, k, _ptr(gm), _ptr(gp),
                                                                _stream(m.device)))
        return gm, gp, None, None


def phase_filter(m: torch.Tensor, p: torch.Tensor, k: int, n_fft: int):
    """(w_re, w_im), each (D, k): c_f m[d] exp(i p[d]) with torch.fft.irfft's Hermitian weights c_f."""
    _require_gpu_f32("magnitude", m)
    _require_gpu_f32("phase", p)
    if m.dim() != 1 or m.shape != p.shape:
        raise ValueError("magnitude and phase must be 1-D tensors of the same length")
    return _PhaseFilter.apply(_dense(m), _dense(p), int(k), int(n_fft))


class _ConvResponse(torch.autograd.Function):
    """H[f] = rfft(zero-pad(kernel), n_fft)[f] * sigmoid(gate_logits[f]) * mask[f] in one native launch
    (reference fft_lm/train_fixed_full.py:511-513, :529, :540-551) -> (Re H, Im H); backward in one more."""

    @staticmethod
    def forward(ctx, kernel, gate_logits, mask, n_fft):
        fb = n_fft // 2 + 1
        h = torch.empty((2, fb), dtype=torch.float32, device=kernel.device)
        _prepare(kernel.device, n_fft)
        with _on_device(kernel.device):
            _lib.check(_lib.lib().smx_conv_response(n_fft, kernel.numel(), kernel.data_ptr(), _ptr(gate_logits),
                                                    _ptr(mask), h[0].data_ptr(), h[1].data_ptr(),
                                                    _stream(kernel.device)))
        ctx.n_fft = n_fft
        ctx.has = (gate_logits is not None, mask is not None)
        e = kernel.new_empty(0)
        ctx.save_for_backward(kernel, gate_logits if gate_logits is not None else e, mask if mask is not None else e)
        return h[0], h[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_re, g_im):
        kernel, logits, mask = ctx.saved_tensors
        logits = logits if ctx.has[0] else None
        mask = mask if ctx.has[1] else None
        n = ctx.n_fft
        fb = n // 2 + 1
        g_re = torch.zeros(fb, device=kernel.device) if g_re is None else _dense(g_re.float())
        g_im = torch.zeros(fb, device=kernel.device) if g_im is None else _dense(g_im.float())
        gk = torch.empty_like(kernel) if ctx.needs_input_grad[0] else None
        gl = torch.empty_like(logits) if (logits is not None and ctx.needs_input_grad[1]) else None
        if gk is not None or gl is not None:
            with _on_device(kernel.device):
                _lib.check(_lib.lib().smx_conv_response_backward(
                    n, kernel.numel(), 0 if logits is None else logits.numel(), kernel.data_ptr(), _ptr(logits),
                    _ptr(mask), g_re.data_ptr(), g_im.data_ptr(), _ptr(gk), _ptr(gl), _stream(kernel.device)))
        return gk, gl, None, None


def conv_response(kernel: torch.Tensor, gate_logits: Optional[torch.Tensor], mask: Optional[torch.Tensor],
                  n_fft: int):
    """(Re H, Im H) of the causal kernel's n_fft-point response times sigmoid(gate_logits) and mask (each may be
    None); gate_logits may be longer than n_fft // 2 + 1 (the reference sizes it for its longest sequence)."""
    _require_gpu_f32("kernel", kernel)
    fb = n_fft // 2 + 1
    if gate_logits is not None:
        _require_gpu_f32("gate_logits", gate_logits)
        if gate_logits.dim() != 1 or gate_logits.numel() < fb:
            raise ValueError(f"gate_logits needs at least n_fft // 2 + 1 = {fb} entries")
    if mask is not None:
        _require_gpu_f32("mask", mask)
        if tuple(mask.shape) != (fb,):
            raise ValueError(f"mask must have n_fft // 2 + 1 = {fb} entries")
    if kernel.dim() != 1 or kernel.numel() > n_fft:
        raise ValueError("kernel must be 1-D with at most n_fft taps")
    return _ConvResponse.apply(_dense(kernel), _dense(gate_logits), _dense(mask), int(n_fft))


def rank_one_conv(x: torch.Tensor, h_re: torch.Tensor, h_im: torch.Tensor,
                  scale: Optional[torch.Tensor], n_fft: int) -> torch.Tensor:
    """Causal / circular convolution of every (batch, channel) column of x (B, rows, D) with the real kernel
    whose one-sided spectrum is h_re + i h_im (n_fft // 2 + 1), times scale[b, c]; needs
    conv_supported(B, rows, D, n_fft)."""
    _require_gpu_f32("x", x)
    _require_gpu_f32("h_re", h_re)
    _require_gpu_f32("h_im", h_im)
    B, R, D = x.shape
    fb = n_fft // 2 + 1
    if tuple(h_re.shape) != (fb,) or tuple(h_im.shape) != (fb,):
        raise ValueError(f"h_re / h_im must have n_fft // 2 + 1 = {fb} entries")
    if scale is not None:
        _require_gpu_f32("scale", scale)
        if tuple(scale.shape) != (B, D):
            raise ValueError(f"scale must be (B, D) = ({B}, {D})")
    if not conv_supported(B, R, D, n_fft):
        raise ValueError(f"rank_one_conv does not take (B={B}, rows={R}, D={D}, n_fft={n_fft}); use spectral_filter")
    return _RankOneConv.apply(_dense(x), _dense(h_re), _dense(h_im), _dense(scale), int(n_fft),
                              torch.is_grad_enabled())
