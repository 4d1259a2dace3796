"""Batch-sharded data parallelism for the spectral-mixing layer (one process per GPU).

The reference has no distributed code at all; this is the MI355X-side addition
BASELINE config 4 asks for.  Every (b, d) column is an independent transform, so forward and
grad_x need no communication.  The only coupling is the parameter gradient: ONE sum all-reduce
of the flat fp32 buffer [grad_weight_real | grad_weight_imag | grad_bias] (2*D*F + D floats,
263 168 B at D=256) per layer per step, over RCCL/xGMI (backend "nccl") or gloo on CPU.

The message is latency-bound.  Two schedules are offered (`GradSync(mode=...)`):

  "overlap"  backward is cut in three: SMX_PHASE_SPECTRUM on the main stream, then the gradient
             reduction (SMX_PHASE_PARAMS) + the collective on a side stream while the grad_x inverse
             transform (SMX_PHASE_INVERSE) runs on the main one.  Hides the collective completely but
             pays ~20 us for the second launch and the parked spectrum (one-GPU rehearsal: 0.91x).
  "fused"    the single fused backward launch on the main stream, then PARAMS + the collective on the
             side stream; the main stream waits for it at the end of backward.  Nothing is hidden, but
             nothing is added to the transform either: better whenever the collective is short.

What is synchronised: exactly the gradients the native ops produce -- weight_real, weight_imag, bias
of every SpectralMixingLayer and, for the fused SpectralMLPBlock line, norm1.weight / norm1.bias
(a second tiny collective).  The reduction is a SUM (a mean-type loss is scaled by the loss, not by
the collective).  Every OTHER parameter of a model (norm2, mlp, embeddings ...) is not touched: reduce
those with `all_reduce_grads(params)` after backward, or wrap the model in DDP with the spectral
parameters and norm1 listed in `_ddp_params_and_buffers_to_ignore` -- wrapping them as well would
reduce them twice.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def shard_batch(B: int, rank: int, world: int) -> slice:
    """Rows of the global batch owned by `rank` (contiguous, remainder spread over low ranks)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


class _Handle:
    def __init__(self, work=None, stream=None, event=None):
        self.work, self.stream, self.event = work, stream, event

    def wait(self) -> None:
        if self.event is not None:                # GPU: make the caller's stream wait, no host sync
            torch.cuda.current_stream().wait_event(self.event)
        elif self.work is not None:
            self.work.wait()


class GradSync:
    """SUM all-reduce of a flat gradient buffer, asynchronous with respect to the compute stream."""

    MODES = ("overlap", "fused")

    def __init__(self, group: Optional[dist.ProcessGroup] = None, mode: str = "overlap"):
        if mode not in self.MODES:
            raise ValueError(f"mode must be one of {self.MODES}, got {mode!r}")
        self.group = group
        self.mode = mode
        self._side = None

    def active(self) -> bool:
        """True when a collective will actually be issued (otherwise backward stays fused)."""
        if not dist.is_initialized():
            return False
        # SMX_FORCE_SYNC=1 keeps the collective at world size 1 (rehearsal of the N>1 code path)
        return dist.get_world_size(self.group) > 1 or os.environ.get("SMX_FORCE_SYNC") == "1"

    def all_reduce(self, flat: torch.Tensor, pre=None) -> _Handle:
        """Sum `flat` over the group.  `pre` (optional callable) fills `flat` and is run where the
        collective runs -- on the side stream for GPU tensors -- so both overlap the caller's next work."""
        if not self.active():
            if pre is not None:
                pre()
            return _Handle()
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                if pre is not None:
                    pre()
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                done = torch.cuda.Event()
                done.record(self._side)
            flat.record_stream(self._side)
            return _Handle(stream=self._side, event=done)
        if pre is not None:
            pre()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _Handle(work=work)


def attach_grad_sync(module: torch.nn.Module, group: Optional[dist.ProcessGroup] = None,
                     mode: str = "overlap"):
    """Make every SpectralMixingLayer under `module` SUM-all-reduce the gradients of its own parameters
    (and, through the fused SpectralMLPBlock line, of norm1) inside backward; `mode` as in the module
    docstring.  Parameters that do not belong to a SpectralMixingLayer / norm1 are NOT synchronised --
    see `all_reduce_grads`.  Returns the module."""
    from .spectral_layers import SpectralMixingLayer
    sync = GradSync(group, mode)
    for m in module.modules():
        if isinstance(m, SpectralMixingLayer):
            m._grad_sync = sync
    return module


def flatten_grads(params: Iterable[torch.Tensor]) -> torch.Tensor:
    return torch.cat([p.grad.reshape(-1) for p in params])


def all_reduce_grads(params: Iterable[torch.nn.Parameter],
                     group: Optional[dist.ProcessGroup] = None) -> None:
    """Post-hoc variant: one flat SUM all-reduce of the .grad of `params`, written back in place.
    Used where the gradients were produced without attach_grad_sync (and by the CPU/gloo tests)."""
    params = [p for p in params if p.grad is not None]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = flatten_grads(params)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
