"""World-size-2 gloo test of the batch-sharded gradient sync (runs on CPU).

The transform itself cannot run here (no GPU), so each rank gets its shard's gradients from the
oracle; what is under test is the distributed logic: contiguous sharding, ONE flat sum all-reduce
of [grad_weight_real | grad_weight_imag | grad_bias], parity with the single-process reference on
the concatenated batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tensor_cuda_fft_amd as pkg
    from oracle import spectral_oracle as so

    torch.manual_seed(7)                                   # same global batch / weights everywhere
    B, N, D, F = 6, 64, 8, 4
    x = torch.randn(B, N, D); g = torch.randn(B, N, D)
    layer = pkg.SpectralMixingLayer(D, num_filters=F)
    with torch.no_grad():
        layer.weight_real.normal_(1, .5); layer.weight_imag.normal_(0, .5); layer.bias.normal_(0, .1)
    sl = pkg.shard_batch(B, rank, world)
    _, gx, gwr, gwi, gb = so.fwd_bwd_port(x[sl], layer.weight_real, layer.weight_imag, layer.bias, g[sl])
    flat = torch.cat([gwr.reshape(-1), gwi.reshape(-1), gb.reshape(-1)]).clone()
    layer.weight_real.grad, layer.weight_imag.grad, layer.bias.grad = gwr, gwi, gb

    # (a) post-hoc flat all-reduce
    pkg.all_reduce_grads(layer.parameters())
    # (b) the in-backward object: same buffer layout the HIP autograd Function hands it, filled by the
    #     `pre` callable (SMX_PHASE_PARAMS in the product) right before the collective is issued
    staged = flat.clone()
    flat.zero_()
    h = pkg.GradSync().all_reduce(flat, pre=lambda: flat.copy_(staged))
    h.wait()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gwr=layer.weight_real.grad.numpy(),
             gwi=layer.weight_imag.grad.numpy(), gb=layer.bias.grad.numpy(), flat=flat.numpy(),
             gx=gx.numpy(), lo=sl.start, hi=sl.stop)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_flat_allreduce_reproduces_single_process_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import tensor_cuda_fft_amd as pkg
    from oracle import spectral_oracle as so
    torch.manual_seed(7)
    B, N, D, F = 6, 64, 8, 4
    x = torch.randn(B, N, D); g = torch.randn(B, N, D)
    layer = pkg.SpectralMixingLayer(D, num_filters=F)
    with torch.no_grad():
        layer.weight_real.normal_(1, .5); layer.weight_imag.normal_(0, .5); layer.bias.normal_(0, .1)
    _, gx, gwr, gwi, gb = so.fwd_bwd_port(x, layer.weight_real, layer.weight_imag, layer.bias, g)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.allclose(z["gwr"], gwr.numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(z["gwi"], gwi.numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(z["gb"], gb.numpy(), rtol=1e-5, atol=1e-5)
        ref_flat = np.concatenate([gwr.numpy().ravel(), gwi.numpy().ravel(), gb.numpy().ravel()])
        assert np.allclose(z["flat"], ref_flat, rtol=1e-5, atol=1e-5)
        # forward / grad_x need no communication: shard-by-shard equality with the full batch
        assert np.allclose(z["gx"], gx.numpy()[int(z["lo"]):int(z["hi"])], rtol=1e-5, atol=1e-6)


def test_sync_is_a_noop_without_a_process_group():
    import tensor_cuda_fft_amd as pkg
    flat = torch.arange(5.0)
    pkg.GradSync().all_reduce(flat).wait()
    assert torch.equal(flat, torch.arange(5.0))
    ran = []
    pkg.GradSync().all_reduce(flat, pre=lambda: ran.append(1)).wait()      # `pre` still runs (it fills flat)
    assert ran == [1]
    layer = pkg.attach_grad_sync(pkg.SpectralMixingLayer(4))
    assert isinstance(layer._grad_sync, pkg.GradSync)
