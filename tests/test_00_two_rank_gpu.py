"""World-size-2 run of the REAL HIP path: two processes share cuda:0 and talk over gloo.

This is SURVEY 8(e)'s parity check on hardware: every rank runs forward + backward of its batch shard
through libsmx.so with `attach_grad_sync` (SPECTRUM on the main stream, PARAMS + collective on the side
stream, INVERSE underneath), the all-reduced parameter gradients are compared with the fp64 oracle on
the CONCATENATED batch, forward / grad_x shard by shard.  The same is done for SpectralMLPBlock, where
every parameter gradient the fused op returns (filter, bias, norm1) must come out identical on both
ranks and equal to the single-process block on the whole batch.

The file sorts first on purpose: the children are started before this pytest process has initialised
the GPU (the box forbids replacing the image of a process that already holds the device).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, TOL_ACT, TOL_PARAM, rel_err

pytestmark = pytest.mark.gpu

WORLD = 2
SHAPE = dict(B=8, N=2048, D=64, F=32)            # decimated fused plan on every shard
BLOCK = dict(B=6, N=1024, D=32, F=16)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(B, N, D, F, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, D, generator=g); gr = torch.randn(B, N, D, generator=g)
    wr = 1 + 0.5 * torch.randn(D, F, generator=g); wi = 0.5 * torch.randn(D, F, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    lw = 1 + 0.3 * torch.randn(D, generator=g); lb = 0.2 * torch.randn(D, generator=g)
    return x, gr, wr, wi, b, lw, lb


def worker(rank, world, port, out_dir):
    """Runs in a child process (python tests/test_00_two_rank_gpu.py worker ...)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tensor_cuda_fft_amd as pkg
    dev = torch.device("cuda:0")
    out = {}

    # ---- the layer ---------------------------------------------------------------------------------
    x, g, wr, wi, b, _, _ = _inputs(seed=11, **SHAPE)
    layer = pkg.SpectralMixingLayer(SHAPE["D"], num_filters=SHAPE["F"]).to(dev)
    layer.load_state_dict({"weight_real": wr, "weight_imag": wi, "bias": b})
    pkg.attach_grad_sync(layer)
    assert layer._grad_sync.active()
    sl = pkg.shard_batch(SHAPE["B"], rank, world)
    xs = x[sl].to(dev).requires_grad_(True)
    y = layer(xs)
    y.backward(g[sl].to(dev))
    torch.cuda.synchronize()
    out.update(y=y.detach().cpu().numpy(), gx=xs.grad.cpu().numpy(),
               gwr=layer.weight_real.grad.cpu().numpy(), gwi=layer.weight_imag.grad.cpu().numpy(),
               gb=layer.bias.grad.cpu().numpy(), lo=sl.start, hi=sl.stop)

    # ---- the fused block half + MLP: every gradient of the native op is synced ------------------------
    x, g, wr, wi, b, lw, lb = _inputs(seed=12, **BLOCK)
    torch.manual_seed(3)                                   # same MLP init on both ranks
    blk = pkg.SpectralMLPBlock(BLOCK["D"], mlp_ratio=1, dropout=0.0).to(dev)
    with torch.no_grad():
        blk.spectral_mix.weight_real.copy_(wr); blk.spectral_mix.weight_imag.copy_(wi)
        blk.spectral_mix.bias.copy_(b); blk.norm1.weight.copy_(lw); blk.norm1.bias.copy_(lb)
    pkg.attach_grad_sync(blk)
    sl = pkg.shard_batch(BLOCK["B"], rank, world)
    xs = x[sl].to(dev).requires_grad_(True)
    yb = blk(xs)
    yb.backward(g[sl].to(dev))
    # the rest of the block (norm2, mlp) goes through the documented post-hoc reduction
    native = {"spectral_mix.weight_real", "spectral_mix.weight_imag", "spectral_mix.bias",
              "norm1.weight", "norm1.bias"}
    pkg.all_reduce_grads([p for n, p in blk.named_parameters() if n not in native])
    torch.cuda.synchronize()
    out.update(by=yb.detach().cpu().numpy(), bgx=xs.grad.cpu().numpy(), blo=sl.start, bhi=sl.stop)
    for n, p in blk.named_parameters():
        out["bgrad." + n] = p.grad.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    with open("/proc/self/maps") as f:
        loaded = any("libsmx.so" in line for line in f)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"libsmx_loaded": loaded}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_single_process_oracle(tmp_path):
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU visible")
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: run the file on its own (it sorts first in "
                    "the suite for that reason)")
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(WORLD),
                               str(port), str(tmp_path)], env=env) for r in range(WORLD)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=540))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    assert rcs == [0] * WORLD, rcs

    from oracle import spectral_oracle as so
    z = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(WORLD)]
    for r in range(WORLD):
        assert json.load(open(tmp_path / f"rank{r}.json"))["libsmx_loaded"]

    # ---- layer: oracle on the concatenated batch ----------------------------------------------------
    x, g, wr, wi, b, _, _ = (t.numpy() for t in _inputs(seed=11, **SHAPE))
    y_ref, _ = so.forward_closed(x, wr, wi, b)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x, wr, wi, g)
    for r in range(WORLD):
        lo, hi = int(z[r]["lo"]), int(z[r]["hi"])
        assert rel_err(z[r]["y"], y_ref[lo:hi]) <= TOL_ACT
        assert rel_err(z[r]["gx"], gx_ref[lo:hi]) <= TOL_ACT
        assert rel_err(z[r]["gwr"], gwr_ref) <= TOL_PARAM
        assert rel_err(z[r]["gwi"], gwi_ref) <= TOL_PARAM
        assert rel_err(z[r]["gb"], gb_ref) <= TOL_PARAM
    for k in ("gwr", "gwi", "gb"):                       # identical bits on both ranks after the sum
        assert np.array_equal(z[0][k], z[1][k])

    # ---- block: single-process torch composition of the oracle on the whole batch ------------------
    x, g, wr, wi, b, lw, lb = _inputs(seed=12, **BLOCK)
    D = BLOCK["D"]
    # the drop-in block built on the CPU only supplies norm2 / mlp with the ranks' initial weights
    import tensor_cuda_fft_amd as pkg
    torch.manual_seed(3)
    ref_blk = pkg.SpectralMLPBlock(D, mlp_ratio=1, dropout=0.0)
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    xr, wr_, wi_, b_, lw_, lb_ = map(leaf, (x, wr, wi, b, lw, lb))
    h = torch.nn.functional.layer_norm(xr, (D,), lw_, lb_, ref_blk.norm1.eps)
    x1 = xr + so.forward_port(h, wr_, wi_, b_)
    yr = x1 + ref_blk.mlp(ref_blk.norm2(x1))
    yr.backward(g)
    ref = {"spectral_mix.weight_real": wr_.grad, "spectral_mix.weight_imag": wi_.grad,
           "spectral_mix.bias": b_.grad, "norm1.weight": lw_.grad, "norm1.bias": lb_.grad}
    for n, p in ref_blk.named_parameters():
        if n not in ref:
            ref[n] = p.grad
    for r in range(WORLD):
        lo, hi = int(z[r]["blo"]), int(z[r]["bhi"])
        assert rel_err(z[r]["by"], yr.detach().numpy()[lo:hi]) <= 2e-5
        assert rel_err(z[r]["bgx"], xr.grad.numpy()[lo:hi]) <= 2e-5
        for n, gref in ref.items():
            assert rel_err(z[r]["bgrad." + n], gref.numpy()) <= TOL_PARAM, (r, n)
    for n in ref:
        assert np.array_equal(z[0]["bgrad." + n], z[1]["bgrad." + n]), n


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
