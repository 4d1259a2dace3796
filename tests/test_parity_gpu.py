"""Parity of the HIP path against the reference's golden vectors and the oracle.  All GPU-only.

Tolerances: max|delta| <= 1e-5 max|ref| for y / grad_x, 1e-4 for parameter grads.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import TOL_ACT, TOL_PARAM, golden_names, load_golden, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu

LAYER = [n for n in golden_names("layer") if "nolearn" not in n]


def _mods():
    import tensor_cuda_fft_amd as pkg
    from tensor_cuda_fft_amd import _lib, functional
    return pkg, _lib, functional


def _layer_from_golden(z, dev):
    pkg, _, _ = _mods()
    D = z["x"].shape[2]
    layer = pkg.SpectralMixingLayer(D, num_filters=int(z["num_filters"])).to(dev)
    layer.load_state_dict({k: torch.from_numpy(z[k]) for k in ("weight_real", "weight_imag", "bias")})
    return layer


def _run_layer(layer, x, g):
    x = x.clone().requires_grad_(True)
    for p in layer.parameters():
        p.grad = None
    y = layer(x)
    y.backward(g)
    torch.cuda.synchronize()
    return (y.detach().cpu().numpy(), x.grad.cpu().numpy(), layer.weight_real.grad.cpu().numpy(),
            layer.weight_imag.grad.cpu().numpy(), layer.bias.grad.cpu().numpy())


def _check(got, z):
    y, gx, gwr, gwi, gb = got
    assert rel_err(y, z["y"]) <= TOL_ACT
    assert rel_err(gx, z["grad_x"]) <= TOL_ACT
    assert rel_err(gwr, z["grad_w_real"]) <= TOL_PARAM
    assert rel_err(gwi, z["grad_w_imag"]) <= TOL_PARAM
    assert rel_err(gb, z["grad_bias"]) <= TOL_PARAM


@pytest.mark.parametrize("name", LAYER)
def test_golden_through_module(gpu, name):
    z = load_golden(name)
    layer = _layer_from_golden(z, gpu)
    got = _run_layer(layer, torch.from_numpy(z["x"]).to(gpu), torch.from_numpy(z["g"]).to(gpu))
    _check(got, z)
    if name.startswith("G10"):                       # unused filter columns: exactly zero, not stale
        k = so.num_bins(z["x"].shape[1], int(z["num_filters"]))
        assert not got[2][:, k:].any() and not got[3][:, k:].any()


@pytest.mark.parametrize("name", [n for n in LAYER if n[:3] in ("G02", "G08", "G09", "G14", "G15", "G16", "G17",
                                                              "G18")])
@pytest.mark.parametrize("variant", ["nsplit2", "nsplit_max", "direct", "nostagger", "rotated"])
def test_golden_all_kernel_variants(gpu, name, variant):
    """Same fixtures through the three-launch split path, the direct path and without staggering."""
    _, _lib, _ = _mods()
    z = load_golden(name)
    if variant == "direct" and z["x"].shape[1] > 8192:
        pytest.skip("direct path is O(N k): keep it to short sequences")
    opts = {"nsplit2": ("nsplit", 2), "nsplit_max": ("nsplit", 1 << 20), "direct": ("force_direct", 1),
            "nostagger": ("placement", 0), "rotated": ("placement", 1)}[variant]
    _lib.set_option(*opts)
    try:
        layer = _layer_from_golden(z, gpu)
        got = _run_layer(layer, torch.from_numpy(z["x"]).to(gpu), torch.from_numpy(z["g"]).to(gpu))
    finally:
        _lib.set_option("nsplit", 0); _lib.set_option("force_direct", 0); _lib.set_option("placement", 2)
    _check(got, z)


def test_golden_nolearn_identity(gpu):
    pkg, _, _ = _mods()
    z = load_golden("G11_nolearn_2x128x32")
    layer = pkg.SpectralMixingLayer(32, learnable=False).to(gpu)
    assert len(layer.state_dict()) == 0
    x = torch.from_numpy(z["x"]).to(gpu).requires_grad_(True)
    y = layer(x)
    y.backward(torch.from_numpy(z["g"]).to(gpu))
    assert rel_err(y.detach().cpu().numpy(), z["y"]) <= TOL_ACT
    assert rel_err(x.grad.cpu().numpy(), z["grad_x"]) <= TOL_ACT


def test_golden_wirtinger(gpu):
    pkg, _, _ = _mods()
    z = load_golden("G12_wirtinger_2x32x16")
    filt = pkg.WirtingerSpectralFilter(16, 8).to(gpu)
    filt.load_state_dict({"weight.real": torch.from_numpy(z["w_real"]),
                          "weight.imag": torch.from_numpy(z["w_imag"])})
    xf = torch.from_numpy(z["x_freq"]).to(gpu).requires_grad_(True)
    gf = torch.from_numpy(z["g_freq"]).to(gpu)
    out = filt(xf)
    out.backward(gf)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(out), z["out"]) <= TOL_ACT
    assert rel_err(c(xf.grad), z["grad_x_freq"]) <= TOL_ACT
    assert rel_err(c(filt.weight.real.grad), z["grad_w_real"]) <= TOL_PARAM
    assert rel_err(c(filt.weight.imag.grad), z["grad_w_imag"]) <= TOL_PARAM
    # raw WirtingerGradient.apply on (B,k,D) x (1,k,D)
    k = 8
    xs = torch.from_numpy(z["x_freq"][:, :k]).to(gpu).requires_grad_(True)
    ws = torch.from_numpy((z["w_real"] + 1j * z["w_imag"])[:, :k].T[None].astype(np.complex64)).to(gpu)
    ws.requires_grad_(True)
    o2 = pkg.WirtingerGradient.apply(xs, ws)
    o2.backward(gf[:, :k].contiguous())
    assert rel_err(c(o2), z["mul_out"]) <= TOL_ACT
    assert rel_err(c(xs.grad), z["mul_grad_x"]) <= TOL_ACT
    assert rel_err(c(ws.grad), z["mul_grad_w"]) <= TOL_PARAM
    assert tuple(ws.grad.shape) == (1, k, 16)


def test_wirtinger_fused_equals_layer(gpu):
    """ifft(filter(fft(x))).real == SpectralMixingLayer with zero bias."""
    pkg, _, _ = _mods()
    torch.manual_seed(5)
    B, N, D, F = 2, 512, 64, 48
    filt = pkg.WirtingerSpectralFilter(D, F).to(gpu)
    with torch.no_grad():
        filt.weight.real.normal_(1.0, 0.5); filt.weight.imag.normal_(0.0, 0.5)
    x = torch.randn(B, N, D, device=gpu)
    y_fused = pkg.spectral_mix_with_filter(x, filt)
    y_unfused = torch.fft.ifft(filt(torch.fft.fft(x, dim=1)), dim=1).real
    assert rel_err(y_fused.detach().cpu().numpy(), y_unfused.detach().cpu().numpy()) <= 2 * TOL_ACT


SHAPES = [  # (B, N, D, F)  decimated path unless noted
    (8, 512, 256, 128),                                                     # BASELINE config C1
    (3, 256, 32, 16), (2, 1024, 64, 128), (1, 2048, 30, 77), (2, 1280, 66, 129),
    (1, 8192, 10, 256), (5, 512, 2, 1), (2, 4096, 96, 48),
    (2, 4096, 64, 512), (3, 1024, 34, 300), (16, 2048, 1024, 512),          # four bands (k <= 512)
    (2, 100, 7, 9), (3, 33, 5, 4), (1, 640, 9, 300), (2, 300, 16, 200), (1, 2048, 8, 700),   # direct path
    (2, 1000, 64, 32), (2, 4000, 32, 16), (3, 1500, 130, 700), (4, 4000, 255, 128), (2, 999, 33, 499),
    (64, 128, 256, 128),                                  # direct path, LDS-tiled kernels (large problems)
]


@pytest.mark.parametrize("B,N,D,F", SHAPES)
def test_random_shapes_vs_oracle(gpu, B, N, D, F):
    pkg, _, _ = _mods()
    rng = np.random.default_rng(B * 1000 + N + D + F)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    layer.load_state_dict({"weight_real": torch.from_numpy(wr), "weight_imag": torch.from_numpy(wi),
                           "bias": torch.from_numpy(b)})
    got = _run_layer(layer, torch.from_numpy(x).to(gpu), torch.from_numpy(g).to(gpu))
    y, _ = so.forward_closed(x, wr, wi, b)
    gx, gwr, gwi, gb = so.backward_closed(x, wr, wi, g)
    _check(got, {"y": y, "grad_x": gx, "grad_w_real": gwr, "grad_w_imag": gwi, "grad_bias": gb})


def test_pruned_rfft_matches_fft(gpu):
    pkg, _, _ = _mods()
    torch.manual_seed(3)
    for (B, N, D, F) in [(2, 1024, 64, 100), (2, 96, 6, 20), (1, 4096, 34, 256)]:
        x = torch.randn(B, N, D, device=gpu)
        xk = pkg.pruned_rfft(x, F)
        ref = np.fft.rfft(x.cpu().numpy().astype(np.float64), axis=1)[:, :min(F, N // 2)]
        assert rel_err(xk.cpu().numpy(), ref) <= TOL_ACT


@pytest.mark.parametrize("B,N,D,F", [(4, 2048, 64, 32), (48, 512, 256, 64), (16, 1024, 128, 200)])
def test_split_backward_phases_equal_fused(gpu, B, N, D, F):
    """smx_backward phases 1 then 2 (what the multi-GPU overlap uses) == phases 3, both when the
    plan is residue-split (few workgroups) and when phase 1 is the single fused launch."""
    _, _lib, fn = _mods()
    torch.manual_seed(11)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    wr = torch.randn(D, F, device=gpu); wi = torch.randn(D, F, device=gpu)
    _, xk = fn.forward_raw(x, wr, wi, None, save_spectrum=True)
    gx3, flat3 = fn.backward_raw(g, xk, wr, wi, phases=7)
    gx12, flat12 = fn.backward_raw(g, xk, wr, wi, phases=1 | 4)
    fn.backward_raw(g, xk, wr, wi, phases=2, grad_x=gx12, flat=flat12)
    torch.cuda.synchronize()
    assert rel_err(gx12.cpu().numpy(), gx3.cpu().numpy()) <= 2e-6
    assert rel_err(flat12.cpu().numpy(), flat3.cpu().numpy()) <= 2e-6


def test_errors_and_type_checks(gpu):
    pkg, _lib, _ = _mods()
    layer = pkg.SpectralMixingLayer(16).to(gpu)
    with pytest.raises(AssertionError, match="Expected embed_dim=16, got 8"):
        layer(torch.randn(1, 32, 8, device=gpu))
    with pytest.raises(TypeError):
        layer(torch.randn(1, 32, 16, device=gpu, dtype=torch.float64))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        layer(torch.randn(1, 32, 16))
    lib = _lib.lib()
    assert lib.smx_forward(None, None, None, None, None, None, None, 0, 1, 256, 2, 1, 0, None) \
        == -1
    assert b"non-NULL" in lib.smx_last_error()
    # non-contiguous input is accepted (made contiguous), like the reference
    x = torch.randn(2, 16, 64, device=gpu).transpose(1, 2)          # (2, 64, 16)
    y = layer(x)
    ref, _ = so.forward_closed(x.cpu().numpy(), np.ones((16, 8), np.float32),
                               np.zeros((16, 8), np.float32), np.zeros(16, np.float32))
    assert rel_err(y.detach().cpu().numpy(), ref) <= TOL_ACT


def test_dropout_and_mlp_block_run(gpu):
    pkg, _, _ = _mods()
    torch.manual_seed(0)
    blk = pkg.SpectralMLPBlock(64, dropout=0.1).to(gpu)
    assert "spectral_mix.weight_real" in blk.state_dict()
    x = torch.randn(2, 256, 64, device=gpu, requires_grad=True)
    y = blk(x)
    y.square().mean().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    blk.eval()
    with torch.no_grad():
        y1, y2 = blk(x), blk(x)
    assert torch.equal(y1, y2)


# ---- BASELINE.json full sizes: size-independent properties --------------------------------------
def _rand_layer(pkg, D, F, dev, seed=1234):
    torch.manual_seed(seed)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(dev)
    with torch.no_grad():
        layer.weight_real.normal_(1.0, 0.5); layer.weight_imag.normal_(0.0, 0.5)
        layer.bias.normal_(0.0, 0.1)
    return layer


@pytest.mark.parametrize("B,N,D,F", [(64, 4096, 256, 128), (8, 65536, 256, 128), (64, 4096, 512, 256)])
def test_full_size_properties(gpu, B, N, D, F):
    pkg, _, fn = _mods()
    layer = _rand_layer(pkg, D, F, gpu)
    gen = torch.Generator(device=gpu).manual_seed(1234)
    x = torch.randn(B, N, D, device=gpu, generator=gen)
    g = torch.randn(B, N, D, device=gpu, generator=gen)
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    y.backward(g)
    gx = xr.grad
    bias = layer.bias.detach()
    # (1) adjoint identity: <y - bias, g> == <x, grad_x>   (the layer is linear in x)
    lhs = ((y.detach() - bias).double() * g.double()).sum().item()
    rhs = (x.double() * gx.double()).sum().item()
    scale = (y.detach() - bias).double().norm().item() * g.double().norm().item()
    assert abs(lhs - rhs) <= 1e-6 * scale
    # (2) linearity: f(2x + x_rolled_batch) - bias == 2 (f(x)-bias) + (f(x_rolled)-bias)
    with torch.no_grad():
        xs = torch.roll(x, 1, 0)
        y2 = layer(2 * x + xs) - bias
        yl = 2 * (y.detach() - bias) + (layer(xs) - bias)
        assert (y2 - yl).abs().max().item() <= 1e-5 * yl.abs().max().item()
    # (3) grad_bias is the plain sum of g
    assert rel_err(layer.bias.grad.cpu().numpy(), g.double().sum((0, 1)).cpu().numpy()) <= TOL_PARAM
    # (4) band-limited round trip: with W0 = 1, Wf = 2 (f>=1), bias 0, a signal that only has bins < k
    #     comes back unchanged (the layer is an exact projector onto that band)
    k = so.num_bins(N, F)
    with torch.no_grad():
        layer.weight_real.fill_(2.0); layer.weight_real[:, 0] = 1.0
        layer.weight_imag.zero_(); layer.bias.zero_()
        n = torch.arange(N, device=gpu, dtype=torch.float64)[None, :, None]
        d = torch.arange(D, device=gpu, dtype=torch.float64)[None, None, :]
        f1 = 1 + (d.long() % (k - 1)).double()
        xb = (0.5 + torch.cos(2 * np.pi * f1 * n / N + 0.1 * d)
              + 0.25 * torch.sin(2 * np.pi * (k - 1) * n / N)).float().expand(2, N, D).contiguous()
        yb = layer(xb)
        assert (yb - xb).abs().max().item() <= 1e-5 * xb.abs().max().item()
    # (5) a sub-batch agrees with the fp64 oracle to the stated tolerances (incl. parameter grads)
    layer2 = _rand_layer(pkg, D, F, gpu)
    sb = 1 if N > 8192 else 2
    dsub = slice(0, D)
    xs_, gs_ = x[:sb].contiguous(), g[:sb].contiguous()
    got = _run_layer(layer2, xs_, gs_)
    wr, wi, b = (t.detach().cpu().numpy() for t in (layer2.weight_real, layer2.weight_imag, layer2.bias))
    yo, _ = so.forward_closed(xs_.cpu().numpy(), wr, wi, b)
    gxo, gwro, gwio, gbo = so.backward_closed(xs_.cpu().numpy(), wr, wi, gs_.cpu().numpy())
    _check(got, {"y": yo, "grad_x": gxo, "grad_w_real": gwro, "grad_w_imag": gwio, "grad_bias": gbo})


def test_loaded_library_is_in_tree(gpu):
    """The round-end harness records which .so the GPU tests loaded: it must be our in-tree one."""
    _, _lib, _ = _mods()
    _lib.lib()
    with open("/proc/self/maps") as f:
        assert any(_lib.LIB_PATH in line for line in f)


def test_bitwise_deterministic(gpu):
    """The batch reduction of the parameter gradients has a fixed order (no atomics): two runs of
    the same step are bit-identical, also through the three-launch split path."""
    pkg, _lib, _ = _mods()
    layer = _rand_layer(pkg, 64, 32, gpu)
    gen = torch.Generator(device=gpu).manual_seed(7)
    x = torch.randn(16, 1024, 64, device=gpu, generator=gen)
    g = torch.randn(16, 1024, 64, device=gpu, generator=gen)
    for opt in (("nsplit", 0), ("nsplit", 4)):
        _lib.set_option(*opt)
        try:
            a = _run_layer(layer, x, g)
            b = _run_layer(layer, x, g)
        finally:
            _lib.set_option("nsplit", 0)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


def test_hipgraph_replay_matches_eager(gpu):
    """The whole step is capturable (no allocation or host sync inside the C ABI once the twiddle
    tables exist) and a replay reproduces the eager result bit for bit.
    (Round 1 saw ONE segfault in capture_end of an earlier form of this test, inside the full pytest
    process; tools/capture_lifetime_repro.py could not reproduce it in isolation in seven variants --
    with / without libsmx, warm-up results kept / dropped, round 1's workspace cache put back --
    profiles/r02_capture_repro.txt.  What has changed since is listed in INTEGRATION.md, "hipGraph capture".)"""
    pkg, _, _ = _mods()
    layer = _rand_layer(pkg, 64, 32, gpu)
    x = torch.randn(4, 2048, 64, device=gpu, requires_grad=True)
    g = torch.randn(4, 2048, 64, device=gpu)

    def step():
        y = layer(x)
        y.backward(g)
        return y.detach()

    def zero():
        x.grad = None
        layer.zero_grad(set_to_none=True)

    step(); zero()                                   # builds tables / workspace before the capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    zero()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y_g = step()
        out = (y_g, x.grad, layer.weight_real.grad, layer.bias.grad)
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in out]
    zero()
    y_e = step()
    ref = (y_e, x.grad, layer.weight_real.grad, layer.bias.grad)
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_two_streams_do_not_share_workspace(gpu):
    """Calls on different streams use different workspaces (the split path writes partial spectra)."""
    pkg, _lib, fn = _mods()
    _lib.set_option("nsplit", 4)
    try:
        wr = torch.randn(32, 16, device=gpu); wi = torch.randn(32, 16, device=gpu)
        xs = [torch.randn(8, 4096, 32, device=gpu) for _ in range(2)]
        ref = [fn.forward_raw(x, wr, wi, None)[0].clone() for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [None, None]
        for rep in range(5):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i] = fn.forward_raw(xs[i], wr, wi, None)[0]
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)
    finally:
        _lib.set_option("nsplit", 0)


def test_mlp_block_matches_reference_block(gpu):
    """The immediate caller (reference SpectralMLPBlock, spectral_layers.py:135-190): the reference's
    own state_dict loads unchanged and forward/backward match its CPU run (LayerNorm / MLP on torch,
    spectral mix on the HIP path)."""
    pkg, _, _ = _mods()
    z = load_golden("B01_mlpblock_2x512x64")
    blk = pkg.SpectralMLPBlock(64, mlp_ratio=2, dropout=0.0).to(gpu)
    sd = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}
    assert set(sd) == set(blk.state_dict())
    blk.load_state_dict(sd)
    x = torch.from_numpy(z["x"]).to(gpu).requires_grad_(True)
    y = blk(x)
    y.backward(torch.from_numpy(z["g"]).to(gpu))
    assert rel_err(y.detach().cpu().numpy(), z["y"]) <= 2e-5
    assert rel_err(x.grad.cpu().numpy(), z["grad_x"]) <= 2e-5
    for name, p in blk.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["grad." + name]) <= TOL_PARAM, name


def test_make_graphed_callables(gpu):
    """torch.cuda.make_graphed_callables wraps the layer (forward AND backward captured): the way an
    eager training loop gets rid of the ~0.3 ms of per-step host overhead."""
    pkg, _, _ = _mods()
    layer = _rand_layer(pkg, 64, 32, gpu)
    ref_layer = _rand_layer(pkg, 64, 32, gpu)
    x = torch.randn(4, 1024, 64, device=gpu, requires_grad=True)
    g = torch.randn(4, 1024, 64, device=gpu)
    y = ref_layer(x); y.backward(g)                      # builds tables / workspace
    ref = (y.detach().clone(), x.grad.clone(), ref_layer.weight_real.grad.clone())
    x.grad = None
    graphed = torch.cuda.make_graphed_callables(layer, (torch.randn_like(x).requires_grad_(True),))
    y2 = graphed(x)
    y2.backward(g)
    torch.cuda.synchronize()
    assert rel_err(y2.detach().cpu().numpy(), ref[0].cpu().numpy()) <= 1e-6
    assert rel_err(x.grad.cpu().numpy(), ref[1].cpu().numpy()) <= 1e-6
    assert rel_err(layer.weight_real.grad.cpu().numpy(), ref[2].cpu().numpy()) <= 1e-6


def test_backward_with_grad_sync_hook_matches_fused(gpu):
    """The multi-GPU backward order (SPECTRUM on the main stream, PARAMS + collective on a side stream,
    INVERSE underneath) gives the fused backward's numbers; the collective here is a recording stub."""
    pkg, _, fn = _mods()
    from tensor_cuda_fft_amd.distributed import GradSync

    class StubSync(GradSync):
        def __init__(self):
            super().__init__()
            self.seen = []

        def active(self):
            return True

        def all_reduce(self, flat, pre=None):
            side = torch.cuda.Stream(device=flat.device)
            ready = torch.cuda.Event(); ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                if pre is not None:
                    pre()
                flat.mul_(2.0)                           # stands in for a world-size-2 sum of equal shards
                done = torch.cuda.Event(); done.record(side)
            flat.record_stream(side)
            self.seen.append(flat.numel())
            from tensor_cuda_fft_amd.distributed import _Handle
            return _Handle(stream=side, event=done)

    for B, N, D, F in [(64, 1024, 256, 128), (2, 4096, 64, 32), (2, 100, 8, 4)]:
        layer = _rand_layer(pkg, D, F, gpu)
        x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
        ref = _run_layer(layer, x, g)
        layer._grad_sync = StubSync()
        got = _run_layer(layer, x, g)
        assert layer._grad_sync.seen == [2 * D * F + D]
        layer._grad_sync = None
        assert rel_err(got[0], ref[0]) == 0 and rel_err(got[1], ref[1]) <= 2e-6
        for a, r in zip(got[2:], ref[2:]):
            assert rel_err(a, 2.0 * r) <= 2e-6


def test_double_backward_is_refused(gpu):
    """The native backward is not itself differentiable; asking for it raises instead of returning zeros."""
    pkg, _, _ = _mods()
    layer = _rand_layer(pkg, 16, 8, gpu)
    x = torch.randn(2, 256, 16, device=gpu, requires_grad=True)
    (gx,) = torch.autograd.grad(layer(x).sum(), x, create_graph=True)
    with pytest.raises(RuntimeError, match="once_differentiable|differentiate twice|does not require grad"):
        gx.sum().backward()


def test_more_than_2_31_elements(gpu):
    """(4, 2^20, 512): 2^31 elements, 8 GiB per tensor -- every index product must be 64-bit.  Checked on
    the LAST batch row / last channel pair (largest offsets) against the fp64 closed form, plus the
    adjoint identity over the whole tensor.  Split plan, L = 4096."""
    pkg, lib, fn = _mods()
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * 2**30:
        pytest.skip("needs ~40 GiB of device memory")
    B, N, D, F = 4, 1 << 20, 512, 128
    assert B * N * D == 1 << 31
    torch.manual_seed(31)
    wr = (1 + 0.5 * torch.randn(D, F, device=gpu)); wi = 0.5 * torch.randn(D, F, device=gpu)
    bias = 0.1 * torch.randn(D, device=gpu)
    x = torch.randn(B, N, D, device=gpu)
    y, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True)
    g = torch.randn(B, N, D, device=gpu)
    gx, flat = fn.backward_raw(g, xk, wr, wi)
    torch.cuda.synchronize()
    sl = (slice(B - 1, B), slice(None), slice(D - 2, D))
    w2r, w2i, b2 = (t[D - 2:].cpu().numpy() for t in (wr, wi, bias))
    y_ref, _ = so.forward_closed(x[sl].cpu().numpy(), w2r, w2i, b2)
    gx_ref, _, _, _ = so.backward_closed(x[sl].cpu().numpy(), w2r, w2i, g[sl].cpu().numpy())
    assert rel_err(y[sl].cpu().numpy(), y_ref) <= TOL_ACT
    assert rel_err(gx[sl].cpu().numpy(), gx_ref) <= TOL_ACT
    # adjoint identity  <y - bias, g> = <x, grad_x>  in fp64, row by row to bound temporaries
    lhs = rhs = 0.0
    for b in range(B):
        lhs += torch.sum((y[b] - bias).double() * g[b].double()).item()
        rhs += torch.sum(x[b].double() * gx[b].double()).item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0) + 1e-3 * (B * N * D) ** 0.5
    gb = flat[2 * D * F:]
    assert rel_err(gb.cpu().numpy(), g.sum(dim=(0, 1), dtype=torch.float64).cpu().numpy()) <= TOL_PARAM


def test_misaligned_and_strided_views_are_accepted(gpu):
    """Inputs that are views at an odd element offset, or non-contiguous, give the numbers of a fresh
    contiguous copy (the wrapper copies when the 8/16-byte vector accesses would be misaligned)."""
    pkg, _, fn = _mods()
    B, N, D, F = 2, 512, 32, 16
    layer = _rand_layer(pkg, D, F, gpu)
    base = torch.randn(B * N * D + 3, device=gpu)
    x_off = base[1:1 + B * N * D].view(B, N, D)                     # starts 4 bytes into the buffer
    assert x_off.data_ptr() % 8 != 0 and x_off.is_contiguous()
    gbase = torch.randn(B * N * D + 3, device=gpu)
    g_off = gbase[3:3 + B * N * D].view(B, N, D)
    ref = _run_layer(layer, x_off.clone(), g_off.clone())
    got = _run_layer(layer, x_off, g_off)
    for a, r in zip(got, ref):
        assert np.array_equal(a, r)
    xt = torch.randn(N, B, D, device=gpu).transpose(0, 1)           # non-contiguous
    ref = _run_layer(layer, xt.contiguous(), g_off.clone())
    got = _run_layer(layer, xt, g_off)
    for a, r in zip(got, ref):
        assert np.array_equal(a, r)
    # block op and pruned_rfft take the same care
    st = fn.pruned_rfft(x_off, F)
    assert torch.equal(st, fn.pruned_rfft(x_off.clone(), F))
    yb = fn.spectral_block_mix(x_off, None, None, 1e-5, layer.weight_real, layer.weight_imag, layer.bias)
    assert torch.equal(yb, fn.spectral_block_mix(x_off.clone(), None, None, 1e-5, layer.weight_real,
                                                 layer.weight_imag, layer.bias))


# ---- more than 512 kept bins: band groups ----------------------------------------------------------
BIGK = [(4, 2048, 64, 1000), (2, 8192, 32, 3000), (3, 1024, 6, 512 + 1), (2, 2048, 10, 1024),
        (2, 3072, 8, 1536), (2, 4096, 16, 2048)]


@pytest.mark.parametrize("B,N,D,F", BIGK)
def test_band_groups_vs_oracle(gpu, B, N, D, F):
    """k > 512 runs the four-band kernels once per group of 512 bins plus the edge bins (multiples of 512):
    forward, backward (single call and SPECTRUM / PARAMS / INVERSE separately) and the spectrum op."""
    pkg, lib, fn = _mods()
    k = so.num_bins(N, F)
    p = lib.plan(B, N, D, F)
    if 5 <= N // 256 <= 16 or N // 256 == 32:                                            # four-step path
        assert p.path == lib.SMX_PATH_DECIMATED and (p.groups, p.bands) == (1, 0)
    else:
        assert p.path == lib.SMX_PATH_DECIMATED and p.groups == (k + 511) // 512 and p.bands == 4
    gen = torch.Generator().manual_seed(N + F)
    x = torch.randn(B, N, D, generator=gen); g = torch.randn(B, N, D, generator=gen)
    wr = 1 + 0.5 * torch.randn(D, F, generator=gen); wi = 0.5 * torch.randn(D, F, generator=gen)
    bias = 0.1 * torch.randn(D, generator=gen)
    y_ref, X_ref = so.forward_closed(x.numpy(), wr.numpy(), wi.numpy(), bias.numpy())
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x.numpy(), wr.numpy(), wi.numpy(), g.numpy())
    xd, gd, wrd, wid, bd = (t.to(gpu) for t in (x, g, wr, wi, bias))
    y, xk = fn.forward_raw(xd, wrd, wid, bd, save_spectrum=True)
    gx, flat = fn.backward_raw(gd, xk, wrd, wid)
    assert rel_err(y.cpu().numpy(), y_ref) <= TOL_ACT
    assert rel_err(xk.cpu().numpy(), X_ref) <= TOL_ACT
    assert rel_err(gx.cpu().numpy(), gx_ref) <= TOL_ACT
    DF = D * F
    assert rel_err(flat[:DF].view(D, F).cpu().numpy(), gwr_ref) <= TOL_PARAM
    assert rel_err(flat[DF:2 * DF].view(D, F).cpu().numpy(), gwi_ref) <= TOL_PARAM
    assert rel_err(flat[2 * DF:].cpu().numpy(), gb_ref) <= TOL_PARAM
    assert torch.count_nonzero(flat[:DF].view(D, F)[:, k:]) == 0            # unused columns exactly zero
    # the three phases issued separately
    gx2, flat2 = fn.backward_raw(gd, xk, wrd, wid, phases=fn.PHASE_SPECTRUM)
    fn.backward_raw(gd, xk, wrd, wid, want_x=False, phases=fn.PHASE_PARAMS, flat=flat2)
    fn.backward_raw(gd, xk, wrd, wid, phases=fn.PHASE_INVERSE, grad_x=gx2, flat=flat2)
    assert rel_err(gx2.cpu().numpy(), gx.cpu().numpy()) <= 2e-6
    assert rel_err(flat2.cpu().numpy(), flat.cpu().numpy()) <= 2e-6
    # spectrum-only op
    assert rel_err(fn.pruned_rfft(xd, F).cpu().numpy(), X_ref) <= TOL_ACT


def test_more_than_512_bins_modules_stay_native(gpu):
    """More than 512 kept bins (round 4): the block's first line and the training-mode dropout are native on the four-step
    plan AND on the band-group plan (the mask as one more native pass; band groups work from a row copy in the workspace
    because their launches re-read the input while the output accumulates)."""
    pkg, lib, fn = _mods()
    D = 1280                                           # default num_filters = 640 > 512
    blk = pkg.SpectralMLPBlock(D, mlp_ratio=1, dropout=0.1).to(gpu)
    sm = blk.spectral_mix
    assert lib.plan(2, 2048, D, D // 2).bands == 0 and lib.plan(2, 8704, D, D // 2).groups == 2
    for N in (2048, 8704):                             # eight tiles: four-step plan; 34 tiles: band groups
        x = torch.randn(2, N, D, device=gpu, requires_grad=True)
        blk.train()
        assert blk._fusable(x)
        y = blk(x)
        y.sum().backward()
        assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
        blk.eval()                                     # eval: the native block line against the composition, fwd + bwd
        g = torch.randn_like(x)
        res = []
        for native in (True, False):
            xr = x.detach().clone().requires_grad_(True)
            blk.zero_grad(set_to_none=True)
            if native:
                h = fn.spectral_block_mix(xr, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, sm.weight_real,
                                          sm.weight_imag, sm.bias, None)
            else:
                h = xr + sm(blk.norm1(xr))
            h.backward(g)
            res.append([h.detach(), xr.grad, sm.weight_real.grad.clone(), blk.norm1.weight.grad.clone()])
        for i, (a, b) in enumerate(zip(*res)):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= (TOL_ACT if i < 2 else TOL_PARAM), (N, i)


def test_ready_filter_pack_is_reused(gpu):
    """A forward call handed an already packed filter (SMX_FILTER_PACK_READY) gives the same output and
    really reads that buffer: with a different filter packed in it, the output follows the buffer."""
    _, _, fn = _mods()
    torch.manual_seed(2)
    B, N, D, F = 16, 4096, 128, 200                                   # 8 Mi samples, two bands: a shape that packs
    assert fn._new_pack(torch.empty(32, 4096, 64, device=gpu), torch.empty(64, 32, device=gpu)) is None   # one band
                                                                      # never does (LDS-staged filter slices)
    x = torch.randn(B, N, D, device=gpu)
    wr = torch.randn(D, F, device=gpu); wi = torch.randn(D, F, device=gpu)
    pack = fn._new_pack(x, wr)
    y0, _ = fn.forward_raw(x, wr, wi, None, pack=pack)
    y1, _ = fn.forward_raw(x, wr, wi, None, pack=pack, pack_ready=True)
    assert torch.equal(y0, y1)
    assert torch.equal(pack, torch.complex(wr, wi)[:, :pack.shape[0]].T.contiguous())     # the documented layout
    y2, _ = fn.forward_raw(x, 2 * wr, 2 * wi, None, pack=pack, pack_ready=True)             # weights ignored
    assert torch.equal(y2, y0)


def test_hybrid_attention_matches_reference(gpu):
    """HybridSpectralAttention (reference spectral_layers.py:193-256): the reference's state_dict loads
    unchanged; output and every gradient match its CPU run (spectral mix native, attention through torch)."""
    pkg, _, _ = _mods()
    z = load_golden("A01_hybrid_2x256x64")
    m = pkg.HybridSpectralAttention(64, num_heads=int(z["heads"]), dropout=0.0).to(gpu)
    sd = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    x = torch.from_numpy(z["x"]).to(gpu).requires_grad_(True)
    y = m(x)
    y.backward(torch.from_numpy(z["g"]).to(gpu))
    assert rel_err(y.detach().cpu().numpy(), z["y"]) <= 2e-5
    assert rel_err(x.grad.cpu().numpy(), z["grad_x"]) <= 2e-5
    for name, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), z["grad." + name]) <= TOL_PARAM, name
