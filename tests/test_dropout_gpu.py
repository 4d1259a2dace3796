"""Training-mode dropout of SpectralMixingLayer.forward (reference fft_tensor/spectral_layers.py:68, :118:
`y = dropout(y + bias)`) fused into the transform's launches.  The reference's mask comes from torch's
generator and cannot be reproduced bit for bit by anyone; what is checked is everything that IS defined:

  * y_train = mask * y_eval / (1 - p) element by element, for a 0/1 mask with the right rate;
  * backward uses exactly the mask of its forward (gradients equal the eval-mode backward of g * mask / (1-p));
  * masks differ from call to call, repeat under torch.manual_seed, and are redrawn at every hipGraph replay;
  * all transform plans (single launch, split, direct) and the fused block half agree on the above.
"""
import numpy as np
import pytest
import torch

from conftest import TOL_ACT, TOL_PARAM, rel_err

pytestmark = pytest.mark.gpu


def _mods():
    import tensor_cuda_fft_amd as pkg
    from tensor_cuda_fft_amd import _lib, functional
    return pkg, _lib, functional


def _layer(pkg, D, F, p, dev, seed=3):
    torch.manual_seed(seed)
    layer = pkg.SpectralMixingLayer(D, num_filters=F, dropout=p).to(dev)
    with torch.no_grad():
        layer.weight_real.normal_(1.0, 0.5); layer.weight_imag.normal_(0.0, 0.5)
        layer.bias.normal_(0.5, 0.1)                 # away from zero: an exact 0.0 in y is a dropped element
    return layer


def _grads(layer, x, g):
    x = x.clone().requires_grad_(True)
    for q in layer.parameters():
        q.grad = None
    y = layer(x)
    y.backward(g)
    torch.cuda.synchronize()
    return y.detach(), x.grad, layer.weight_real.grad.clone(), layer.weight_imag.grad.clone(), \
        layer.bias.grad.clone()


CASES = [  # (B, N, D, F, option)            plan
    (8, 1024, 64, 32, None),                # decimated, single launch (one band)
    (4, 512, 40, 20, None),                 # ragged d-tile
    (2, 4096, 16, 8, ("nsplit", 4)),        # split plan
    (3, 300, 10, 5, None),                  # direct plan
    (2, 512, 14, 200, None),                # two bands
    (3, 2000, 40, 20, None),                # sixteen-row decimation (N = 16 * 125)
    (2, 1200, 12, 200, None),               # ... two bands
    (2, 4096, 8, 600, None),                # more than 512 bins: four-step plan (the mask as one more native pass)
    (2, 2048, 6, 700, None),                # ... eight tiles
    (1, 16384, 4, 3000, None),              # ... two-level four-step columns
    (2, 8704, 6, 600, None),                # ... 34 tiles: band groups (mask on a row copy in the workspace)
]


@pytest.mark.parametrize("B,N,D,F,opt", CASES)
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_training_forward_backward_consistent_with_eval(gpu, B, N, D, F, opt, p):
    pkg, lib, _ = _mods()
    if opt:
        lib.set_option(*opt)
    try:
        layer = _layer(pkg, D, F, p, gpu)
        x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
        layer.eval()
        y_eval = _grads(layer, x, g)[0]
        layer.train()
        y, gx, gwr, gwi, gb = _grads(layer, x, g)
        mask = (y != 0)
        thr = round(p * 65536)
        keep = 1.0 - thr / 65536.0
        scale = 1.0 / keep
        n = mask.numel()
        assert abs(mask.float().mean().item() - keep) <= 5.0 * (keep * (1 - keep) / n) ** 0.5 + 1e-3
        ref = torch.where(mask, y_eval * scale, torch.zeros_like(y_eval))
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) <= TOL_ACT
        # backward of the SAME mask: eval-mode backward of g * mask * scale
        layer.eval()
        _, gx_r, gwr_r, gwi_r, gb_r = _grads(layer, x, g * mask * scale)
        assert rel_err(gx.cpu().numpy(), gx_r.cpu().numpy()) <= TOL_ACT
        for a, r in ((gwr, gwr_r), (gwi, gwi_r), (gb, gb_r)):
            assert rel_err(a.cpu().numpy(), r.cpu().numpy()) <= TOL_PARAM
    finally:
        if opt:
            lib.set_option(opt[0], 0)


def test_masks_change_per_call_and_repeat_under_manual_seed(gpu):
    pkg, _, _ = _mods()
    x = torch.randn(4, 512, 32, device=gpu)

    def two_masks():
        torch.manual_seed(99)
        layer = _layer(pkg, 32, 16, 0.3, gpu, seed=99).train()
        return [(layer(x) != 0) for _ in range(2)]

    a0, a1 = two_masks()
    b0, b1 = two_masks()
    assert torch.equal(a0, b0) and torch.equal(a1, b1)           # reproducible
    agree = (a0 == a1).float().mean().item()
    assert abs(agree - (0.7 * 0.7 + 0.3 * 0.3)) < 0.01            # independent draws
    # rows and batch entries are not copies of each other
    assert (a0[0] == a0[1]).float().mean().item() < 0.65
    assert (a0[:, :-1] == a0[:, 1:]).float().mean().item() < 0.65
    assert (a0[..., 0::2] == a0[..., 1::2]).float().mean().item() < 0.65


def test_eval_mode_and_p0_are_untouched_and_fuse_flag(gpu):
    pkg, _, _ = _mods()
    layer = _layer(pkg, 32, 16, 0.25, gpu)
    x = torch.randn(2, 256, 32, device=gpu)
    layer.eval()
    y0 = layer(x)
    assert (y0 != 0).all()
    layer.train()
    layer.fuse_dropout = False                       # torch's nn.Dropout as a separate pass
    torch.manual_seed(5); ya = layer(x)
    torch.manual_seed(5); yb = torch.nn.functional.dropout(y0, 0.25, True)
    assert torch.equal(ya, yb)
    layer.fuse_dropout = True
    assert not torch.equal(layer(x), ya)


@pytest.mark.parametrize("B,N,D,F", [(4, 1024, 64, 32), (2, 4096, 8, 600), (2, 8704, 8, 600)])   # one launch; > 512 bins: four-step, band groups
def test_block_training_path_is_fused_and_consistent(gpu, B, N, D, F):
    """SpectralMLPBlock(dropout=0.1).train(): the first residual line still runs as one native op;
    y - x is the dropped-out mix, and backward matches the eval composition fed with the same mask."""
    pkg, _, fn = _mods()
    torch.manual_seed(21)
    p = 0.1
    x = (0.5 + torch.randn(B, N, D, device=gpu)); g = torch.randn(B, N, D, device=gpu)
    lw = (1 + 0.3 * torch.randn(D, device=gpu)); lb = 0.2 * torch.randn(D, device=gpu)
    wr = (1 + 0.5 * torch.randn(D, F, device=gpu)); wi = 0.5 * torch.randn(D, F, device=gpu)
    bias = 0.5 + 0.1 * torch.randn(D, device=gpu)
    leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, wr, wi, bias)]
    st = fn.DropoutState(gpu)
    y = fn.spectral_block_mix(leaves[0], leaves[1], leaves[2], 1e-5, leaves[3], leaves[4], leaves[5],
                              None, dropout_p=p, drop_state=st)
    y.backward(g)
    got = [y.detach()] + [t.grad for t in leaves]
    # reference: composition of separately tested native ops, mask inferred from the forward result
    ref_leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, wr, wi, bias)]
    h = torch.nn.functional.layer_norm(ref_leaves[0], (D,), ref_leaves[1], ref_leaves[2], 1e-5)
    m = fn.spectral_mix(h, ref_leaves[3], ref_leaves[4], ref_leaves[5])
    scale = 65536.0 / (65536 - round(p * 65536))
    m_train = got[0] - x
    mask = (m_train.abs() > 1e-6 * m.detach().abs().clamp_min(1e-3))
    assert abs(mask.float().mean().item() - 1 / scale) < 0.01
    y_ref = ref_leaves[0] + m * mask * scale
    y_ref.backward(g)
    assert rel_err(got[0].cpu().numpy(), y_ref.detach().cpu().numpy()) <= TOL_ACT
    for i, (a, t) in enumerate(zip(got[1:], ref_leaves)):
        tol = TOL_ACT if i == 0 else TOL_PARAM
        assert rel_err(a.cpu().numpy(), t.grad.cpu().numpy()) <= tol, i
    # and the module takes that path in training mode
    blk = pkg.SpectralMLPBlock(D, mlp_ratio=2, dropout=p).to(gpu).train()
    assert blk._fusable(x)
    blk.spectral_mix.fuse_dropout = False
    assert not blk._fusable(x)


def test_graph_replay_draws_a_new_mask(gpu):
    pkg, _, _ = _mods()
    layer = _layer(pkg, 32, 16, 0.5, gpu).train()
    x = torch.randn(4, 512, 32, device=gpu)
    s = torch.cuda.Stream(gpu)
    s.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(s), torch.no_grad():
        layer(x)                                         # tables, workspace, generator state
    torch.cuda.current_stream(gpu).wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(gr):
        y = layer(x)
    masks = []
    for _ in range(3):
        gr.replay()
        torch.cuda.synchronize()
        masks.append((y != 0).clone())
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    assert abs((masks[0] == masks[1]).float().mean().item() - 0.5) < 0.02


def test_dropout_argument_checks(gpu):
    _, lib, fn = _mods()
    x = torch.randn(2, 256, 16, device=gpu)
    w = torch.ones(16, 8, device=gpu)
    with pytest.raises(ValueError):
        fn.spectral_mix(x, w, w, None, dropout_p=0.2)                 # no generator state
    with pytest.raises(ValueError):
        fn.spectral_mix(x, w, w, None, dropout_p=1.0, drop_state=fn.DropoutState(gpu))
    y = torch.empty_like(x)
    rc = lib.lib().smx_forward_dropout(x.data_ptr(), w.data_ptr(), w.data_ptr(), None, y.data_ptr(), None,
                                       None, 0, 2, 256, 16, 8, 0, 0.25, None, None, None)
    assert rc == -1 and b"rng_state" in lib.lib().smx_last_error()


@pytest.mark.parametrize("B,N,D,F", [(64, 4096, 256, 128)])
def test_full_size_dropout_rate_and_adjoint(gpu, B, N, D, F):
    """C2-sized: keep rate to 5 sigma, and <dropout-layer(x) - dropped bias, g> = <x, grad_x> (the masked
    operator is still linear in x for a fixed mask)."""
    pkg, _, fn = _mods()
    torch.manual_seed(4)
    p = 0.1
    wr = (1 + 0.5 * torch.randn(D, F, device=gpu)); wi = 0.5 * torch.randn(D, F, device=gpu)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    st = fn.DropoutState(gpu)
    rng = st.next()
    y, xk = fn.forward_raw(x, wr, wi, None, save_spectrum=True, dropout_p=p, rng=rng)
    gx, _ = fn.backward_raw(g, xk, wr, wi, dropout_p=p, rng=rng)
    keep = 1 - round(p * 65536) / 65536
    rate = (y != 0).float().mean().item()
    assert abs(rate - keep) < 5 * (keep * (1 - keep) / y.numel()) ** 0.5 + 1e-4
    lhs = torch.sum(y.double() * g.double()).item()
    rhs = torch.sum(x.double() * gx.double()).item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-2 * y.numel() ** 0.5


@pytest.mark.parametrize("B,N,D,F,opt", CASES)
def test_gpu_mask_is_the_documented_function_of_position(gpu, B, N, D, F, opt):
    """Whatever the plan, element (b, n, d) is dropped iff the CPU evaluation of the same hash
    (tests/emu, compiled from smx_core.h) drops element n*D + d of batch row b."""
    import ctypes, os, subprocess
    from conftest import ROOT
    pkg, lib, fn = _mods()
    subprocess.run(["bash", os.path.join(ROOT, "tests", "emu", "build.sh")], check=True, capture_output=True)
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libsmx_emu.so"))
    if opt:
        lib.set_option(*opt)
    try:
        p = 0.3
        wr = torch.ones(D, F, device=gpu); wi = torch.zeros(D, F, device=gpu)
        bias = torch.ones(D, device=gpu)
        x = torch.randn(B, N, D, device=gpu)
        rng = fn.DropoutState(gpu).next()
        y, _ = fn.forward_raw(x, wr, wi, bias, dropout_p=p, rng=rng)
        seed, counter = (int(v) & (2**64 - 1) for v in rng.cpu().tolist())
        got = (y != 0).cpu().numpy()
        for b in range(B):
            exp = np.zeros(N * D, np.uint8)
            emu.emu_drop_mask(ctypes.c_ulonglong(seed), ctypes.c_ulonglong(counter), b,
                              ctypes.c_longlong(N * D), round(p * 65536),
                              exp.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
            assert np.array_equal(got[b].reshape(-1), exp.astype(bool)), b
    finally:
        if opt:
            lib.set_option(opt[0], 0)


@pytest.mark.parametrize("B,N,D,F,opt", CASES)
def test_phase_split_backward_with_dropout(gpu, B, N, D, F, opt):
    """SPECTRUM / PARAMS / INVERSE issued separately (the multi-GPU order) with the same generator words
    give the single-call gradients, for every plan."""
    _, lib, fn = _mods()
    if opt:
        lib.set_option(*opt)
    try:
        torch.manual_seed(8)
        p = 0.2
        wr = 1 + 0.5 * torch.randn(D, F, device=gpu); wi = 0.5 * torch.randn(D, F, device=gpu)
        x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
        rng = fn.DropoutState(gpu).next()
        kw = dict(dropout_p=p, rng=rng)
        _, xk = fn.forward_raw(x, wr, wi, None, save_spectrum=True, **kw)
        gx, flat = fn.backward_raw(g, xk, wr, wi, **kw)
        gx2, flat2 = fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_SPECTRUM, **kw)
        fn.backward_raw(g, xk, wr, wi, want_x=False, phases=fn.PHASE_PARAMS, flat=flat2, **kw)
        fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_INVERSE, grad_x=gx2, flat=flat2, **kw)
        torch.cuda.synchronize()
        assert rel_err(gx2.cpu().numpy(), gx.cpu().numpy()) <= 2e-6
        assert rel_err(flat2.cpu().numpy(), flat.cpu().numpy()) <= 2e-6
    finally:
        if opt:
            lib.set_option(opt[0], 0)


def test_activation_checkpointing_regenerates_the_same_mask(gpu):
    """torch.utils.checkpoint re-runs the forward during backward with the device RNG state restored;
    the fused dropout draws its key from that generator, so the recomputed mask is the original one
    and the gradients equal those of the plain run."""
    from torch.utils.checkpoint import checkpoint
    pkg, _, _ = _mods()
    layer = _layer(pkg, 32, 16, 0.3, gpu).train()
    x = torch.randn(4, 512, 32, device=gpu)
    g = torch.randn(4, 512, 32, device=gpu)

    def run(use_ckpt):
        torch.manual_seed(123)
        xx = x.clone().requires_grad_(True)
        for q in layer.parameters():
            q.grad = None
        y = checkpoint(layer, xx, use_reentrant=False) if use_ckpt else layer(xx)
        y.backward(g)
        return y.detach(), xx.grad, layer.weight_real.grad.clone(), layer.bias.grad.clone()

    a = run(False)
    b = run(True)
    assert torch.equal(a[0] != 0, b[0] != 0)                        # same mask
    for u, v in zip(a, b):
        assert rel_err(u.cpu().numpy(), v.cpu().numpy()) <= 2e-6


def test_dropout_on_the_eight_band_plan(gpu):
    """Option fourstep = 0 sends N = 2048 with more than 512 bins to the eight-band kernel: the mask goes on as one more
    native pass there too (same generator words, same function of position as everywhere else); a phase-split SPECTRUM
    call, which that plan does not serve with a mask, is refused with a message."""
    _, lib, fn = _mods()
    B, N, D, F, p = 2, 2048, 6, 700, 0.3
    torch.manual_seed(3)
    wr = 1 + 0.5 * torch.randn(D, F, device=gpu); wi = 0.5 * torch.randn(D, F, device=gpu)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    rng = fn.DropoutState(gpu).next()
    y_fs, xk = fn.forward_raw(x, wr, wi, None, save_spectrum=True, dropout_p=p, rng=rng)
    gx_fs, flat_fs = fn.backward_raw(g, xk, wr, wi, dropout_p=p, rng=rng)
    lib.set_option("fourstep", 0)
    try:
        assert lib.plan(B, N, D, F).bands == 8
        y8, xk8 = fn.forward_raw(x, wr, wi, None, save_spectrum=True, dropout_p=p, rng=rng)
        gx8, flat8 = fn.backward_raw(g, xk8, wr, wi, dropout_p=p, rng=rng)
        torch.cuda.synchronize()
        assert torch.equal(y8 != 0, y_fs != 0)
        assert rel_err(y8.cpu().numpy(), y_fs.cpu().numpy()) <= TOL_ACT
        assert rel_err(gx8.cpu().numpy(), gx_fs.cpu().numpy()) <= TOL_ACT
        assert rel_err(flat8.cpu().numpy(), flat_fs.cpu().numpy()) <= TOL_PARAM
        with pytest.raises(Exception, match="one call"):
            fn.backward_raw(g, xk8, wr, wi, phases=fn.PHASE_SPECTRUM, dropout_p=p, rng=rng)
    finally:
        lib.set_option("fourstep", 1)
