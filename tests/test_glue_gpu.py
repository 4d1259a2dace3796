"""Round-3 glue: workspace lifetime under stream capture, twiddle-table eviction, the row-scale predicate under
every phase split, lazily conjugated input of the sequence FFT.  All through the
C ABI on the GPU."""
import numpy as np
import pytest
import torch

from conftest import TOL_ACT, TOL_PARAM, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _pkg():
    import tensor_cuda_fft_amd as pkg
    from tensor_cuda_fft_amd import _lib, functional
    return pkg, _lib, functional


def test_workspace_of_a_captured_call_lives_in_the_graphs_pool(gpu):
    """The condition behind round 1's capture_end segfault candidate (hipGraph capture): a workspace
    allocated INSIDE a capture comes from that graph's private pool.  Round 1 put such a buffer into the global
    per-stream cache; later eager calls on the same stream id then ran on memory owned by a graph that might be
    gone.  Now: (1) a captured call never reads or writes the eager cache, (2) nothing allocated in a capture is
    cached, (3) the eager workspace is untouched by a capture, replay after the eager cache was dropped still
    works, and an eager call after the graph died gets a buffer of its own."""
    pkg, lib, fn = _pkg()
    dev = gpu
    B, N, D, F = 2, 8192, 64, 32                      # residue-split plan: forward AND backward use the workspace
    assert lib.plan(B, N, D, F).nsplit > 1
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(dev)
    with torch.no_grad():
        layer.weight_real.normal_(1, .5); layer.weight_imag.normal_(0, .5); layer.bias.normal_(0, .1)
    x = torch.randn(B, N, D, device=dev, requires_grad=True)
    g = torch.randn(B, N, D, device=dev)

    def step():
        y = layer(x)
        y.backward(g)
        out = (y.detach().clone(), x.grad.clone(), layer.weight_real.grad.clone())
        x.grad = None
        layer.zero_grad(set_to_none=True)
        return out

    fn.release_workspaces()
    ref = step()                                      # eager: tables + the eager workspace of the current stream
    torch.cuda.synchronize()
    eager_keys = dict(fn._ws_cache)
    assert len(eager_keys) == 1
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cap = step()
    # (1)+(2): the capture neither added an entry for its capture stream nor replaced the eager buffer
    assert {k: v.data_ptr() for k, v in fn._ws_cache.items()} == {k: v.data_ptr() for k, v in eager_keys.items()}
    fn.release_workspaces()                           # (3) the graph does not depend on the eager cache
    junk = torch.full((64 << 20,), 7, dtype=torch.uint8, device=dev)     # may reuse the freed eager block
    gr.replay(); gr.replay()
    torch.cuda.synchronize()
    for a, b in zip(cap, ref):
        assert torch.equal(a, b)
    del gr, cap, junk
    torch.cuda.synchronize()
    again = step()                                    # eager after the graph (and its pool) is gone
    torch.cuda.synchronize()
    for a, b in zip(again, ref):
        assert torch.equal(a, b)


def test_table_eviction_is_per_device_pinned_and_visible_to_python(gpu):
    pkg, lib, fn = _pkg()
    L = lib.lib()
    x = torch.randn(1, 256, 4, device=gpu)
    wr = torch.randn(4, 2, device=gpu); wi = torch.randn(4, 2, device=gpu)
    lens = [256 * m for m in (37, 39, 41, 43)]        # lengths nothing else in the suite uses
    e0 = L.smx_tables_epoch()
    lib.set_option("table_cache_entries", 1)
    try:
        for n in lens:
            fn._prepare(gpu, n)
        assert L.smx_tables_epoch() > e0              # something was evicted
        # Python's "prepared" set follows the epoch: it cannot claim a length whose tables are gone ...
        fn._prepare(gpu, lens[-1])
        assert (gpu.index, lens[0]) not in fn._prepared
        # ... so a capture of an evicted length is refused up front by the library, cleanly, and works after prepare
        xs = torch.randn(1, lens[0], 4, device=gpu)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        y = torch.empty_like(xs)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            gr.capture_begin()
            rc = L.smx_forward(xs.data_ptr(), wr.data_ptr(), wi.data_ptr(), None, y.data_ptr(), None, None, 0,
                               1, lens[0], 4, 2, 0, s.cuda_stream)
            gr.capture_end()
        assert rc == -2
        torch.cuda.current_stream().wait_stream(s)
    finally:
        lib.set_option("table_cache_entries", 256)
    # results after eviction + re-upload are the same numbers
    a = fn.spectral_mix(xs, wr, wi)
    b = fn.spectral_mix(xs, wr, wi)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    y_ref, _ = so.forward_closed(xs.cpu().numpy(), wr.cpu().numpy(), wi.cpu().numpy(), None)
    assert rel_err(a.cpu().numpy(), y_ref) <= TOL_ACT


@pytest.mark.parametrize("fourstep", [1, 0])
def test_row_scale_with_a_frozen_input_on_every_plan_of_n2048(gpu, fourstep):
    """smx_row_scale_supported() said yes for the eight-band plan (N = 2048, k > 512, fourstep = 0) while
    backward accepted row_scale there only with SPECTRUM and INVERSE together -- a frozen x with a trainable gate
    (SPECTRUM | PARAMS) then failed after forward had succeeded.  The predicate now excludes that plan, the wrapper
    multiplies the output instead, and both settings give the oracle's numbers."""
    pkg, lib, fn = _pkg()
    B, R, D, n_fft, k = 3, 1024, 8, 2048, 1025
    F = k
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)
    with lib.options(fourstep=fourstep):
        key = (B, R, D, F, n_fft, k)
        assert fn.row_scale_supported(key) == bool(fourstep)
        xd = T(x).to(gpu)                                        # requires_grad False
        wrd, wid, scd = (T(a).to(gpu).requires_grad_(True) for a in (wr, wi, sc))
        y = fn.spectral_filter(xd, wrd, wid, None, n_fft=n_fft, k=k, row_scale=scd)
        y.backward(T(g).to(gpu))
        torch.cuda.synchronize()
    y0, _ = so.forward_closed_ex(x, wr, wi, None, n_fft, k)
    _, gwr_ref, gwi_ref, _ = so.backward_closed_ex(x, wr, wi, g * sc[:, None, :], n_fft, k)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y0 * sc[:, None, :]) <= TOL_ACT
    assert rel_err(c(wrd.grad), gwr_ref) <= TOL_PARAM and rel_err(c(wid.grad), gwi_ref) <= TOL_PARAM
    assert rel_err(c(scd.grad), (g.astype(np.float64) * y0).sum(axis=1)) <= TOL_PARAM


@pytest.mark.parametrize("N", [1024, 300])
def test_sequence_fft_accepts_a_lazily_conjugated_input(gpu, N):
    """torch.fft.fft(x.conj(), dim=1) works on the reference side (frequency_ops.py:201); x.conj() is a lazy view
    whose conj bit view_as_real refuses -- seq_fft materialises it (and checks alignment) through _dense."""
    pkg, lib, fn = _pkg()
    z = torch.randn(2, N, 5, dtype=torch.complex64, device=gpu)
    zc = z.conj()
    assert zc.is_conj()
    got = fn.seq_fft(zc)
    ref = torch.fft.fft(z.cpu().to(torch.complex128).conj(), dim=1)
    assert rel_err(got.cpu().numpy(), ref.numpy()) <= TOL_ACT
    zg = z.clone().requires_grad_(True)
    fn.seq_fft(zg.conj()).abs().sum().backward()                 # the backward path takes a conj-view gradient too
    assert torch.isfinite(torch.view_as_real(zg.grad.resolve_conj())).all()


def test_rank_one_conv_error_names_the_real_limits(gpu):
    pkg, lib, fn = _pkg()
    import ctypes
    sh = lib.smx_shape(2, 100, 8, 193, 384, 193)                # n_fft = 384: not a power of two
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.lib().smx_conv_workspace_bytes(sh, ctypes.byref(a), ctypes.byref(b)) == -2
    msg = lib.lib().smx_last_error().decode()
    assert "65536" in msg and "4096}" not in msg


@pytest.mark.parametrize("B,N,D,F", [(64, 4096, 256, 128), (16, 1024, 96, 40), (5, 512, 34, 17), (64, 1024, 512, 256),
                                     (130, 256, 64, 32)])
def test_parameter_gradients_folded_into_the_backward_launch(gpu, B, N, D, F):
    """VERDICT r2 #4a (option fold_gradw = 1, off by default -- measured slower, profiles/r03_fold_gradw_ab.txt): the
    k_gradw reduction rides behind the transform workgroups of the same launch (appended reduction workgroups that
    only wait for lower-numbered ones).  Same additions in the same order as the separate launch: bit-identical
    gradients, run after run, and the sync area is left clean (second call equals the first)."""
    pkg, lib, fn = _pkg()
    assert lib.plan(B, N, D, F).nsplit == 1
    torch.manual_seed(B + N)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    wr = torch.randn(D, F, device=gpu); wi = torch.randn(D, F, device=gpu); bias = torch.randn(D, device=gpu)
    _, xk = fn.forward_raw(x, wr, wi, bias, save_spectrum=True)
    outs = {}
    for fold in (1, 0, 1):
        with lib.options(fold_gradw=fold):
            gx, flat = fn.backward_raw(g, xk, wr, wi)
            torch.cuda.synchronize()
            outs.setdefault(fold, []).append((gx.clone(), flat.clone()))
    (gx1, f1), (gx1b, f1b) = outs[1]
    (gx0, f0), = outs[0]
    assert torch.equal(gx1, gx0) and torch.equal(f1, f0)          # folded == separate launch, bit for bit
    assert torch.equal(gx1, gx1b) and torch.equal(f1, f1b)        # and reproducible (counters were left at zero)
    xs, gs = x[:4].cpu().numpy(), g[:4].cpu().numpy()             # a sub-batch against the oracle
    with lib.options(fold_gradw=1):
        _, xk4 = fn.forward_raw(x[:4].contiguous(), wr, wi, bias, save_spectrum=True)
        _, flat4 = fn.backward_raw(g[:4].contiguous(), xk4, wr, wi)
    _, gwr_ref, gwi_ref, gb_ref = so.backward_closed(xs, wr.cpu().numpy(), wi.cpu().numpy(), gs)
    fl = flat4.cpu().numpy()
    assert rel_err(fl[:D * F].reshape(D, F), gwr_ref) <= TOL_PARAM
    assert rel_err(fl[D * F:2 * D * F].reshape(D, F), gwi_ref) <= TOL_PARAM
    assert rel_err(fl[2 * D * F:], gb_ref) <= TOL_PARAM


@pytest.mark.parametrize("other", ["rank_one_conv", "seq_fft", "dwconv3_backward", "spectral_gate_backward", "mix_backward"])
def test_sync_area_survives_other_calls_on_the_shared_workspace(gpu, other):
    """every workspace layout keeps its first 64 KiB for the flag words of the folded
    parameter-gradient reduction -- the rank-one convolution and the complex sequence FFT used to start their scratch
    at offset 0.  The autograd functions share ONE per-(device, stream) buffer and the layer's backward trusts the flag
    words its forward left zero (SYNC_CLEAN): layer forward, then the other op on the same stream, then the layer's
    backward must give the gradients of the unfolded reduction, bit for bit."""
    pkg, lib, fn = _pkg()
    B, N, D = 64, 1024, 256
    torch.manual_seed(7)
    layer = pkg.SpectralMixingLayer(D).to(gpu)
    with torch.no_grad():
        layer.weight_real.normal_(1.0, 0.5); layer.weight_imag.normal_(0.0, 0.5); layer.bias.normal_(0.0, 0.1)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    z = torch.randn(B, N, D // 2, 2, device=gpu)
    n_fft = 2 * N
    h_re = torch.randn(n_fft // 2 + 1, device=gpu); h_im = torch.randn(n_fft // 2 + 1, device=gpu)

    def run(fold):
        with lib.options(fold_gradw=fold):
            xr = x.clone().requires_grad_(True)
            y = layer(xr)
            if other == "rank_one_conv":            # writes tile spectra / partial sums into the shared buffer
                fn.rank_one_conv(x, h_re, h_im, None, n_fft)
            elif other == "dwconv3_backward":       # BicameralBlock's time path: partial sums of its backward
                xt = x.clone().requires_grad_(True)
                wt = torch.randn(D, 1, 3, device=gpu, requires_grad=True)
                fn.causal_dwconv3(xt, wt).backward(g)
            elif other == "spectral_gate_backward":  # the twin blocks' gate chain: per-workgroup partials of S1
                zc = torch.view_as_complex(z).clone().requires_grad_(True)
                ac = torch.randn(N, dtype=torch.complex64, device=gpu, requires_grad=True)
                qc = torch.rand(B, D // 2, device=gpu, requires_grad=True)
                fn.spectral_gate(zc, ac, q=qc).backward(torch.view_as_complex(z))
            elif other == "mix_backward":            # BicameralBlock's fusion line: partial sums of the two scalars
                ts = [x.clone().requires_grad_(True) for _ in range(3)]
                wm = torch.rand(2, device=gpu, requires_grad=True)
                fn.mix_paths(ts[0], ts[1], ts[2], None, wm).backward(g)
            else:
                fn.seq_fft(torch.view_as_complex(z))
            y.backward(g)
            torch.cuda.synchronize()
            out = [xr.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
            for p in layer.parameters():
                p.grad = None
            return out

    ref = run(0)
    for _ in range(2):                               # twice: the area must also be left clean
        got = run(1)
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,D,F", [(4, 1024, 255, 100), (2, 2048, 33, 16), (4, 4000, 255, 128), (6, 1000, 63, 31)])
def test_odd_channel_count_runs_the_streaming_kernels(gpu, B, N, D, F):
    """VERDICT r2 missing #2 (the odd-D half): the reference takes any D (spectral_layers.py:88); an odd D at a
    length the decimated kernels take is padded by one zero channel instead of falling to the O(N k) products.
    Same numbers as the oracle, gradients in the caller's shapes, columns >= k still exactly zero."""
    pkg, lib, fn = _pkg()
    assert lib.plan(B, N, D, F).path == lib.SMX_PATH_DIRECT
    assert lib.plan(B, N, D + 1, F).path == (lib.SMX_PATH_DECIMATED if N % 256 == 0 else
                                             lib.SMX_PATH_DECIM16 if N % 16 == 0 else lib.SMX_PATH_DIRECT)
    rng = np.random.default_rng(D)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    xd, wrd, wid, bd = (T(a).to(gpu).requires_grad_(True) for a in (x, wr, wi, b))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # the slow-plan warning must not fire any more
        y = fn.spectral_mix(xd, wrd, wid, bd)
    y.backward(T(g).to(gpu))
    torch.cuda.synchronize()
    assert y.shape == (B, N, D) and xd.grad.shape == (B, N, D) and wrd.grad.shape == (D, F) and bd.grad.shape == (D,)
    y_ref, _ = so.forward_closed(x, wr, wi, b)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x, wr, wi, g)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y_ref) <= TOL_ACT and rel_err(c(xd.grad), gx_ref) <= TOL_ACT
    assert rel_err(c(wrd.grad), gwr_ref) <= TOL_PARAM and rel_err(c(wid.grad), gwi_ref) <= TOL_PARAM
    assert rel_err(c(bd.grad), gb_ref) <= TOL_PARAM
    # the layer itself (module path) takes the same route
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    with torch.no_grad():
        layer.weight_real.copy_(T(wr)); layer.weight_imag.copy_(T(wi)); layer.bias.copy_(T(b))
    assert rel_err(c(layer(T(x).to(gpu))), y_ref) <= TOL_ACT


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_every_dft_product_kernel_family_agrees_with_the_oracle(gpu, mode):
    """Shapes the decimated kernels do not take (N % 256 != 0) run the pruned DFT as matrix products: literal fp64
    (0), LDS-tiled VALU (1), f32 MFMA (2), bf16 x 3 MFMA (3, default from round 3: every fp32 operand split into
    three bf16 terms, six exact products per product, fp32 accumulation).  All four within the stated tolerance on a
    ragged shape (D % 128 != 0, k % 32 != 0, N % 32 != 0) -- the split must not cost accuracy."""
    pkg, lib, fn = _pkg()
    B, N, D, F = 3, 1500, 130, 300
    rng = np.random.default_rng(77)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    lib.set_option("tiled_dft", mode)
    try:
        xd, wrd, wid, bd = (T(a).to(gpu).requires_grad_(True) for a in (x, wr, wi, b))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            y = fn.spectral_mix(xd, wrd, wid, bd)
        y.backward(T(g).to(gpu))
        torch.cuda.synchronize()
    finally:
        lib.set_option("tiled_dft", 3)
    y_ref, _ = so.forward_closed(x, wr, wi, b)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x, wr, wi, g)
    c = lambda t: t.detach().cpu().numpy()
    errs = (rel_err(c(y), y_ref), rel_err(c(xd.grad), gx_ref), rel_err(c(wrd.grad), gwr_ref),
            rel_err(c(wid.grad), gwi_ref), rel_err(c(bd.grad), gb_ref))
    assert errs[0] <= TOL_ACT and errs[1] <= TOL_ACT and max(errs[2:]) <= TOL_PARAM, errs
    assert errs[0] <= 1e-6 and errs[1] <= 1e-6, errs          # in fact all four sit at fp32 noise


@pytest.mark.parametrize("B,N,D,F", [(3, 4000, 64, 128), (2, 2000, 34, 60), (5, 128, 16, 64), (2, 48, 8, 24),
                                     (1, 16, 4, 8), (2, 4112, 32, 128), (70, 1200, 96, 100), (2, 6000, 256, 128),
                                     (2, 4000, 512, 256), (3, 2000, 34, 200), (2, 1200, 40, 256),     # two bands
                                     (40, 4000, 256, 256),                                            # ... with a packed filter
                                     (1, 8000, 64, 100), (2, 32016, 32, 128), (1, 16016, 34, 200),    # few workgroups: tiles split
                                     (2, 32, 4, 1), (3, 80, 6, 3)])                                   # DC only / three bins
def test_sixteen_row_decimation_for_lengths_that_are_multiples_of_16(gpu, B, N, D, F):
    """VERDICT r2 missing #2: N % 256 != 0 no longer means O(N k) DFT products when 16 | N -- k_fused16 runs one
    16-point transform per residue and O(N k / 16) accumulation, x read once, y written once (SMX_PATH_DECIM16).
    Against the fp64 oracle (output, input gradient, every parameter gradient, saved spectrum), against the DFT-product
    plan of the same library (option decim16 = 0), ragged D, a last tile with one residue (N = 16 * 257), one tile with
    most residues padding (N = 16, 48, 128)."""
    pkg, lib, fn = _pkg()
    assert lib.plan(B, N, D, F).path == lib.SMX_PATH_DECIM16
    if B <= 2 and N >= 8000:
        assert lib.plan(B, N, D, F).nsplit > 1                          # few workgroups, long sequence: tiles split
    with lib.options(decim16=0):
        assert lib.plan(B, N, D, F).path == lib.SMX_PATH_DIRECT
    rng = np.random.default_rng(N + D)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    outs = {}
    for on in (1, 0):
        with lib.options(decim16=on):
            xd, wrd, wid, bd = (T(a).to(gpu).requires_grad_(True) for a in (x, wr, wi, b))
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                y = fn.spectral_mix(xd, wrd, wid, bd)
                xk = fn.pruned_rfft(T(x).to(gpu), F)
            y.backward(T(g).to(gpu))
            torch.cuda.synchronize()
            outs[on] = [t.detach().cpu().numpy() for t in (y, xd.grad, wrd.grad, wid.grad, bd.grad, xk)]
    y_ref, xk_ref = so.forward_closed(x, wr, wi, b)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x, wr, wi, g)
    k = min(F, N // 2)
    for on in (1, 0):
        y, gx, gwr, gwi, gb, xk = outs[on]
        assert rel_err(y, y_ref) <= TOL_ACT and rel_err(gx, gx_ref) <= TOL_ACT, on
        assert rel_err(gwr, gwr_ref) <= TOL_PARAM and rel_err(gwi, gwi_ref) <= TOL_PARAM and rel_err(gb, gb_ref) <= TOL_PARAM
        assert rel_err(xk, np.fft.fft(x.astype(np.float64), axis=1)[:, :k]) <= TOL_ACT, on
        assert np.all(gwr[:, k:] == 0) and np.all(gwi[:, k:] == 0)          # unused columns exactly zero


@pytest.mark.parametrize("nsplit", [0, 3])
@pytest.mark.parametrize("B,N,D,F", [(4, 2000, 64, 100), (3, 1200, 40, 200)])
def test_sixteen_row_plan_phase_split_backward_and_dropout(gpu, B, N, D, F, nsplit):
    """Every phase split of smx_backward on a SMX_PATH_DECIM16 shape -- SPECTRUM, PARAMS, INVERSE separately (gradient
    sync "overlap"), SPECTRUM | INVERSE then PARAMS ("fused") -- gives the numbers of the single call: the products go
    to the slab in every case, the filtered spectrum is parked for k_inv16.  The fused dropout is served as well."""
    pkg, lib, fn = _pkg()
    assert lib.plan(B, N, D, F).path == lib.SMX_PATH_DECIM16
    torch.manual_seed(5)
    x = torch.randn(B, N, D, device=gpu); g = torch.randn(B, N, D, device=gpu)
    wr = torch.randn(D, F, device=gpu); wi = torch.randn(D, F, device=gpu)
    _, xk0 = fn.forward_raw(x, wr, wi, None, save_spectrum=True)
    gx0, flat0 = fn.backward_raw(g, xk0, wr, wi)
    if nsplit:
        ctx = lib.options(nsplit=nsplit)
        ctx.__enter__()
    try:
        _check_phase_splits(lib, fn, gpu, B, N, D, F, x, g, wr, wi, xk0, gx0, flat0, nsplit)
    finally:
        if nsplit:
            ctx.__exit__(None, None, None)


def _check_phase_splits(lib, fn, gpu, B, N, D, F, x, g, wr, wi, xk0, gx0, flat0, nsplit):
    if nsplit:
        assert lib.plan(B, N, D, F).nsplit == nsplit
    _, xk = fn.forward_raw(x, wr, wi, None, save_spectrum=True)
    # the tile-split plan against the single launch: same arithmetic per tile, another summation order across chunks
    assert rel_err(torch.view_as_real(xk).cpu().numpy(), torch.view_as_real(xk0).cpu().numpy()) <= TOL_ACT
    gx_all, flat_all = fn.backward_raw(g, xk, wr, wi)
    assert rel_err(gx_all.cpu().numpy(), gx0.cpu().numpy()) <= TOL_ACT
    assert rel_err(flat_all.cpu().numpy(), flat0.cpu().numpy()) <= TOL_PARAM
    ws = torch.zeros(fn._ws_bytes(B, N, D, F), dtype=torch.uint8, device=gpu)
    gx_s, flat_s = fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_SPECTRUM, ws=ws)
    fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_PARAMS, want_x=False, flat=flat_s, ws=ws)
    fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_INVERSE, grad_x=gx_s, flat=flat_s, ws=ws)
    gx_f, flat_f = fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_SPECTRUM | fn.PHASE_INVERSE, ws=ws)
    fn.backward_raw(g, xk, wr, wi, phases=fn.PHASE_PARAMS, want_x=False, flat=flat_f, ws=ws)
    torch.cuda.synchronize()
    for gx_, flat_ in ((gx_s, flat_s), (gx_f, flat_f)):
        assert torch.equal(gx_, gx_all) and torch.equal(flat_, flat_all)      # the same launches' arithmetic: bit-equal
    rng = fn.DropoutState(gpu).next()
    y_d, _ = fn.forward_raw(x, wr, wi, None, dropout_p=0.25, rng=rng)
    y_0, _ = fn.forward_raw(x, wr, wi, None)
    keep = y_d != 0
    assert 0.6 < keep.float().mean().item() < 0.9
    assert rel_err((y_d[keep] * 0.75).cpu().numpy(), y_0[keep].cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("B,N,D,F", [(4, 1000, 256, 128), (3, 3000, 34, 77), (2, 200, 512, 100), (9, 5000, 64, 17)])
def test_lengths_that_are_8_mod_16_run_as_the_even_bins_of_twice_the_length(gpu, B, N, D, F):
    """N = 1000, 3000, 5000 ...: the N-point bins are the even bins of the 2N-point transform of the zero-padded
    sequence, and 2N is a multiple of 16 -> k_fused16 with cropped rows instead of DFT products.  Same numbers as the
    oracle; every gradient in the caller's shape; grad columns >= k exactly zero."""
    pkg, lib, fn = _pkg()
    k = min(F, N // 2)
    sh = lib.smx_shape(B, N, D, 2 * k - 1, 2 * N, 2 * k - 1)
    assert lib.plan(B, N, D, F).path == lib.SMX_PATH_DIRECT and lib.plan_ex(sh).path == lib.SMX_PATH_DECIM16
    rng = np.random.default_rng(N + F)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    with torch.no_grad():
        layer.weight_real.copy_(T(wr)); layer.weight_imag.copy_(T(wi)); layer.bias.copy_(T(b))
    xd = T(x).to(gpu).requires_grad_(True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y = layer(xd)
    y.backward(T(g).to(gpu))
    torch.cuda.synchronize()
    y_ref, _ = so.forward_closed(x, wr, wi, b)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(x, wr, wi, g)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y_ref) <= TOL_ACT and rel_err(c(xd.grad), gx_ref) <= TOL_ACT
    assert rel_err(c(layer.weight_real.grad), gwr_ref) <= TOL_PARAM
    assert rel_err(c(layer.weight_imag.grad), gwi_ref) <= TOL_PARAM
    assert rel_err(c(layer.bias.grad), gb_ref) <= TOL_PARAM
    assert layer.weight_real.grad.shape == (D, F) and torch.all(layer.weight_real.grad[:, k:] == 0)


def test_scoped_options_follow_the_node_into_the_autograd_thread(gpu):
    """Autograd runs backward on its own thread; thread-local scoped options of the forward call would not be in force
    there: forward (one fused launch, small workspace) and backward (default plan: residue split, larger workspace)
    would disagree about the workspace forward hands over.  The Functions carry the forward's options along."""
    pkg, lib, fn = _pkg()
    B, N, D, F = 2, 8192, 64, 32
    assert lib.plan(B, N, D, F).nsplit > 1
    torch.manual_seed(1)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    x = torch.randn(B, N, D, device=gpu, requires_grad=True); g = torch.randn(B, N, D, device=gpu)
    y0 = layer(x); y0.backward(g)
    ref = (y0.detach().clone(), x.grad.clone(), layer.weight_real.grad.clone())
    x.grad = None; layer.zero_grad(set_to_none=True)
    with lib.options(nsplit=1):
        assert lib.plan(B, N, D, F).nsplit == 1 and lib.current_options()["nsplit"] == 1
        y1 = layer(x)
    y1.backward(g)                                   # outside the block, on the autograd thread
    torch.cuda.synchronize()
    assert lib.current_options() is None
    for a, b in zip((y1.detach(), x.grad, layer.weight_real.grad), ref):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= TOL_PARAM


def test_a_default_changed_between_forward_and_backward_does_not_reach_the_backward(gpu):
    """forward without a scoped push used to note "no options", so a set_option() before backward made the
    backward plan differently on the workspace / save layout of the forward (here: the residue split against one fused
    launch).  The node now carries a snapshot of the effective options."""
    pkg, lib, fn = _pkg()
    B, N, D, F = 2, 8192, 64, 32
    assert lib.plan(B, N, D, F).nsplit > 1
    torch.manual_seed(2)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    x = torch.randn(B, N, D, device=gpu, requires_grad=True); g = torch.randn(B, N, D, device=gpu)
    y0 = layer(x); y0.backward(g)
    ref = (x.grad.clone(), layer.weight_real.grad.clone())
    x.grad = None; layer.zero_grad(set_to_none=True)
    y1 = layer(x)
    lib.set_option("nsplit", 1)
    try:
        assert lib.plan(B, N, D, F).nsplit == 1
        y1.backward(g)
        torch.cuda.synchronize()
    finally:
        lib.set_option("nsplit", 0)
    for a, b in zip((x.grad, layer.weight_real.grad), ref):
        assert torch.equal(a, b)                      # the same plan ran: bit-identical


@pytest.mark.parametrize("B,N,D", [(4, 1000, 64), (3, 2048, 33), (2, 2000, 48)])
def test_block_line_at_lengths_and_widths_that_stream_through_the_python_routes(gpu, B, N, D):
    """SpectralMLPBlock's first line at N = 8 (odd) / odd D (composition around spectral_mix's routes) and at
    N = 16 P (native block call, sixteen-row kernels): against the oracle's port of the reference lines."""
    pkg, lib, fn = _pkg()
    torch.manual_seed(N + D)
    F = max(2, D // 2)
    x = torch.randn(B, N, D); g = torch.randn(B, N, D)
    lw = 1 + 0.3 * torch.randn(D); lb = 0.2 * torch.randn(D)
    wr = 1 + 0.5 * torch.randn(D, F); wi = 0.5 * torch.randn(D, F); bias = 0.1 * torch.randn(D)
    ref = so.block_half_port(x, lw, lb, 1e-5, wr, wi, bias, g)
    leaves = [t.to(gpu).requires_grad_(True) for t in (x, lw, lb, wr, wi, bias)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y = fn.spectral_block_mix(leaves[0], leaves[1], leaves[2], 1e-5, leaves[3], leaves[4], leaves[5])
    y.backward(g.to(gpu))
    got = [y.detach()] + [t.grad for t in leaves]
    floor = 1e-3 * float(ref[4].abs().max())
    for i, (a, r) in enumerate(zip(got, ref)):
        err = float((a.cpu() - r).abs().max()) / max(float(r.abs().max()), floor if i >= 2 else 1e-30)
        assert err <= (2e-5 if i < 2 else 1e-4), (i, err)


@pytest.mark.parametrize("B,R,D,n_fft,k", [(3, 300, 34, 400, 100), (2, 2000, 64, 2000, 250), (1, 120, 6, 240, 30)])
def test_pruned_rfft_of_zero_padded_rows_on_the_sixteen_row_plan(gpu, B, R, D, n_fft, k):
    """functional.rfft (smx_rfft_ex) with n_fft a multiple of 16: spectrum-only mode of k_fused16 / k_split16_a with
    zero-padded rows, and its gradient (synthesis: DFT products) -- against numpy / the adjoint identity."""
    pkg, lib, fn = _pkg()
    assert lib.plan_ex(lib.smx_shape(B, R, D, k, n_fft, k)).path == lib.SMX_PATH_DECIM16
    rng = np.random.default_rng(R + k)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    xd = T(x).to(gpu).requires_grad_(True)
    X = fn.rfft(xd, n_fft, k)
    ref = np.fft.rfft(x.astype(np.float64), n=n_fft, axis=1)[:, :k]
    assert rel_err(X.detach().cpu().numpy(), ref) <= TOL_ACT
    gS = torch.randn(B, k, D, dtype=torch.complex64, device=gpu)
    (torch.view_as_real(X) * torch.view_as_real(gS)).sum().backward()        # <X, gS> (real inner product)
    # adjoint of rfft: grad_x[n] = Re sum_f conj... = sum_f (gS.re cos + gS.im (-sin))... checked through numpy's matrices
    n = np.arange(R)[:, None]; f = np.arange(k)[None, :]
    Wc = np.cos(2 * np.pi * n * f / n_fft); Ws = -np.sin(2 * np.pi * n * f / n_fft)        # X = x (Wc + i Ws)
    g = gS.cpu().numpy().astype(np.complex128)
    gx_ref = np.einsum("nf,bfd->bnd", Wc, g.real) + np.einsum("nf,bfd->bnd", Ws, g.imag)
    assert rel_err(xd.grad.cpu().numpy(), gx_ref) <= TOL_ACT
