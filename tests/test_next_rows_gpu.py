"""SURVEY 8(f) rows on the GPU: the general fused transform (zero-padded rows, explicit bin count,
Nyquist bin) against the fp64 closed form, and the drop-in modules built on it against fixtures the
reference modules produced (FixedSpectralBlock, FrequencyConvFunc, PhaseAwareSpectralMixing,
MultiScaleSpectralFeatures, ComplexRoPESpectralLayer, fnet_attention).

Tolerances: max|delta| <= 1e-5 max|ref| for the transform itself (y, grad_x), 1e-4 for parameter
gradients; 2e-5 for whole blocks that add LayerNorm / Linear layers on torch around it.
"""
import numpy as np
import pytest
import torch

from conftest import TOL_ACT, TOL_PARAM, load_golden, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
TOL_BLOCK = 2e-5
T = torch.from_numpy


def _pkg():
    import tensor_cuda_fft_amd as pkg
    from tensor_cuda_fft_amd import _lib, functional
    return pkg, _lib, functional


# (B, rows, D, F, n_fft, k)   k = n_fft // 2 + 1 keeps the Nyquist bin
EX_SHAPES = [
    (2, 256, 8, 129, 256, 129),        # one band, self-paired Nyquist slot
    (2, 192, 32, 129, 256, 129),       # ... with zero-padded rows (F01's transform)
    (3, 512, 6, 257, 512, 257),        # two bands + Nyquist
    (2, 1024, 4, 513, 1024, 513),      # four bands + Nyquist (split plan: few workgroups)
    (16, 1024, 64, 513, 1024, 513),    # four bands + Nyquist, single fused launch
    (2, 768, 6, 385, 768, 385),        # L = 3: Nyquist is an ordinary +/- pair of the four-band kernel
    (1, 1024, 8, 1025, 2048, 1025),    # n_fft 2048 (four-step path, L = 8), padded rows (F03's transform)
    (2, 1100, 4, 700, 2048, 700),      # ... padded rows, 700 bins
    (4, 2048, 64, 1025, 2048, 1025),   # ... all rows, several workgroups
    (8, 1024, 90, 1025, 2048, 1025),   # ... ragged channel tile
    (2, 6144, 4, 3073, 6144, 3073),    # L = 24: four-step, radix-2 step over two 12-point products
    (2, 7000, 6, 2500, 7680, 2500),    # L = 30, padded rows, pruned
    (2, 4352, 4, 2177, 4352, 2177),    # L = 17: odd tile count in one thread's registers (round 3; band groups before)
    (2, 6400, 6, 3201, 6400, 3201),    # L = 25
    (1, 7900, 4, 3000, 7936, 3000),    # L = 31, padded rows, pruned
    (1, 12288, 4, 6145, 12288, 6145),  # L = 48 = 12 x 4: two-level columns with a 12-point first level (round 3; band groups before)
    (2, 9216, 6, 4609, 9216, 4609),    # L = 36 = 9 x 4: one thread of each column pair holds padding
    (2, 13000, 4, 5000, 13312, 5000),  # L = 52 = 13 x 4, padded rows, pruned
    (1, 20480, 34, 10241, 20480, 10241),  # L = 80 = 10 x 8, ragged channel tile
    (2, 30000, 2, 18433, 36864, 18433),  # L = 144 = 9 x 16, padded rows
    (1, 61440, 4, 30721, 61440, 30721),  # L = 240 = 15 x 16
    (1, 14336, 4, 7169, 14336, 7169),  # L = 56 = 14 x 4
    (1, 11264, 4, 5633, 11264, 5633),  # L = 44 = 11 x 4
    (1, 8704, 4, 4353, 8704, 4353),    # L = 34: still band groups (2 x 17)
    (2, 3072, 6, 1537, 3072, 1537),    # L = 12: four-step with the generic L-point product
    (3, 1500, 4, 897, 1792, 897),      # L = 7, padded rows
    (2, 3840, 2, 1000, 3840, 1000),    # L = 15, pruned to 1000 bins
    (4, 4096, 64, 2049, 4096, 2049),   # four-step path, L = 16, several workgroups
    (16, 1024, 8, 1025, 2048, 1025),   # four-step, slab summed over 8 batch groups of 2 rows
    (19, 3000, 4, 2049, 4096, 2049),   # ... ragged groups (7 groups of 3, last of 1)
    (2, 5000, 6, 4097, 8192, 4097),    # four-step path, L = 32, padded rows
    (1, 8192, 8, 3000, 8192, 3000),    # four-step path, pruned to 3000 bins
    (1, 1280, 6, 641, 1280, 641),      # L = 5: four-step, odd L (Nyquist at column 128)
    (2, 16384, 6, 8193, 16384, 8193),  # L = 64: two-level column transform
    (3, 20000, 4, 3000, 32768, 3000),  # L = 128, padded rows, pruned
    (1, 65536, 34, 32769, 65536, 32769),  # L = 256, ragged channel tile
    (2, 3000, 2, 2049, 4096, 2049),    # four groups + Nyquist edge bin, padded
    (2, 300, 16, 100, 512, 100),       # padded rows, pruned bins, two... one band
    (40, 600, 64, 60, 1024, 60),       # padded rows on the single fused launch (one band)
    (2, 100, 16, 65, 128, 65),         # direct plan (n_fft % 256 != 0), padded, Nyquist
    (2, 33, 6, 17, 33, 17),            # odd length: every bin >= 1 has a mirror image
    (2, 300, 9, 257, 512, 257),        # odd channel count -> direct plan, padded
    (1, 64, 4, 8, 64, 0),              # k = 0: y = bias
]


@pytest.mark.parametrize("B,R,D,F,n_fft,k", EX_SHAPES)
def test_spectral_filter_vs_closed_form(gpu, B, R, D, F, n_fft, k):
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(B * 7 + R + D + F + n_fft + k)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    xd = T(x).to(gpu).requires_grad_(True)
    wrd, wid, bd = (T(a).to(gpu).requires_grad_(True) for a in (wr, wi, b))
    y = fn.spectral_filter(xd, wrd, wid, bd, n_fft=n_fft, k=k)
    y.backward(T(g).to(gpu))
    torch.cuda.synchronize()
    y_ref, X_ref = so.forward_closed_ex(x, wr, wi, b, n_fft, k)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed_ex(x, wr, wi, g, n_fft, k)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y_ref) <= TOL_ACT
    assert rel_err(c(xd.grad), gx_ref) <= TOL_ACT
    assert rel_err(c(wrd.grad), gwr_ref) <= TOL_PARAM
    assert rel_err(c(wid.grad), gwi_ref) <= TOL_PARAM
    assert rel_err(c(bd.grad), gb_ref) <= TOL_PARAM
    assert not c(wrd.grad)[:, k:].any() and not c(wid.grad)[:, k:].any()      # unused columns exactly zero
    if k > 0:
        assert rel_err(c(fn.rfft_bins(xd, k, n_fft)), X_ref) <= TOL_ACT


@pytest.mark.parametrize("variant", ["nsplit2", "nsplit_max", "direct", "groups", "full8", "bgroups"])
@pytest.mark.parametrize("B,R,D,F,n_fft,k", [EX_SHAPES[1], EX_SHAPES[2], EX_SHAPES[3], EX_SHAPES[6], EX_SHAPES[7],
                                             (2, 300, 16, 100, 512, 100), (2, 3000, 2, 2049, 4096, 2049),
                                             (19, 1024, 4, 1025, 2048, 1025)])
def test_spectral_filter_kernel_variants(gpu, variant, B, R, D, F, n_fft, k):
    """The same general shapes through the residue-split launches, the direct plan and -- for n_fft 2048 --
    the band-group plan the eight-band kernel replaces."""
    pkg, lib, fn = _pkg()
    opts = {"nsplit2": ("nsplit", 2), "nsplit_max": ("nsplit", 1 << 20), "direct": ("force_direct", 1),
            "groups": ("full8", 0), "full8": ("fourstep", 0), "bgroups": ("fs_bgroups", 8)}[variant]
    if variant == "bgroups" and B < 16:
        pytest.skip("batch-grouped slab needs B >= 2 groups")
    if variant == "groups":
        lib.set_option("fourstep", 0)            # n_fft 2048 / 4096 through the band groups
    if variant == "direct" and n_fft > 2048:
        pytest.skip("direct plan is O(N k)")
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = rng.standard_normal((D, F)).astype(np.float32); wi = rng.standard_normal((D, F)).astype(np.float32)
    lib.set_option(*opts)
    try:
        xd = T(x).to(gpu).requires_grad_(True)
        wrd, wid = (T(a).to(gpu).requires_grad_(True) for a in (wr, wi))
        y = fn.spectral_filter(xd, wrd, wid, None, n_fft=n_fft, k=k)
        y.backward(T(g).to(gpu))
        torch.cuda.synchronize()
    finally:
        lib.set_option("nsplit", 0); lib.set_option("force_direct", 0); lib.set_option("full8", 1)
        lib.set_option("fourstep", 1); lib.set_option("fs_bgroups", 0)
    y_ref, _ = so.forward_closed_ex(x, wr, wi, None, n_fft, k)
    gx_ref, gwr_ref, gwi_ref, _ = so.backward_closed_ex(x, wr, wi, g, n_fft, k)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y_ref) <= TOL_ACT and rel_err(c(xd.grad), gx_ref) <= TOL_ACT
    assert rel_err(c(wrd.grad), gwr_ref) <= TOL_PARAM and rel_err(c(wid.grad), gwi_ref) <= TOL_PARAM


def test_layer_entry_points_are_the_special_case(gpu):
    """smx_forward == smx_forward_ex with rows = n_fft and k = min(F, n_fft/2): bit-identical."""
    pkg, lib, fn = _pkg()
    torch.manual_seed(1)
    x = torch.randn(4, 1024, 64, device=gpu)
    wr = torch.randn(64, 32, device=gpu); wi = torch.randn(64, 32, device=gpu); b = torch.randn(64, device=gpu)
    assert torch.equal(fn.spectral_mix(x, wr, wi, b), fn.spectral_filter(x, wr, wi, b))


def _load(mod, z, gpu):
    sd = {k[3:]: T(v) for k, v in z.items() if k.startswith("sd.")}
    assert set(sd) == set(mod.state_dict()), set(sd) ^ set(mod.state_dict())
    mod.load_state_dict(sd)
    return mod.to(gpu)


def _check_module(mod, z, gpu, fwd=None, tol=TOL_BLOCK):
    x = T(z["x"]).to(gpu).requires_grad_(True)
    y = fwd(mod, x) if fwd else mod(x)
    y.backward(T(z["g"]).to(gpu))
    torch.cuda.synchronize()
    c = lambda t: torch.view_as_real(t).cpu().numpy() if t.is_complex() else t.detach().cpu().numpy()
    assert rel_err(c(y.detach()), z["y"]) <= tol
    assert rel_err(c(x.grad), z["grad_x"]) <= tol
    for name, p in mod.named_parameters():
        ref = z["grad." + name]
        ref = np.stack([ref.real, ref.imag], -1) if np.iscomplexobj(ref) else ref
        assert rel_err(c(p.grad), ref) <= TOL_PARAM, name


@pytest.mark.parametrize("name", ["F01_fixed_2x192x32", "F02_fixed_2x512x16", "F03_fixed_1x1024x8",
                                  "F04_fixed_2x100x16", "F05_fixed_2x100x16", "F06_fixed_2x300x9",
                                  "F07_fixed_1x8000x4"])
@pytest.mark.parametrize("conv1", [1, 0, 2, 3])
def test_fixed_spectral_block_matches_reference(gpu, name, conv1):
    """fft_lm FixedSpectralBlock (reference train_fixed_full.py:427-563): the reference's state_dict loads
    unchanged; output, grad_x and every parameter gradient (kernel taps, gain, frequency gate logits,
    context gate, norms, FFN) match its CPU run, with and without the cutoff curriculum.
    conv1: the fixtures have at most two (batch row, 32-channel tile) items, so the default
    rule (1) runs them through the three-launch form; 0 = that form by request, 2 / 3 = the ONE-launch kernel
    k_conv1 wherever the shape allows it, on 512-thread (32 channels) / 256-thread (16 channels) workgroups -- every
    reference-held vector of this row goes through every form of the convolution."""
    pkg, lib, fn = _pkg()
    z = load_golden(name)
    B, R, C = z["x"].shape
    blk = pkg.FixedSpectralBlock(C, seq_len=int(z["seq_len"]), kernel_len=int(z["kernel_len"]),
                                 transition_bins=int(z["transition_bins"]), dropout=0.0)
    _load(blk, z, gpu)
    cutoff = None if int(z["cutoff"]) < 0 else int(z["cutoff"])
    n_fft = 1 << (R + int(z["kernel_len"]) - 2).bit_length()               # next_pow2(T + K - 1), reference :507-509
    with lib.options(conv1=conv1):
        if conv1 >= 2 and n_fft <= 2048 and fn.conv_supported(B, R, C, n_fft):
            # the one-launch form is what runs: it needs no tile-spectra scratch, so its workspace is smaller
            ws1, _ = fn._conv_plan(B, R, C, n_fft)
            with lib.options(conv1=0):
                ws0, _ = fn._conv_plan(B, R, C, n_fft)
            assert ws1 < ws0, (conv1, ws1, ws0)
        _check_module(blk, z, gpu, fwd=lambda m, x: m(x, cutoff=cutoff))


@pytest.mark.parametrize("name", ["T01_freqnative_2x192x16", "T02_freqnative_1x1024x8", "T03_freqnative_2x100x6",
                                  "T11_bicameral_2x192x16", "T12_bicameral_1x1024x8", "T13_bicameral_2x300x10"])
def test_spectrum_domain_twins_match_reference(gpu, name):
    """FrequencyNativeBlock (reference fft_lm/frequency_native.py:242-362) and BicameralBlock (reference
    fft_lm/bicameral.py:26-278) on the native transform pair (functional.rfft / irfft): the reference's state_dict
    loads unchanged; output, grad_x and every parameter gradient match its CPU run."""
    pkg, _, _ = _pkg()
    z = load_golden(name)
    C = z["x"].shape[2]
    cls = pkg.FrequencyNativeBlock if "freqnative" in name else pkg.BicameralBlock
    blk = cls(C, seq_len=int(z["seq_len"]), kernel_len=int(z["kernel_len"]),
              transition_bins=int(z["transition_bins"]), dropout=0.0)
    _load(blk, z, gpu)
    cutoff = None if int(z["cutoff"]) < 0 else int(z["cutoff"])
    _check_module(blk, z, gpu, fwd=lambda m, x: m(x, cutoff=cutoff))


def test_phase_shift_is_the_polar_form(gpu):
    """PhaseShift multiplies by m e^{i r}; the reference rebuilds |z| m e^{i (arg z + r)} (frequency_native.py:62-77)."""
    pkg, _, _ = _pkg()
    torch.manual_seed(3)
    ps = pkg.PhaseShift(12, 40).to(gpu)
    with torch.no_grad():
        ps.phase_weights.normal_(); ps.magnitude_logits.normal_()
    z = torch.complex(torch.randn(3, 33, 12), torch.randn(3, 33, 12)).to(gpu).requires_grad_(True)
    go = torch.complex(torch.randn(3, 33, 12), torch.randn(3, 33, 12)).to(gpu)
    out = ps(z)
    out.backward(go)
    zr = z.detach().cpu().to(torch.complex128).requires_grad_(True)
    pw = ps.phase_weights.detach().cpu().double().requires_grad_(True)
    ml = ps.magnitude_logits.detach().cpu().double().requires_grad_(True)
    ref = (zr.abs() * (1.0 + 0.1 * torch.tanh(ml[:33]))) * torch.exp(1j * (zr.angle() + torch.tanh(pw[:33]) * np.pi))
    ref.backward(go.cpu().to(torch.complex128))
    c = lambda t: torch.view_as_real(t).detach().cpu().numpy() if t.is_complex() else t.detach().cpu().numpy()
    assert rel_err(c(out), c(ref)) <= TOL_ACT and rel_err(c(z.grad), c(zr.grad)) <= TOL_ACT
    assert rel_err(c(ps.phase_weights.grad), c(pw.grad)) <= TOL_PARAM
    assert rel_err(c(ps.magnitude_logits.grad), c(ml.grad)) <= TOL_PARAM


def test_frequency_conv_func_matches_reference(gpu):
    pkg, _, _ = _pkg()
    z = load_golden("FC1_freqconv_2x33x8")
    x = T(z["x_freq"]).to(gpu).requires_grad_(True)
    kf = T(z["kernel_freq"]).to(gpu).requires_grad_(True)
    gain = T(z["gain"]).to(gpu).requires_grad_(True)
    out = pkg.FrequencyConvFunc.apply(x, kf, gain)
    out.backward(T(z["g"]).to(gpu))
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(out), z["out"]) <= TOL_ACT and rel_err(c(x.grad), z["grad_x"]) <= TOL_ACT
    assert rel_err(c(kf.grad), z["grad_kernel"]) <= TOL_PARAM
    assert rel_err(c(gain.grad), z["grad_gain"]) <= TOL_PARAM


@pytest.mark.parametrize("name", ["P01_phase_2x512x32", "P02_phase_2x33x6", "P03_phase_1x2048x4"])
def test_phase_aware_mixing_matches_reference(gpu, name):
    pkg, _, _ = _pkg()
    z = load_golden(name)
    m = _load(pkg.PhaseAwareSpectralMixing(z["x"].shape[2]), z, gpu)
    _check_module(m, z, gpu, tol=TOL_ACT)


@pytest.mark.parametrize("name", ["M01_multi_2x1024x16", "M02_multi_2x50x8"])
def test_multiscale_features_match_reference(gpu, name):
    pkg, _, _ = _pkg()
    z = load_golden(name)
    m = _load(pkg.MultiScaleSpectralFeatures(z["x"].shape[2]), z, gpu)
    _check_module(m, z, gpu)
    # the three bands themselves against the oracle's port of reference :237-262
    x = T(z["x"])
    for got, ref in zip(m.bands(x.to(gpu)), so.multiscale_bands_port(x)):
        assert rel_err(got.cpu().numpy(), ref.numpy()) <= TOL_ACT * max(1.0, float(x.abs().max() / ref.abs().max()))


@pytest.mark.parametrize("name", ["R01_rope_2x256x16", "R02_rope_2x40x8", "R03_rope_1x768x6"])
def test_complex_rope_layer_matches_reference(gpu, name):
    pkg, _, _ = _pkg()
    z = load_golden(name)
    m = _load(pkg.ComplexRoPESpectralLayer(z["x"].shape[2], dropout=0.0), z, gpu)
    _check_module(m, z, gpu)


@pytest.mark.parametrize("name", ["N01_fnet_2x256x8", "N02_fnet_2x30x5", "N03_fnet_1x1024x3", "N04_fnet_2x2048x5"])
def test_fnet_attention_matches_reference(gpu, name):
    pkg, _, _ = _pkg()
    z = load_golden(name)
    zz = T(z["z"]).to(gpu).requires_grad_(True)
    out = pkg.FrequencyAttention.fnet_attention(zz)
    out.backward(T(z["gz"]).to(gpu))
    assert rel_err(out.detach().cpu().numpy(), z["out"]) <= TOL_ACT
    assert rel_err(zz.grad.cpu().numpy(), z["grad_z"]) <= TOL_ACT


def test_rope_apply_to_fft(gpu):
    """ComplexRoPE.apply_to_fft (reference complex_rope.py:100-118) = ifft(rope(fft(x))).real."""
    pkg, _, _ = _pkg()
    rope = pkg.ComplexRoPE(8).to(gpu)
    x = torch.randn(2, 256, 8)
    ref = so.complex_rope_mix_port(x, rope.rotation.cpu(), torch.ones(8, dtype=torch.complex64))
    assert rel_err(rope.apply_to_fft(x.to(gpu)).cpu().numpy(), ref.numpy()) <= TOL_ACT


@pytest.mark.parametrize("B,T_,C,K", [(8, 1024, 512, 128), (64, 1024, 512, 128)])
def test_causal_conv_properties_at_the_reference_config(gpu, B, T_, C, K):
    """fft_lm's default lengths (TrainConfig: d_model 512, seq_len 1024, kernel_len 128 -> n_fft 2048) at
    the reference's batch 8 and at 64: size-independent properties of the native causal convolution --
    causality (no future leakage with the plain kernel), linearity, adjoint identity, a time-domain
    check of channel slices against the literal convolution sum."""
    pkg, _, fn = _pkg()
    torch.manual_seed(3)
    kern = (0.3 * torch.randn(K, device=gpu)).requires_grad_(True)
    gain = (1 + 0.3 * torch.randn(C, device=gpu)).requires_grad_(True)
    x = torch.randn(B, T_, C, device=gpu, requires_grad=True)
    y = pkg.causal_spectral_conv(x, kern, gain)
    g = torch.randn_like(y)
    y.backward(g)
    # literal causal convolution of a few channels: y[n] = gain * sum_m kernel[m] x[n - m]
    sl = slice(0, 3)
    xs = x.detach()[:2, :, sl].double().cpu(); ks = kern.detach().double().cpu()
    ref = torch.zeros_like(xs)
    for m in range(K):
        ref[:, m:] += ks[m] * xs[:, :T_ - m]
    ref = ref * gain.detach()[sl].double().cpu()
    assert rel_err(y.detach()[:2, :, sl].cpu().numpy(), ref.numpy()) <= TOL_ACT
    # causality: changing x from position t0 on leaves y[:, :t0] untouched (to fp32 noise of the transform)
    t0 = 700
    x2 = x.detach().clone(); x2[:, t0:] = torch.randn_like(x2[:, t0:])
    y2 = pkg.causal_spectral_conv(x2, kern.detach(), gain.detach())
    assert (y2[:, :t0] - y.detach()[:, :t0]).abs().max().item() <= 2e-5 * y.detach().abs().max().item()
    # adjoint identity <y, g> = <x, grad_x>
    lhs = (y.detach().double() * g.double()).sum().item()
    rhs = (x.detach().double() * x.grad.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * y.detach().double().norm().item() * g.double().norm().item()
    # gain gradient = sum_{b,n} g * y / gain
    gg = (g.double() * y.detach().double()).sum((0, 1)) / gain.detach().double()
    assert rel_err(gain.grad.cpu().numpy(), gg.cpu().numpy()) <= TOL_PARAM


def test_capture_before_prepare_is_refused_cleanly(gpu):
    """A shape whose twiddle tables are not on the device yet cannot be captured: the library says so
    (SMX_ERR_UNSUPPORTED) instead of invalidating the capture with a blocking copy; after one eager
    call -- or smx_prepare -- the same capture works."""
    pkg, lib, fn = _pkg()
    N = 11 * 256                                           # a length nothing else in the suite uses
    x = torch.randn(2, N, 8, device=gpu)
    wr = torch.randn(8, 4, device=gpu); wi = torch.randn(8, 4, device=gpu)
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        gr.capture_begin()
        rc = lib.lib().smx_forward(x.data_ptr(), wr.data_ptr(), wi.data_ptr(), None, y.data_ptr(), None,
                                   None, 0, 2, N, 8, 4, 0, s.cuda_stream)
        msg = lib.lib().smx_last_error()
        gr.capture_end()
    assert rc == -2 and b"smx_prepare" in msg
    torch.cuda.current_stream().wait_stream(s)
    assert lib.lib().smx_prepare(N) == 0
    ref = fn.spectral_mix(x, wr, wi)
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2):
        out = fn.spectral_mix(x, wr, wi)
    gr2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("B,R,D,F,n_fft,k", [
    (2, 192, 32, 129, 256, 129),       # one band (fused)
    (3, 512, 6, 257, 512, 257),        # two bands
    (16, 1024, 64, 513, 1024, 513),    # four bands, fused launch
    (2, 1024, 4, 513, 1024, 513),      # four bands, residue split (batched unpack)
    (8, 1024, 90, 1025, 2048, 1025),   # four-step L = 8
    (20, 1024, 6, 1025, 2048, 1025),   # four-step L = 8, batch-grouped slab
    (2, 5000, 6, 4097, 8192, 4097),    # four-step L = 32
    (2, 6144, 4, 3073, 6144, 3073),    # four-step L = 24
    (2, 12000, 6, 8193, 16384, 8193),  # two-level columns, L = 64: partial sums of 33 column blocks
    (2, 4352, 4, 2177, 4352, 2177),    # four-step L = 17 (odd tile count, round 3)
    (2, 8704, 4, 4353, 8704, 4353),    # band groups (L = 34): factor applied as a multiply of the output
    (2, 9216, 6, 4609, 9216, 4609),    # two-level columns L = 36 = 9 x 4: partial sums of 33 column blocks, padded threads
    (2, 100, 16, 65, 128, 65),         # direct plan: same fallback
])
def test_row_scale_and_its_gradient(gpu, B, R, D, F, n_fft, k):
    """W_eff[b,d,f] = W[d,f] row_scale[b,d]: output, grad_x, grad_W (factor included) and
    d/d row_scale = sum_n g y0 against the fp64 closed form."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(R + k + D)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)
    xd, wrd, wid, scd = (T(a).to(gpu).requires_grad_(True) for a in (x, wr, wi, sc))
    y = fn.spectral_filter(xd, wrd, wid, None, n_fft=n_fft, k=k, row_scale=scd)
    y.backward(T(g).to(gpu))
    torch.cuda.synchronize()
    y0, _ = so.forward_closed_ex(x, wr, wi, None, n_fft, k)
    gx_ref, gwr_ref, gwi_ref, _ = so.backward_closed_ex(x, wr, wi, g * sc[:, None, :], n_fft, k)
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), y0 * sc[:, None, :]) <= TOL_ACT
    assert rel_err(c(xd.grad), gx_ref) <= TOL_ACT
    assert rel_err(c(wrd.grad), gwr_ref) <= TOL_PARAM and rel_err(c(wid.grad), gwi_ref) <= TOL_PARAM
    assert rel_err(c(scd.grad), (g.astype(np.float64) * y0).sum(axis=1)) <= TOL_PARAM


@pytest.mark.parametrize("B,N,D", [(3, 1280, 7), (2, 4096, 40), (1, 8192, 3), (2, 3072, 5), (4, 1024, 16), (3, 512, 9),
                                   (2, 16384, 5), (1, 32768, 8), (1, 65536, 3), (2, 6144, 6), (1, 5632, 4),
                                   (2, 12288, 5), (1, 9216, 3), (1, 20480, 4), (1, 61440, 2)])
def test_complex_sequence_fft_four_step(gpu, B, N, D):
    """smx_cfft_ex: the packed spectrum of the four-step plan written straight out, against numpy."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(N + D)
    z = (rng.standard_normal((B, N, D)) + 1j * rng.standard_normal((B, N, D))).astype(np.complex64)
    out = fn.seq_fft_raw(T(z).to(gpu))
    assert fn._cfft_native[(B, N, 2 * D, N // 2 + 1, N, N // 2 + 1)]
    assert rel_err(out.cpu().numpy(), np.fft.fft(z.astype(np.complex128), axis=1)) <= TOL_ACT


@pytest.mark.parametrize("B,R,D,n_fft", [(2, 1024, 8, 2048), (3, 1000, 34, 2048), (1, 640, 2, 2048), (16, 1024, 96, 2048),
                                         (3, 512, 10, 1024), (2, 300, 66, 1024), (5, 256, 32, 512), (2, 100, 4, 512),
                                         # rows > n_fft / 2: folded onto the lower half first
                                         (3, 1500, 34, 2048), (2, 2048, 8, 2048), (2, 1025, 6, 2048), (2, 700, 6, 1024),
                                         (4, 449, 64, 512), (2, 512, 4, 512)])
def test_single_launch_rank_one_conv_against_the_three_launch_form(gpu, B, R, D, n_fft):
    """k_conv1 (one launch per direction: the n_fft-point spectrum as two half-length transforms by parity of the
    bin, reference train_fixed_full.py:515-555) takes these shapes by default; option conv1 = 0 runs the same call
    through k_fs_a / k_fs_conv / k_fs_b.  Both against float64 autograd of the op sequence, and against each other."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(7 * R + D)
    fb = n_fft // 2 + 1
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    hr = rng.standard_normal(fb).astype(np.float32); hi = rng.standard_normal(fb).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)

    def run(**opts):
        with lib.options(**opts):
            xd, hrd, hid, scd = (T(a).to(gpu).requires_grad_(True) for a in (x, hr, hi, sc))
            y = fn.rank_one_conv(xd, hrd, hid, scd, n_fft)
            y.backward(T(g).to(gpu))
            torch.cuda.synchronize()
            return [t.detach().cpu().numpy() for t in (y, xd.grad, scd.grad, hrd.grad, hid.grad)]

    new, old, new8 = run(conv1=2), run(conv1=0), run(conv1=3)      # 3: 256-thread workgroups on 16 channels
    xt, hrt, hit, sct = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, hr, hi, sc))
    X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, n_fft - R)), dim=1)
    yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=n_fft, dim=1)[:, :R] * sct[:, None, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    ref = [yr.detach().numpy(), xt.grad.numpy(), sct.grad.numpy(), hrt.grad.numpy(), hit.grad.numpy()]
    tols = [TOL_ACT, TOL_ACT, TOL_PARAM, TOL_PARAM, TOL_PARAM]
    for a, b, c8, r, tol in zip(new, old, new8, ref, tols):
        assert rel_err(a, r) <= tol and rel_err(b, r) <= tol and rel_err(a, b) <= tol
        assert rel_err(c8, r) <= tol and rel_err(c8, a) <= tol


@pytest.mark.parametrize("D,k,n_fft", [(256, 513, 1024), (6, 17, 33), (40, 100, 4096), (2, 1, 8), (33, 129, 256)])
def test_phase_filter_matches_float64_autograd(gpu, D, k, n_fft):
    """smx_phase_filter / _backward: W[d, f] = c_f m[d] exp(i p[d]) (reference spectral_enhancements.py:147-164 with
    irfft's Hermitian weights) and the gradients of m and p from a random grad_W."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(D + k)
    m = (1 + 0.3 * rng.standard_normal(D)).astype(np.float32); p = rng.standard_normal(D).astype(np.float32)
    gr = rng.standard_normal((D, k)).astype(np.float32); gi = rng.standard_normal((D, k)).astype(np.float32)
    md, pd_ = T(m).to(gpu).requires_grad_(True), T(p).to(gpu).requires_grad_(True)
    wr, wi = fn.phase_filter(md, pd_, k, n_fft)
    (wr * T(gr).to(gpu) + wi * T(gi).to(gpu)).sum().backward()
    mt = torch.tensor(m, dtype=torch.float64, requires_grad=True); pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    c = torch.full((k,), 2.0, dtype=torch.float64); c[0] = 1.0
    if n_fft % 2 == 0 and k > n_fft // 2:
        c[n_fft // 2] = 1.0
    wr_ref = (mt * torch.cos(pt)).unsqueeze(1) * c.unsqueeze(0); wi_ref = (mt * torch.sin(pt)).unsqueeze(1) * c.unsqueeze(0)
    (wr_ref * torch.tensor(gr, dtype=torch.float64) + wi_ref * torch.tensor(gi, dtype=torch.float64)).sum().backward()
    cc = lambda t: t.detach().cpu().numpy()
    assert rel_err(cc(wr), wr_ref.detach().numpy()) <= TOL_ACT and rel_err(cc(wi), wi_ref.detach().numpy()) <= TOL_ACT
    assert rel_err(cc(md.grad), mt.grad.numpy()) <= TOL_PARAM and rel_err(cc(pd_.grad), pt.grad.numpy()) <= TOL_PARAM


@pytest.mark.parametrize("n_fft,K,nl,use_mask", [(2048, 128, 1025, True), (2048, 128, 1400, False), (512, 64, 257, True),
                                                 (8192, 128, 4097, True), (1000, 7, 501, False), (256, 256, 129, True)])
def test_conv_response_matches_float64_autograd(gpu, n_fft, K, nl, use_mask):
    """smx_conv_response / _backward: H = rfft(zero-pad(kernel), n_fft) * sigmoid(logits[:fb]) * mask (reference
    train_fixed_full.py:511-513, :529, :540-551) and the gradients of the taps and the logits."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(n_fft + K)
    fb = n_fft // 2 + 1
    k = (0.3 * rng.standard_normal(K)).astype(np.float32)
    lg = rng.standard_normal(nl).astype(np.float32)
    m = rng.random(fb).astype(np.float32) if use_mask else None
    gr = rng.standard_normal(fb).astype(np.float32); gi = rng.standard_normal(fb).astype(np.float32)
    kd, ld = T(k).to(gpu).requires_grad_(True), T(lg).to(gpu).requires_grad_(True)
    md = None if m is None else T(m).to(gpu)
    hr, hi = fn.conv_response(kd, ld, md, n_fft)
    (hr * T(gr).to(gpu) + hi * T(gi).to(gpu)).sum().backward()
    kt = torch.tensor(k, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(lg, dtype=torch.float64, requires_grad=True)
    H = torch.fft.rfft(torch.nn.functional.pad(kt, (0, n_fft - K))) * torch.sigmoid(lt[:fb])
    if m is not None:
        H = H * torch.tensor(m, dtype=torch.float64)
    (H.real * torch.tensor(gr, dtype=torch.float64) + H.imag * torch.tensor(gi, dtype=torch.float64)).sum().backward()
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(hr), H.real.detach().numpy()) <= TOL_ACT and rel_err(c(hi), H.imag.detach().numpy()) <= TOL_ACT
    assert rel_err(c(kd.grad), kt.grad.numpy()) <= TOL_PARAM
    assert rel_err(c(ld.grad), lt.grad.numpy()) <= TOL_PARAM
    assert np.all(c(ld.grad)[fb:] == 0)
    # no gate, no mask; only the taps need a gradient
    k2 = T(k).to(gpu).requires_grad_(True)
    hr2, hi2 = fn.conv_response(k2, None, None, n_fft)
    (hr2 * T(gr).to(gpu)).sum().backward()
    kt2 = torch.tensor(k, dtype=torch.float64, requires_grad=True)
    (torch.fft.rfft(torch.nn.functional.pad(kt2, (0, n_fft - K))).real * torch.tensor(gr, dtype=torch.float64)).sum().backward()
    assert rel_err(c(k2.grad), kt2.grad.numpy()) <= TOL_PARAM


def test_single_launch_rank_one_conv_inference_and_partial_gradients(gpu):
    """x_spectra = NULL (no backward follows), no row scale, and backward calls that ask for a subset of the
    parameter gradients."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(11)
    B, R, D, n_fft = 3, 1024, 40, 2048
    fb = n_fft // 2 + 1
    x = T(rng.standard_normal((B, R, D)).astype(np.float32)).to(gpu)
    g = T(rng.standard_normal((B, R, D)).astype(np.float32)).to(gpu)
    hr = T(rng.standard_normal(fb).astype(np.float32)).to(gpu)
    hi = T(rng.standard_normal(fb).astype(np.float32)).to(gpu)
    sc = T((0.5 + rng.random((B, D))).astype(np.float32)).to(gpu)
    with lib.options(conv1=2):
        with torch.no_grad():
            y_inf = fn.rank_one_conv(x, hr, hi, sc, n_fft)
            y_ns = fn.rank_one_conv(x, hr, hi, None, n_fft)
        xd = x.clone().requires_grad_(True)
        y = fn.rank_one_conv(xd, hr, hi, sc, n_fft)          # only grad_x
        y.backward(g)
        assert torch.equal(y, y_inf)
        assert rel_err(y_ns.cpu().numpy() * sc.cpu().numpy()[:, None, :], y_inf.cpu().numpy()) <= TOL_ACT
        xe, hre, hie, sce = (t.clone().requires_grad_(True) for t in (x, hr, hi, sc))
        fn.rank_one_conv(xe, hre, hie, sce, n_fft).backward(g)
        assert torch.equal(xd.grad, xe.grad)
        hrf = hr.clone().requires_grad_(True)                # only grad_h (x frozen)
        fn.rank_one_conv(x, hrf, hi, sc, n_fft).backward(g)
        assert torch.equal(hrf.grad, hre.grad)


@pytest.mark.parametrize("B", [64, 8])
def test_single_launch_rank_one_conv_at_the_fft_lm_default_size(gpu, B):
    """(B, 1024, 512) in n_fft 2048 -- fft_lm's default block (reference train_fixed_full.py:497-563; its trainer's
    default batch is 8, :51): the one-launch form the plan picks (512-thread workgroups at B = 64, 256-thread ones at
    B = 8) against the three-launch form of the same library, all outputs and gradients."""
    pkg, lib, fn = _pkg()
    gen = torch.Generator(device="cpu").manual_seed(5)
    R, D, n_fft = 1024, 512, 2048
    fb = n_fft // 2 + 1
    x = torch.randn(B, R, D, generator=gen).to(gpu)
    g = torch.randn(B, R, D, generator=gen).to(gpu)
    hr = torch.randn(fb, generator=gen).to(gpu); hi = torch.randn(fb, generator=gen).to(gpu)
    sc = (0.5 + torch.rand(B, D, generator=gen)).to(gpu)

    def run(**opts):
        with lib.options(**opts):
            xd, hrd, hid, scd = (t.clone().requires_grad_(True) for t in (x, hr, hi, sc))
            y = fn.rank_one_conv(xd, hrd, hid, scd, n_fft)
            y.backward(g)
            torch.cuda.synchronize()
            return [t.detach() for t in (y, xd.grad, scd.grad, hrd.grad, hid.grad)]

    new, old = run(conv1=1), run(conv1=0)
    for a, b, tol in zip(new, old, [TOL_ACT, TOL_ACT, TOL_PARAM, TOL_PARAM, TOL_PARAM]):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    # adjoint identity on the one-launch form: <y / s, g> = <x, grad_x / s> (both linear maps share H)
    y0 = (new[0] / sc[:, None, :]).double(); gx0 = (new[1] / sc[:, None, :]).double()
    lhs = float((y0 * g.double()).sum()); rhs = float((x.double() * gx0).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-3


@pytest.mark.parametrize("B,R,D,n_fft", [(2, 1024, 8, 2048), (3, 1500, 34, 2048), (16, 1024, 64, 2048),
                                         (2, 2100, 6, 4096), (4, 4096, 32, 4096), (3, 512, 10, 1024),
                                         (40, 449, 64, 512), (2, 300, 4, 512),
                                         (2, 4096, 34, 8192), (3, 5000, 8, 8192),       # two-level columns, L = 32
                                         (2, 16000, 6, 16384), (1, 20000, 4, 32768), (1, 65536, 2, 65536)])
def test_rank_one_conv_vs_autograd_of_the_reference_sequence(gpu, B, R, D, n_fft):
    """smx_conv_forward / backward (packed spectrum x Hermitian extension of H, scale at the store) against
    float64 autograd of rfft -> * H -> irfft -> crop -> * s (reference train_fixed_full.py:515-555)."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(R + D)
    fb = n_fft // 2 + 1
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    hr = rng.standard_normal(fb).astype(np.float32); hi = rng.standard_normal(fb).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)
    assert fn.conv_supported(B, R, D, n_fft)
    xd, hrd, hid, scd = (T(a).to(gpu).requires_grad_(True) for a in (x, hr, hi, sc))
    y = fn.rank_one_conv(xd, hrd, hid, scd, n_fft)
    y.backward(T(g).to(gpu))
    torch.cuda.synchronize()
    xt, hrt, hit, sct = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, hr, hi, sc))
    X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, n_fft - R)), dim=1)
    yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=n_fft, dim=1)[:, :R] * sct[:, None, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), yr.detach().numpy()) <= TOL_ACT
    assert rel_err(c(xd.grad), xt.grad.numpy()) <= TOL_ACT
    assert rel_err(c(scd.grad), sct.grad.numpy()) <= TOL_PARAM
    assert rel_err(c(hrd.grad), hrt.grad.numpy()) <= TOL_PARAM
    assert rel_err(c(hid.grad), hit.grad.numpy()) <= TOL_PARAM


@pytest.mark.parametrize("B,T,C,use_scale,use_bias", [(2, 192, 16, True, True), (3, 100, 10, True, True), (1, 1, 4, True, True),
                                                      (2, 2, 6, False, True), (2, 3, 7, True, False), (4, 65, 33, True, True),
                                                      (8, 1024, 512, True, True), (2, 33, 260, False, False)])
def test_time_path_conv_matches_the_reference_op_sequence(gpu, B, T, C, use_scale, use_bias):
    """BicameralBlock's time path (reference fft_lm/bicameral.py:214-227): transpose, shift right by one dropping the
    last position, depthwise Conv1d(kernel 3, padding 1, groups C), transpose back, time gate -- as one native launch
    on (B, T, C) (smx_dwconv3_*), against float64 autograd of exactly that op sequence: output, grad_x and the
    gradients of the taps, the bias and the gate, incl. T = 1, 2, 3 and channel counts off the vector path."""
    import torch.nn.functional as F
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(100 * T + C)
    x = rng.standard_normal((B, T, C)); g = rng.standard_normal((B, T, C))
    w = rng.standard_normal((C, 1, 3)); bias = rng.standard_normal(C) if use_bias else None
    sc = 0.5 + rng.random((B, C)) if use_scale else None
    t64 = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    xr, wr, br, sr = t64(x), t64(w), t64(bias), t64(sc)
    xc = xr.transpose(1, 2)
    y_ref = F.conv1d(F.pad(xc[:, :, :-1], (1, 0)), wr, br, padding=1, groups=C).transpose(1, 2)
    if sr is not None:
        y_ref = y_ref * sr.unsqueeze(1)
    y_ref.backward(torch.tensor(g, dtype=torch.float64))
    t32 = lambda a: None if a is None else torch.tensor(a, dtype=torch.float32, device=gpu, requires_grad=True)
    xd, wd, bd, sd = t32(x), t32(w), t32(bias), t32(sc)
    outs = []
    for _ in range(2):                                       # twice: fixed-order sums, bit-identical
        for t in (xd, wd, bd, sd):
            if t is not None:
                t.grad = None
        y = fn.causal_dwconv3(xd, wd, bd, sd)
        y.backward(torch.tensor(g, dtype=torch.float32, device=gpu))
        torch.cuda.synchronize()
        outs.append([y.detach().clone()] + [None if t is None else t.grad.clone() for t in (xd, wd, bd, sd)])
    for a, b in zip(*outs):
        assert (a is None and b is None) or torch.equal(a, b)
    refs = [y_ref.detach(), xr.grad, wr.grad, None if br is None else br.grad, None if sr is None else sr.grad]
    for i, (a, r) in enumerate(zip(outs[0], refs)):
        if r is None:
            assert a is None
            continue
        assert rel_err(a.cpu().numpy().reshape(r.shape), r.numpy()) <= (TOL_ACT if i < 2 else TOL_PARAM), i


@pytest.mark.parametrize("B,Fq,C", [(2, 97, 16), (3, 33, 6), (1, 5, 70), (4, 129, 512), (2, 9, 1000), (5, 1, 130)])
def test_spectral_layer_norm_matches_the_reference_op_sequence(gpu, B, Fq, C):
    """SpectralLayerNorm (reference fft_lm/frequency_native.py:203-239: abs, mean, biased var, rsqrt, gamma / beta rows
    of the bin, angle, exp(i angle)) as one native launch each way (smx_spectral_ln_*), against complex128 autograd of
    exactly that op sequence: output, grad_z, grad_gamma, grad_beta; zeros of both signs keep the reference's phases
    (angle(-0 + 0i) = pi) and get a zero gradient; the sums over the batch are bit-identical run to run."""
    pkg, lib, fn = _pkg()
    rng = np.random.default_rng(7 * Fq + C)
    z = (rng.standard_normal((B, Fq, C)) + 1j * rng.standard_normal((B, Fq, C))).astype(np.complex64)
    z[0, 0, :3] = [0.0, complex(-0.0, 0.0), complex(0.0, -0.0)]          # masked bins (a cutoff zeroes whole rows)
    if Fq > 2:
        z[:, 2, :] = 0
    g = (rng.standard_normal((B, Fq, C)) + 1j * rng.standard_normal((B, Fq, C))).astype(np.complex64)
    gamma = (1 + 0.3 * rng.standard_normal((Fq, C))).astype(np.float32)
    beta = (0.2 * rng.standard_normal((Fq, C))).astype(np.float32)
    eps = 1e-5
    zr = torch.tensor(z, dtype=torch.complex128, requires_grad=True)
    gr, br = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (gamma, beta))
    mag = zr.abs()
    mean = mag.mean(dim=-1, keepdim=True); var = mag.var(dim=-1, keepdim=True, unbiased=False)
    scaled = (mag - mean) * torch.rsqrt(var + eps) * gr + br
    ang = zr.angle()
    ref = scaled * torch.complex(torch.cos(ang), torch.sin(ang))
    ref.backward(torch.tensor(g, dtype=torch.complex128))
    zd = torch.tensor(z, device=gpu, requires_grad=True)
    gd, bd = (torch.tensor(a, device=gpu, requires_grad=True) for a in (gamma, beta))
    outs = []
    for _ in range(2):
        zd.grad = gd.grad = bd.grad = None
        out = fn.spectral_layer_norm(zd, gd, bd, eps)
        out.backward(torch.tensor(g, device=gpu))
        torch.cuda.synchronize()
        outs.append([out.detach().clone(), zd.grad.clone(), gd.grad.clone(), bd.grad.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    c = lambda t: torch.view_as_real(t).cpu().numpy() if t.is_complex() else t.cpu().numpy()
    refs = [ref.detach(), zr.grad, gr.grad, br.grad]
    for i, (a, r) in enumerate(zip(outs[0], refs)):
        assert rel_err(c(a), c(r)) <= (TOL_ACT if i < 2 else TOL_PARAM), i
    assert float(outs[0][1][0, 0, :3].abs().max()) == 0.0                # zero gradient at the zeros


@pytest.mark.parametrize("B,Fq,C,p_drop", [(2, 65, 16, 0.0), (3, 33, 24, 0.0), (2, 129, 130, 0.0)])
def test_planar_feed_forward_equals_the_interleaved_one(gpu, B, Fq, C, p_drop):
    """SpectralFFN.residual (the (2, B, F, C)-plane route: SpectralLayerNorm writing planes, both nn.Linear on contiguous
    rows, PhaseShift as smx_planar_cmul, the residual folding the planes back: reference fft_lm/frequency_native.py
    :167-189, :355-356) against x + SpectralFFN.forward(x) (complex tensors, (de)interleaving copies): output, grad_x and
    the gradient of every parameter."""
    pkg, lib, fn = _pkg()
    torch.manual_seed(Fq + C)
    ffn = pkg.SpectralFFN(C, Fq + 3, expansion=2, dropout=p_drop).to(gpu)
    with torch.no_grad():
        for prm in ffn.parameters():
            prm.add_(0.3 * torch.randn_like(prm))
    x = torch.randn(B, Fq, C, dtype=torch.complex64, device=gpu)
    g = torch.randn(B, Fq, C, dtype=torch.complex64, device=gpu)
    res = []
    for route in ("planar", "interleaved"):
        xr = x.clone().requires_grad_(True)
        ffn.zero_grad(set_to_none=True)
        y = ffn.residual(xr) if route == "planar" else xr + ffn(xr)
        y.backward(g)
        torch.cuda.synchronize()
        res.append([y.detach(), xr.grad] + [prm.grad.clone() for prm in ffn.parameters()])
    c = lambda t: torch.view_as_real(t).cpu().numpy() if t.is_complex() else t.cpu().numpy()
    names = ["y", "grad_x"] + [n for n, _ in ffn.named_parameters()]
    for n, a, b in zip(names, *res):
        assert rel_err(c(a), c(b)) <= (TOL_ACT if n in ("y", "grad_x") else TOL_PARAM), n



@pytest.mark.parametrize("B,Fq,C,parts,quirk", [
    (2, 65, 16, "upqm", True), (3, 33, 24, "upq", True), (2, 129, 130, "upqm", False), (4, 17, 256, "q", False),
    (1, 9, 2, "um", True), (5, 257, 66, "", False), (2, 40, 300, "pm", False)])
def test_spectral_gate_matches_the_reference_sequence(gpu, B, Fq, C, parts, quirk):
    """smx_spectral_gate_* against fp64 torch of the lines it replaces (reference fft_lm/frequency_native.py:95, :338,
    :351): forward, grad_x and every parameter gradient -- with `quirk` the gain takes the gradient
    FrequencyConvFunc.backward writes by hand (:115, no conjugate; the oracle's freqconv_port holds those lines), else
    the autograd one."""
    pkg, lib, fn = _pkg()
    torch.manual_seed(B * 1000 + Fq + C)
    x = torch.randn(B, Fq, C, dtype=torch.complex64, device=gpu)
    g = torch.randn(B, Fq, C, dtype=torch.complex64, device=gpu)
    a = torch.randn(Fq, dtype=torch.complex64, device=gpu)
    u = (1 + 0.3 * torch.randn(C, device=gpu)) if "u" in parts else None
    p = torch.sigmoid(torch.randn(Fq, device=gpu)) if "p" in parts else None
    q = torch.sigmoid(torch.randn(B, C, device=gpu)) if "q" in parts else None
    m = None
    if "m" in parts:
        m = torch.ones(Fq, device=gpu)
        m[Fq // 2:] = torch.linspace(1, 0, Fq - Fq // 2, device=gpu)
        m[-max(1, Fq // 8):] = 0.0
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (x, a, u, p, q)]
    y = fn.spectral_gate(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], m, reference_gain_grad=quirk)
    y.backward(g)
    torch.cuda.synchronize()
    got = [y.detach()] + [t.grad if t is not None else None for t in leaves]

    d = lambda t: None if t is None else (t.detach().to(torch.complex128 if t.is_complex() else torch.float64).cpu())
    X, A, U, P, Q, M, G = (d(t) for t in (x, a, u, p, q, m, g))
    class Conv(torch.autograd.Function):                 # FrequencyConvFunc with the reference's own backward
        @staticmethod
        def forward(ctx, xf, kf, gain):
            ctx.save_for_backward(xf, kf, gain)
            return so.freqconv_port(xf, kf, gain, xf)[0]
        @staticmethod
        def backward(ctx, go):
            return so.freqconv_port(*ctx.saved_tensors, go)[1:]
    rl = [t.clone().requires_grad_(True) if t is not None else None for t in (X, A, U, P, Q)]
    if quirk:
        assert rl[2] is not None
        r = Conv.apply(rl[0], rl[1], rl[2])
    else:
        r = rl[0] * rl[1].view(1, -1, 1)
        if rl[2] is not None:
            r = r * rl[2].view(1, 1, -1)
    if rl[3] is not None:
        r = r * rl[3].view(1, -1, 1)
    if rl[4] is not None:
        r = r * rl[4].unsqueeze(1)
    if M is not None:
        r = r * M.view(1, -1, 1)
    r.backward(G)
    ref = [r.detach()] + [t.grad if t is not None else None for t in rl]
    c = lambda t: torch.view_as_real(t).cpu().numpy() if t.is_complex() else t.cpu().numpy()
    for n, ga, rf in zip(("y", "grad_x", "grad_a", "grad_u", "grad_p", "grad_q"), got, ref):
        assert (ga is None) == (rf is None), n
        if ga is not None:
            assert rel_err(c(ga), c(rf)) <= (TOL_ACT if n in ("y", "grad_x") else TOL_PARAM), n
    if m is not None:
        # a masked bin is an exact zero whose signs are the ones torch's complex-times-real product leaves (the real factor
        # promoted to complex): SpectralLayerNorm's arg() reads them (reference :223, :236)
        t = x * a.view(1, -1, 1)
        for f_ in (u.view(1, 1, -1) if u is not None else None, p.view(1, -1, 1) if p is not None else None,
                   q.unsqueeze(1) if q is not None else None, m.view(1, -1, 1)):
            if f_ is not None:
                t = t * f_
        dead = (m == 0).nonzero().flatten()
        assert dead.numel() > 0 and float(got[0][:, dead].abs().max()) == 0.0
        for part in ("real", "imag"):
            assert torch.equal(torch.signbit(getattr(got[0][:, dead], part)), torch.signbit(getattr(t[:, dead], part))), part


def test_spectral_gate_is_bitwise_reproducible_and_refuses_odd_channels(gpu):
    pkg, lib, fn = _pkg()
    torch.manual_seed(5)
    B, Fq, C = 3, 100, 200
    x = torch.randn(B, Fq, C, dtype=torch.complex64, device=gpu)
    g = torch.randn_like(x)
    a = torch.randn(Fq, dtype=torch.complex64, device=gpu)
    u, p, q = torch.randn(C, device=gpu), torch.rand(Fq, device=gpu), torch.rand(B, C, device=gpu)
    runs = []
    for _ in range(2):
        lv = [t.clone().requires_grad_(True) for t in (x, a, u, p, q)]
        fn.spectral_gate(*lv, None, reference_gain_grad=True).backward(g)
        torch.cuda.synchronize()
        runs.append([t.grad.clone() for t in lv])
    for r0, r1 in zip(*runs):
        assert torch.equal(torch.view_as_real(r0) if r0.is_complex() else r0, torch.view_as_real(r1) if r1.is_complex() else r1)
    with pytest.raises(ValueError, match="even channel count"):
        fn.spectral_gate(x[:, :, :199].contiguous(), a)


@pytest.mark.parametrize("shape,with_c", [((2, 64, 16), True), ((3, 33, 24), True), ((1, 1000, 4), False), ((64, 128, 96), True)])
def test_mix_paths_matches_the_reference_line(gpu, shape, with_c):
    """smx_mix_* against fp64 torch of BicameralBlock's fusion line (reference fft_lm/bicameral.py:261-268):
    out = residual + w_f y_spectral + w_t y_time + 0.1 y_cross, every gradient, sums bitwise reproducible."""
    pkg, lib, fn = _pkg()
    torch.manual_seed(sum(shape))
    ts = [torch.randn(*shape, device=gpu) for _ in range(4)]
    if not with_c:
        ts[3] = None
    w = torch.rand(2, device=gpu)
    g = torch.randn(*shape, device=gpu)
    runs = []
    for _ in range(2):
        lv = [t.clone().requires_grad_(True) if t is not None else None for t in ts + [w]]
        out = fn.mix_paths(lv[0], lv[1], lv[2], lv[3], lv[4], 0.1)
        out.backward(g)
        torch.cuda.synchronize()
        runs.append([out.detach()] + [t.grad.clone() if t is not None else None for t in lv])
    for r0, r1 in zip(*runs):
        assert (r0 is None and r1 is None) or torch.equal(r0, r1)
    rl = [t.detach().double().cpu().requires_grad_(True) if t is not None else None for t in ts + [w]]
    ref = rl[0] + rl[4][0] * rl[1] + rl[4][1] * rl[2]
    if with_c:
        ref = ref + 0.1 * rl[3]
    ref.backward(g.double().cpu())
    refs = [ref.detach()] + [t.grad if t is not None else None for t in rl]
    for n, a, b in zip(("out", "grad_r", "grad_a", "grad_b", "grad_c", "grad_w"), runs[0], refs):
        assert (a is None) == (b is None), n
        if a is not None:
            assert rel_err(a.cpu().numpy(), b.numpy()) <= (TOL_PARAM if n == "grad_w" else TOL_ACT), n
