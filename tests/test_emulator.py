"""CPU emulation of the fused decimated kernel (tests/emu) against the reference's golden vectors.

The emulator executes the very phase functions the GPU kernel is built from
(tensor-cuda-fft-_amd/csrc/smx_core.h) one 'thread' at a time, so index maps, twiddle
conventions, the two-band accumulators, the Hermitian unpack and the ragged-D masking are all
checked here without a GPU.  What it cannot see (barriers, alignment, launch geometry) is
covered by the -m gpu tests.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, TOL_ACT, TOL_PARAM, golden_names, load_golden, rel_err
from oracle import spectral_oracle as so

EMU_DIR = os.path.join(ROOT, "tests", "emu")
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["bash", os.path.join(EMU_DIR, "build.sh")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(EMU_DIR, "libsmx_emu.so"))
    lib.emu_fused.restype = ctypes.c_int
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(FP)


def run_emu(lib, mode, xin, wr, wi, bias, xk, conj, stagger):
    B, N, D = xin.shape
    F = wr.shape[1]
    k = so.num_bins(N, F)
    y = np.zeros((B, N, D), np.float32)
    if xk is None:
        xk = np.zeros((B, k, D, 2), np.float32)
    ps = np.zeros((B, k, D, 2), np.float32)
    gb = np.zeros((B, D), np.float32)
    rc = lib.emu_fused(mode, _p(xin), _p(wr), _p(wi), _p(bias), _p(y), _p(xk), _p(ps), _p(gb),
                       B, N, D, F, conj, stagger)
    assert rc == 0
    return y, xk, ps, gb


def fast_path(z):
    B, N, D = z["x"].shape
    return N % 256 == 0 and D % 2 == 0 and so.num_bins(N, int(z["num_filters"])) <= 512


CASES = [n for n in golden_names("layer") if "nolearn" not in n and fast_path(load_golden(n))]


def test_case_list_covers_both_band_counts_and_tails():
    assert any("G09" in n for n in CASES) and any("G15" in n for n in CASES)
    assert any("G14" in n for n in CASES) and any("G16" in n for n in CASES)
    assert any("G17" in n for n in CASES) and any("G18" in n for n in CASES)      # four bands


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("stagger", [0, 1])
def test_emulated_kernel_matches_reference(emu, name, stagger):
    z = load_golden(name)
    if stagger and z["x"].shape[1] > 8192:
        pytest.skip("one pass over the long fixture is enough")
    x, g = z["x"], z["g"]
    wr, wi, b = z["weight_real"], z["weight_imag"], z["bias"]
    B, N, D = x.shape
    F = wr.shape[1]
    k = so.num_bins(N, F)
    y, xk, _, _ = run_emu(emu, 0, x, wr, wi, b, None, 0, stagger)
    gx, _, ps, gb = run_emu(emu, 1, g, wr, wi, None, xk, 1, stagger)
    P = (ps[..., 0] + 1j * ps[..., 1]).sum(0)
    gwr = np.zeros((D, F)); gwi = np.zeros((D, F))
    gwr[:, :k] = P.real.T; gwi[:, :k] = -P.imag.T
    assert rel_err(y, z["y"]) <= TOL_ACT
    assert rel_err(gx, z["grad_x"]) <= TOL_ACT
    assert rel_err(gwr, z["grad_w_real"]) <= TOL_PARAM
    assert rel_err(gwi, z["grad_w_imag"]) <= TOL_PARAM
    assert rel_err(gb.sum(0), z["grad_bias"]) <= TOL_PARAM
    # the saved spectrum is fft(x)[:, :k]
    X = np.fft.rfft(x.astype(np.float64), axis=1)[:, :k]
    assert rel_err(xk[..., 0] + 1j * xk[..., 1], X) <= TOL_ACT


def test_emulator_rejects_shapes_outside_the_decimated_path(emu):
    x = np.zeros((1, 100, 4), np.float32)
    w = np.zeros((4, 2), np.float32)
    assert emu.emu_fused(0, _p(x), _p(w), _p(w), None, _p(x), None, None, None, 1, 100, 4, 2, 0, 0) == -2


def test_dropout_mask_function_statistics(emu):
    """The counter-based mask the kernels regenerate in forward and backward (smx_core.h drop_hash):
    right keep rate, no visible correlation between neighbours, rows, batch rows or calls."""
    n = 1 << 20
    thr = round(0.1 * 65536)

    def mask(seed, counter, b):
        out = np.zeros(n, np.uint8)
        emu.emu_drop_mask(ctypes.c_ulonglong(seed), ctypes.c_ulonglong(counter), b, ctypes.c_longlong(n),
                          thr, out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        return out.astype(np.float64)

    m = mask(1234, 0, 0)
    keep = 1 - thr / 65536
    assert abs(m.mean() - keep) < 5 * (keep * (1 - keep) / n) ** 0.5
    sig = 5 / n ** 0.5
    for other in (np.roll(m, 1), np.roll(m, 256), mask(1234, 0, 1), mask(1234, 1, 0), mask(1235, 0, 0)):
        assert abs(np.corrcoef(m, other)[0, 1]) < sig
    assert abs(np.corrcoef(m[0::2], m[1::2])[0, 1]) < sig * 2 ** 0.5      # the two halves of one hash
    assert np.array_equal(m, mask(1234, 0, 0))


# ---- general shapes (smx_*_ex): padded rows, Nyquist bin, the eight-band full-spectrum kernel ----------
def run_emu_ex(lib, mode, xin, wr, wi, bias, xk, conj, n_fft, k):
    B, R, D = xin.shape
    F = wr.shape[1]
    y = np.zeros((B, R, D), np.float32)
    if xk is None:
        xk = np.zeros((B, k, D, 2), np.float32)
    ps = np.zeros((B, k, D, 2), np.float32)
    gb = np.zeros((B, D), np.float32)
    lib.emu_fused_ex.restype = ctypes.c_int
    rc = lib.emu_fused_ex(mode, _p(xin), _p(wr), _p(wi), _p(bias), _p(y), _p(xk), _p(ps), _p(gb),
                          B, R, D, F, n_fft, k, conj, 0)
    assert rc == 0
    return y, xk, ps, gb


EX = [  # (B, rows, D, F, n_fft, k)
    (1, 256, 4, 129, 256, 129),        # one band, self-paired Nyquist slot
    (2, 192, 6, 129, 256, 129),        # + zero-padded rows
    (1, 512, 4, 257, 512, 257),        # two bands + Nyquist
    (1, 1024, 2, 513, 1024, 513),      # four bands + Nyquist
    (1, 768, 2, 385, 768, 385),        # L = 3: Nyquist as an ordinary +/- pair
    (1, 300, 4, 100, 512, 100),        # padded rows, pruned
    (1, 2048, 2, 1025, 2048, 1025),    # eight bands: per-residue spectra + fft8 across them
    (1, 1024, 4, 1025, 2048, 1025),    # ... on zero-padded rows (fft_lm's default lengths)
    (1, 1500, 2, 700, 2048, 700),      # ... pruned to 700 bins
]


@pytest.mark.parametrize("B,R,D,F,n_fft,k", EX)
def test_emulated_general_shapes(emu, B, R, D, F, n_fft, k):
    rng = np.random.default_rng(R + D + k)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    y, xk, _, _ = run_emu_ex(emu, 0, x, wr, wi, b, None, 0, n_fft, k)
    y_ref, X_ref = so.forward_closed_ex(x, wr, wi, b, n_fft, k)
    assert rel_err(y, y_ref) <= TOL_ACT
    assert rel_err(xk[..., 0] + 1j * xk[..., 1], X_ref) <= TOL_ACT
    gx, _, ps, gb = run_emu_ex(emu, 1, g, wr, wi, None, xk, 1, n_fft, k)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed_ex(x, wr, wi, g, n_fft, k)
    assert rel_err(gx, gx_ref) <= TOL_ACT
    P = (ps[..., 0] + 1j * ps[..., 1]).sum(axis=0)                     # (k, D) = sum_b X conj(G) / N
    assert rel_err(P.real.T, gwr_ref[:, :k]) <= TOL_PARAM
    assert rel_err(-P.imag.T, gwi_ref[:, :k]) <= TOL_PARAM
    assert rel_err(gb.sum(axis=0), gb_ref) <= TOL_PARAM


# ---- sixteen-row decimation (k_fused16): N = 16 P for any P, zero-padded rows, one and two bands ----------------
D16 = [  # (B, rows, D, F, n_fft, k)
    (2, 80, 6, 20, 80, 20),            # P = 5: one tile, 11 of 16 residues are padding
    (1, 4000, 4, 128, 4000, 128),      # the benchmark length: 250 residues = 15 full tiles + 10
    (1, 4112, 2, 100, 4112, 100),      # P = 257: a last tile with ONE residue
    (1, 400, 4, 180, 400, 180),        # two bands
    (2, 104, 4, 103, 208, 103),        # zero-padded rows: N = 104 as the even bins of 208 (functional.spectral_mix)
    (1, 16, 2, 8, 16, 8),              # P = 1
]


@pytest.mark.parametrize("B,R,D,F,n_fft,k", D16)
def test_emulated_sixteen_row_decimation(emu, B, R, D, F, n_fft, k):
    """The tile functions of k_fused16 (load_tile16, fwd16_phase2, inv16_phase1, store_tile16 around the shared
    fwd_phase1 / unpack / inv_phase2) for a whole workgroup on the CPU, against the fp64 closed forms."""
    rng = np.random.default_rng(R + D + k)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)

    def run(mode, xin, bias, xk, conj):
        y = np.zeros((B, R, D), np.float32)
        if xk is None:
            xk = np.zeros((B, k, D, 2), np.float32)
        ps = np.zeros((B, k, D, 2), np.float32)
        gb = np.zeros((B, D), np.float32)
        emu.emu_fused16.restype = ctypes.c_int
        assert emu.emu_fused16(mode, _p(xin), _p(wr), _p(wi), _p(bias), _p(y), _p(xk), _p(ps), _p(gb),
                               B, R, D, F, n_fft, k, conj) == 0
        return y, xk, ps, gb

    y, xk, _, _ = run(0, x, b, None, 0)
    y_ref, X_ref = so.forward_closed_ex(x, wr, wi, b, n_fft, k)
    assert rel_err(y, y_ref) <= TOL_ACT
    assert rel_err(xk[..., 0] + 1j * xk[..., 1], X_ref) <= TOL_ACT
    gx, _, ps, gb = run(1, g, None, xk, 1)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed_ex(x, wr, wi, g, n_fft, k)
    assert rel_err(gx, gx_ref) <= TOL_ACT
    P = (ps[..., 0] + 1j * ps[..., 1]).sum(axis=0)
    assert rel_err(P.real.T, gwr_ref[:, :k]) <= TOL_PARAM
    assert rel_err(-P.imag.T, gwi_ref[:, :k]) <= TOL_PARAM
    assert rel_err(gb.sum(axis=0), gb_ref) <= TOL_PARAM


# ---- four-step path: tile spectra -> workspace, per-thread column pairs, inverse tiles ------------------
FS = [  # (B, rows, D, F, n_fft, k)
    (1, 2048, 2, 1025, 2048, 1025),    # L = 8, full spectrum
    (1, 1024, 4, 1025, 2048, 1025),    # L = 8, zero-padded rows
    (1, 2000, 2, 700, 2048, 700),      # L = 8, pruned to 700 bins, padded
    (1, 4096, 2, 2049, 4096, 2049),    # L = 16
    (1, 5000, 2, 4097, 8192, 4097),    # L = 32, padded
    (1, 1280, 2, 641, 1280, 641),      # L = 5: odd L, Nyquist in column 128
    (1, 3072, 2, 1537, 3072, 1537),    # L = 12: generic L-point product
    (1, 6144, 2, 3073, 6144, 3073),    # L = 24: radix-2 step over two 12-point products
    (1, 6000, 2, 2000, 6656, 2000),    # L = 26 (2 x 13), padded, pruned
    (1, 16384, 2, 8193, 16384, 8193),  # L = 64: two-level column transform (4 threads per column pair)
    (1, 20000, 2, 3000, 32768, 3000),  # L = 128 (8 threads), padded rows, pruned
    (1, 65536, 2, 32769, 65536, 32769),  # L = 256 (16 threads)
    (1, 4352, 2, 2177, 4352, 2177),    # round 3: odd tile counts above 16 in one thread's registers: L = 17
    (1, 6000, 2, 3201, 6400, 3201),    # L = 25, padded rows
    # round 3: first-level length L1 = 9 ... 15 (L = L1 L2)
    (1, 12288, 2, 6145, 12288, 6145),  # L = 48 = 12 x 4
    (1, 9000, 4, 4609, 9216, 4609),    # L = 36 = 9 x 4: the fourth thread of a column pair holds padding, padded rows
    (1, 13312, 2, 3000, 13312, 3000),  # L = 52 = 13 x 4, pruned
    (1, 20480, 2, 10241, 20480, 10241),  # L = 80 = 10 x 8: two q1 per thread, threads 5 ... 7 padding
    (1, 30000, 2, 18433, 36864, 18433),  # L = 144 = 9 x 16: one q1 per thread, seven threads padding; padded rows
]


@pytest.mark.parametrize("B,R,D,F,n_fft,k", FS)
def test_emulated_fourstep(emu, B, R, D, F, n_fft, k):
    rng = np.random.default_rng(R + D + k + 1)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)

    def run(mode, xin, bias, xk, conj):
        y = np.zeros((B, R, D), np.float32)
        if xk is None:
            xk = np.zeros((B, k, D, 2), np.float32)
        ps = np.zeros((B, k, D, 2), np.float32)
        gb = np.zeros((B, D), np.float32)
        emu.emu_fourstep_ex.restype = ctypes.c_int
        assert emu.emu_fourstep_ex(mode, _p(xin), _p(wr), _p(wi), _p(bias), _p(y), _p(xk), _p(ps), _p(gb),
                                   B, R, D, F, n_fft, k, conj, None, None) == 0
        return y, xk, ps, gb

    y, xk, _, _ = run(0, x, b, None, 0)
    y_ref, X_ref = so.forward_closed_ex(x, wr, wi, b, n_fft, k)
    assert rel_err(y, y_ref) <= TOL_ACT
    assert rel_err(xk[..., 0] + 1j * xk[..., 1], X_ref) <= TOL_ACT
    gx, _, ps, gb = run(1, g, None, xk, 1)
    gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed_ex(x, wr, wi, g, n_fft, k)
    assert rel_err(gx, gx_ref) <= TOL_ACT
    P = (ps[..., 0] + 1j * ps[..., 1]).sum(axis=0)
    assert rel_err(P.real.T, gwr_ref[:, :k]) <= TOL_PARAM
    assert rel_err(-P.imag.T, gwi_ref[:, :k]) <= TOL_PARAM
    assert rel_err(gb.sum(axis=0), gb_ref) <= TOL_PARAM


# ---- per-(batch row, channel) factor on the filter (row_scale) and its gradient -----------------------------
@pytest.mark.parametrize("B,R,D,F,n_fft,k,path", [
    (2, 192, 6, 129, 256, 129, "fused"), (2, 512, 4, 257, 512, 257, "fused"), (2, 1024, 4, 513, 1024, 513, "fused"),
    (2, 1024, 4, 1025, 2048, 1025, "fused"), (2, 1024, 4, 1025, 2048, 1025, "fourstep"),
    (2, 3000, 2, 2049, 4096, 2049, "fourstep"), (1, 12000, 2, 8193, 16384, 8193, "fourstep"),
    (1, 9216, 2, 4609, 9216, 4609, "fourstep"), (1, 20000, 2, 10241, 20480, 10241, "fourstep")])
def test_emulated_row_scale(emu, B, R, D, F, n_fft, k, path):
    rng = np.random.default_rng(R + k)
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    wr = (1 + 0.5 * rng.standard_normal((D, F))).astype(np.float32)
    wi = (0.5 * rng.standard_normal((D, F))).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)

    def run(mode, xin, xk, conj, gsc):
        y = np.zeros((B, R, D), np.float32)
        if xk is None:
            xk = np.zeros((B, k, D, 2), np.float32)
        ps = np.zeros((B, k, D, 2), np.float32)
        gb = np.zeros((B, D), np.float32)
        if path == "fused":
            emu.emu_fused_ex2.restype = ctypes.c_int
            rc = emu.emu_fused_ex2(mode, _p(xin), _p(wr), _p(wi), None, _p(y), _p(xk), _p(ps), _p(gb), B, R, D, F,
                                   n_fft, k, conj, 0, _p(sc), _p(gsc))
        else:
            emu.emu_fourstep_ex.restype = ctypes.c_int
            rc = emu.emu_fourstep_ex(mode, _p(xin), _p(wr), _p(wi), None, _p(y), _p(xk), _p(ps), _p(gb), B, R, D, F,
                                     n_fft, k, conj, _p(sc), _p(gsc))
        assert rc == 0
        return y, xk, ps

    y, xk, _ = run(0, x, None, 0, None)
    y0, _ = so.forward_closed_ex(x, wr, wi, None, n_fft, k)
    assert rel_err(y, y0 * sc[:, None, :]) <= TOL_ACT
    gsc = np.zeros((B, D), np.float32)
    gx, _, ps = run(1, g, xk, 1, gsc)
    gx_ref, gwr_ref, gwi_ref, _ = so.backward_closed_ex(x, wr, wi, g * sc[:, None, :], n_fft, k)
    assert rel_err(gx, gx_ref) <= TOL_ACT
    P = (ps[..., 0] + 1j * ps[..., 1]).sum(axis=0)
    assert rel_err(P.real.T, gwr_ref[:, :k]) <= TOL_PARAM and rel_err(-P.imag.T, gwi_ref[:, :k]) <= TOL_PARAM
    assert rel_err(gsc, (g.astype(np.float64) * y0).sum(axis=1)) <= TOL_PARAM


# ---- rank-one filter on the four-step path: fft_lm's causal convolution, packed spectrum times H ---------------
@pytest.mark.parametrize("B,R,D,N", [(2, 1024, 4, 2048), (1, 1500, 6, 2048), (2, 2048, 2, 4096),
                                     (1, 5000, 4, 8192), (1, 16384, 2, 16384), (1, 40000, 2, 65536)])
def test_emulated_rank_one_conv(emu, B, R, D, N):
    import torch
    rng = np.random.default_rng(R + D)
    Fb = N // 2 + 1
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    hr = rng.standard_normal(Fb).astype(np.float32)
    hi = rng.standard_normal(Fb).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)
    ndt = (D + 31) // 32
    xs = np.zeros((B * ndt * (N // 256) * 4096, 2), np.float32)
    emu.emu_conv.restype = ctypes.c_int
    y = np.zeros((B, R, D), np.float32)
    assert emu.emu_conv(0, _p(x), _p(hr), _p(hi), _p(sc), _p(y), _p(xs), None, None, B, R, D, N) == 0
    gx = np.zeros((B, R, D), np.float32)
    P = np.zeros((N, 2), np.float32)
    gs = np.zeros((B, D), np.float32)
    assert emu.emu_conv(1, _p(g), _p(hr), _p(hi), _p(sc), _p(gx), _p(xs), _p(P), _p(gs), B, R, D, N) == 0
    # reference: autograd of the op sequence in float64
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    hrt = torch.tensor(hr, dtype=torch.float64, requires_grad=True)
    hit = torch.tensor(hi, dtype=torch.float64, requires_grad=True)
    sct = torch.tensor(sc, dtype=torch.float64, requires_grad=True)
    X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, N - R)), dim=1)
    yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=N, dim=1)[:, :R] * sct[:, None, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    assert rel_err(y, yr.detach().numpy()) <= TOL_ACT
    assert rel_err(gx, xt.grad.numpy()) <= TOL_ACT
    assert rel_err(gs, sct.grad.numpy()) <= TOL_PARAM
    # grad_H from the packed sums: Q = Hermitian part of P, grad_H[f] = c_f Q[f] / N
    Pc = P[:, 0].astype(np.float64) + 1j * P[:, 1]
    Q = 0.5 * (Pc[:Fb] + np.conj(Pc[(N - np.arange(Fb)) % N]))
    c = np.full(Fb, 2.0); c[0] = 1.0; c[-1] = 1.0
    gH = c * Q / N
    assert rel_err(gH.real, hrt.grad.numpy()) <= TOL_PARAM
    gi = gH.imag.copy(); gi[0] = 0.0; gi[-1] = 0.0
    assert rel_err(gi, hit.grad.numpy()) <= TOL_PARAM


# ---- the same filter in one launch (k_conv1): two half-length transforms by parity of the bin, 512 threads ------
@pytest.mark.parametrize("nj", [16, 8])
@pytest.mark.parametrize("B,R,D,N", [(2, 1024, 4, 2048), (1, 1000, 6, 2048), (1, 700, 34, 2048), (2, 512, 4, 1024),
                                     (1, 300, 2, 1024), (2, 256, 6, 512), (1, 101, 2, 512), (1, 1024, 18, 2048),
                                     (1, 1500, 4, 2048), (1, 2048, 2, 2048), (2, 1025, 6, 2048), (1, 700, 4, 1024),
                                     (1, 512, 2, 512), (1, 300, 34, 512)])          # the last six: rows > n_fft / 2 (folded)
def test_emulated_rank_one_conv_single_launch(emu, B, R, D, N, nj):
    import torch
    rng = np.random.default_rng(R + D + 1)
    Fb = N // 2 + 1
    x = rng.standard_normal((B, R, D)).astype(np.float32)
    g = rng.standard_normal((B, R, D)).astype(np.float32)
    hr = rng.standard_normal(Fb).astype(np.float32)
    hi = rng.standard_normal(Fb).astype(np.float32)
    sc = (0.5 + rng.random((B, D))).astype(np.float32)
    ndt = (D + 2 * nj - 1) // (2 * nj)                 # nj channel pairs per workgroup: 16 (512 threads) or 8 (256)
    xs = np.zeros((B * ndt * (N // 512) * 16 * 32 * nj, 2), np.float32)
    emu.emu_conv1.restype = ctypes.c_int
    y = np.zeros((B, R, D), np.float32)
    assert emu.emu_conv1(0, _p(x), _p(hr), _p(hi), _p(sc), _p(y), _p(xs), None, None, B, R, D, N, nj) == 0
    gx = np.zeros((B, R, D), np.float32)
    P = np.zeros((N, 2), np.float32)
    gs = np.zeros((B, D), np.float32)
    assert emu.emu_conv1(1, _p(g), _p(hr), _p(hi), _p(sc), _p(gx), _p(xs), _p(P), _p(gs), B, R, D, N, nj) == 0
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    hrt = torch.tensor(hr, dtype=torch.float64, requires_grad=True)
    hit = torch.tensor(hi, dtype=torch.float64, requires_grad=True)
    sct = torch.tensor(sc, dtype=torch.float64, requires_grad=True)
    X = torch.fft.rfft(torch.nn.functional.pad(xt, (0, 0, 0, N - R)), dim=1)
    yr = torch.fft.irfft(X * torch.complex(hrt, hit)[None, :, None], n=N, dim=1)[:, :R] * sct[:, None, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    assert rel_err(y, yr.detach().numpy()) <= TOL_ACT
    assert rel_err(gx, xt.grad.numpy()) <= TOL_ACT
    assert rel_err(gs, sct.grad.numpy()) <= TOL_PARAM
    Pc = P[:, 0].astype(np.float64) + 1j * P[:, 1]
    Q = 0.5 * (Pc[:Fb] + np.conj(Pc[(N - np.arange(Fb)) % N]))
    c = np.full(Fb, 2.0); c[0] = 1.0; c[-1] = 1.0
    gH = c * Q / N
    assert rel_err(gH.real, hrt.grad.numpy()) <= TOL_PARAM
    gi = gH.imag.copy(); gi[0] = 0.0; gi[-1] = 0.0
    assert rel_err(gi, hit.grad.numpy()) <= TOL_PARAM


# ---- synthesis from a given one-sided spectrum (smx_irfft_ex): synth_fill / fs_synth_columns ------------------
@pytest.mark.parametrize("B,R,D,N,k,fs", [
    (2, 256, 6, 256, 129, 0),       # one band, self-paired Nyquist slot
    (1, 200, 4, 256, 40, 0),        # cropped rows, few bins
    (2, 512, 4, 512, 257, 0),       # two bands + Nyquist
    (1, 700, 34, 1024, 513, 0),     # four bands + Nyquist, ragged d-tile, cropped rows
    (1, 1024, 2, 4096, 300, 0),     # four bands, k < N/2
    (2, 496, 10, 768, 385, 0),      # three tiles under four bands: the Nyquist bin sits in two slots
    (1, 1024, 4, 2048, 1025, 1),    # four-step, L = 8 (fft_lm default: seq 1024 + kernel 128)
    (1, 1500, 6, 2048, 1025, 2),    # the same length, eight bands in registers (k_synth8)
    (1, 2048, 2, 2048, 700, 2),
    (1, 1280, 2, 1280, 600, 1),     # four-step, L = 5, k < N/2 + 1
    (1, 3000, 2, 4096, 2049, 1),    # four-step, L = 16
    (1, 6144, 4, 6144, 3073, 1),    # four-step, L = 24
    (1, 9000, 2, 16384, 8193, 1),   # two-level columns, L = 64
    (1, 32768, 2, 32768, 5000, 1),  # L = 128, pruned
    (1, 4352, 2, 4352, 2177, 1),    # round 3: L = 17
    (1, 9216, 2, 9216, 4609, 1),    # round 3: L = 36 = 9 x 4 (a padded thread per column pair)
    (1, 11000, 4, 12288, 6145, 1),  # L = 48 = 12 x 4, cropped rows
    (1, 20480, 2, 20480, 7000, 1),  # L = 80 = 10 x 8, pruned
])
@pytest.mark.parametrize("herm", [0, 1])
def test_emulated_synthesis(emu, B, R, D, N, k, fs, herm):
    rng = np.random.default_rng(N + k + D)
    spec = rng.standard_normal((B, k, D, 2)).astype(np.float32)
    y = np.zeros((B, R, D), np.float32)
    emu.emu_synth.restype = ctypes.c_int
    emu.emu_synth.argtypes = [FP, FP] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
    scale = 1.0 / N if herm else 1.0
    assert emu.emu_synth(_p(spec), _p(y), B, R, D, N, k, scale, herm, fs) == 0
    S = spec[..., 0].astype(np.float64) + 1j * spec[..., 1]
    full = np.zeros((B, N // 2 + 1, D), np.complex128)
    full[:, :k] = S
    if herm:
        ref = np.fft.irfft(full, n=N, axis=1)[:, :R]                      # numpy ignores Im of DC / Nyquist too
    else:                                                                 # weight 1 on every bin: Re of the sum
        n = np.arange(R)[:, None] * np.arange(k)[None, :]
        E = np.exp(2j * np.pi * (n % N) / N)
        ref = np.einsum("nf,bfd->bnd", E, S).real
    assert rel_err(y, ref) <= TOL_ACT


@pytest.mark.parametrize("B,N,Dc", [(1, 16384, 1), (1, 32768, 2), (1, 65536, 1), (1, 9216, 2), (1, 20480, 1)])
def test_emulated_complex_fft_two_level(emu, B, N, Dc):
    """MODE 3 of the two-level column transform: the packed bins of a complex sequence go straight out."""
    rng = np.random.default_rng(N + Dc)
    z = (rng.standard_normal((B, N, Dc)) + 1j * rng.standard_normal((B, N, Dc))).astype(np.complex64)
    out = np.zeros((B, N, Dc, 2), np.float32)
    emu.emu_cfft_big.restype = ctypes.c_int
    zr = np.ascontiguousarray(z.view(np.float32).reshape(B, N, 2 * Dc))
    assert emu.emu_cfft_big(_p(zr), _p(out), B, N, 2 * Dc) == 0
    ref = np.fft.fft(z.astype(np.complex128), axis=1)
    assert rel_err(out[..., 0] + 1j * out[..., 1], ref) <= TOL_ACT
