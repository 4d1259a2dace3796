"""functional.rfft / functional.irfft (include/smx.h, smx_rfft_ex / smx_irfft_ex) against torch.fft in float64:
values and gradients on every plan the pair can take."""
import numpy as np
import pytest
import torch

from conftest import TOL_ACT, rel_err

pytestmark = pytest.mark.gpu

# (B, rows, D, n_fft, k or None = every bin)
SHAPES = [
    (2, 256, 6, 256, None),        # one band + self-paired Nyquist slot
    (3, 200, 32, 256, 40),         # cropped rows, few bins
    (2, 512, 8, 512, None),        # two bands
    (2, 700, 34, 1024, None),      # four bands, ragged d-tile, zero-padded rows
    (5, 496, 10, 768, None),       # three tiles under four bands: the Nyquist bin sits in two slots
    (4, 1024, 64, 2048, None),     # n_fft 2048, every bin: eight bands in registers (fft_lm default lengths)
    (2, 2048, 8, 2048, 700),       # n_fft 2048, 512 < k < n/2 + 1
    (1, 1280, 4, 1280, 600),       # four-step L = 5
    (2, 3000, 8, 4096, None),      # four-step L = 16
    (1, 8192, 4, 8192, None),      # four-step L = 32
    (2, 12000, 6, 16384, None),    # L = 64: two-level column transform
    (1, 32768, 4, 32768, 5000),    # L = 128, pruned
    (1, 65536, 2, 65536, None),    # L = 256
    (1, 6144, 4, 6144, None),      # L = 24: four-step
    (1, 12288, 4, 12288, None),    # L = 48 = 12 x 4: two-level columns, 12-point first level (round 3)
    (2, 9000, 6, 9216, None),      # L = 36 = 9 x 4 (a padded thread per column pair), zero-padded rows
    (1, 20480, 4, 20480, 7000),    # L = 80 = 10 x 8, pruned
    (1, 36864, 2, 36864, None),    # L = 144 = 9 x 16
    (1, 4352, 4, 4352, None),      # L = 17 (round 3: four-step; band groups + DFT products before)
    (1, 8704, 4, 8704, None),      # band groups (L = 34): DFT products for the synthesis
    (2, 5888, 6, 5888, None),      # L = 23
    (1, 65536, 8, 65536, 128),     # residue split plan: park + k_split_b
    (2, 100, 16, 128, None),       # direct plan (n_fft % 256 != 0)
    (2, 300, 7, 300, None),        # direct plan, odd D
    (1, 4000, 256, 4000, 64),      # direct plan at the MFMA threshold? (small k keeps it fast)
]


def _ref_rfft(x64, n, k):
    return torch.fft.rfft(x64, n=n, dim=1)[:, :k]


@pytest.mark.parametrize("B,R,D,n,k", SHAPES)
def test_rfft_forward_and_backward(B, R, D, n, k):
    from tensor_cuda_fft_amd import functional as Fn
    kk = n // 2 + 1 if k is None else k
    gen = torch.Generator().manual_seed(R + D)
    x = torch.randn(B, R, D, generator=gen)
    gr = torch.randn(B, kk, D, generator=gen)
    gi = torch.randn(B, kk, D, generator=gen)
    xg = x.cuda().requires_grad_(True)
    y = Fn.rfft(xg, n, k)
    assert y.shape == (B, kk, D) and y.dtype == torch.complex64
    y.backward(torch.complex(gr, gi).cuda())
    x64 = x.double().requires_grad_(True)
    y64 = _ref_rfft(x64, n, kk)
    y64.backward(torch.complex(gr.double(), gi.double()))
    assert rel_err(torch.view_as_real(y.detach()).cpu().numpy(), torch.view_as_real(y64.detach()).numpy()) <= TOL_ACT
    assert rel_err(xg.grad.cpu().numpy(), x64.grad.numpy()) <= TOL_ACT


@pytest.mark.parametrize("B,R,D,n,k", SHAPES)
def test_irfft_forward_and_backward(B, R, D, n, k):
    from tensor_cuda_fft_amd import functional as Fn
    kk = n // 2 + 1 if k is None else k
    gen = torch.Generator().manual_seed(R + 3 * D)
    s = torch.complex(torch.randn(B, kk, D, generator=gen), torch.randn(B, kk, D, generator=gen))
    g = torch.randn(B, R, D, generator=gen)
    sg = s.cuda().requires_grad_(True)
    y = Fn.irfft(sg, n, R)
    assert y.shape == (B, R, D) and y.dtype == torch.float32
    y.backward(g.cuda())
    s64 = s.to(torch.complex128).requires_grad_(True)
    y64 = torch.fft.irfft(s64, n=n, dim=1)[:, :R]
    y64.backward(g.double())
    assert rel_err(y.detach().cpu().numpy(), y64.detach().numpy()) <= TOL_ACT
    got, ref = sg.grad.cpu(), s64.grad
    # the imaginary parts of the DC / Nyquist rows do not reach y: their gradient is zero on both sides
    assert rel_err(torch.view_as_real(got).numpy(), torch.view_as_real(ref).numpy()) <= TOL_ACT


def test_round_trip_and_defaults():
    from tensor_cuda_fft_amd import functional as Fn
    x = torch.randn(2, 1024, 32, device="cuda")
    X = Fn.rfft(x)
    assert X.shape == (2, 513, 32)
    assert rel_err(Fn.irfft(X).cpu().numpy(), x.cpu().numpy()) <= TOL_ACT
    # zero-padded causal layout of fft_lm: rows 1024 in n = 2048, cropped back
    X2 = Fn.rfft(x, 2048)
    assert rel_err(Fn.irfft(X2, 2048, 1024).cpu().numpy(), x.cpu().numpy()) <= TOL_ACT
    # a spectrum longer than n // 2 + 1 is trimmed, a shorter one zero-padded, as torch.fft.irfft does
    ref = torch.fft.irfft(X.cpu().to(torch.complex128), n=512, dim=1)
    assert rel_err(Fn.irfft(X, 512).cpu().numpy(), ref.numpy()) <= TOL_ACT
    ref = torch.fft.irfft(X[:, :100].cpu().to(torch.complex128), n=1024, dim=1)
    assert rel_err(Fn.irfft(X[:, :100], 1024).cpu().numpy(), ref.numpy()) <= TOL_ACT


def test_argument_errors():
    from tensor_cuda_fft_amd import functional as Fn
    x = torch.randn(1, 300, 4, device="cuda")
    with pytest.raises(ValueError):
        Fn.rfft(x, 256)
    with pytest.raises(TypeError):
        Fn.irfft(x, 300)
    with pytest.raises(RuntimeError):
        Fn.irfft(torch.zeros(1, 5, 4, dtype=torch.complex64), 8)
    with pytest.raises(ValueError):
        Fn.irfft(torch.zeros(1, 5, 4, dtype=torch.complex64, device="cuda"), 8, 9)
