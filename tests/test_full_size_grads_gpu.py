"""Full-batch parity at the BASELINE sizes, oracle evaluated in channel slabs.

Every (b, d) column is an independent transform and the parameter gradients of channel d only involve
channel d, so the fp64 oracle can be run 32 channels at a time (< 1 GB of host memory) and still cover
the WHOLE batch: the 8-wide unrolled branch of k_gradw (B >= 57) and both band counts are compared with
the oracle here, which the sub-batch checks of test_parity_gpu.py cannot do.

C5 goes through its own API, `spectral_mix_with_filter(x, WirtingerSpectralFilter)` =
ifft(filter(fft(x))).real (reference wirtinger_ops.py:170-203), forward AND backward; small shapes of
the same unit are pinned by reference-generated fixtures (tests/golden/W0*.npz).
"""
import numpy as np
import pytest
import torch

from conftest import TOL_ACT, TOL_PARAM, load_golden, rel_err
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
SLAB = 32


def _pkg():
    import tensor_cuda_fft_amd as pkg
    return pkg


@pytest.mark.parametrize("name", ["W01_wfused_2x512x64", "W02_wfused_2x1024x12"])
def test_wirtinger_fused_unit_matches_reference(gpu, name):
    """One- and two-band fixtures produced by the reference: y, grad_x and the gradients that reach
    filt.weight.real / filt.weight.imag through the fused path."""
    pkg = _pkg()
    z = load_golden(name)
    D, F = z["w_real"].shape
    filt = pkg.WirtingerSpectralFilter(D, F).to(gpu)
    filt.load_state_dict({"weight.real": torch.from_numpy(z["w_real"]),
                          "weight.imag": torch.from_numpy(z["w_imag"])})
    x = torch.from_numpy(z["x"]).to(gpu).requires_grad_(True)
    y = pkg.spectral_mix_with_filter(x, filt)
    y.backward(torch.from_numpy(z["g"]).to(gpu))
    c = lambda t: t.detach().cpu().numpy()
    assert rel_err(c(y), z["y"]) <= TOL_ACT
    assert rel_err(c(x.grad), z["grad_x"]) <= TOL_ACT
    assert rel_err(c(filt.weight.real.grad), z["grad_w_real"]) <= TOL_PARAM
    assert rel_err(c(filt.weight.imag.grad), z["grad_w_imag"]) <= TOL_PARAM


def _slab_check(x, g, wr, wi, bias, y, gx, gwr, gwi, gb):
    """Compare device results with the fp64 closed forms, SLAB channels at a time.  Errors are
    max-normalised over the WHOLE tensor (the stated tolerance), so the maxima are accumulated first."""
    D = x.shape[2]
    num = dict(y=0.0, gx=0.0, gwr=0.0, gwi=0.0, gb=0.0)
    den = dict(y=0.0, gx=0.0, gwr=0.0, gwi=0.0, gb=0.0)
    for c0 in range(0, D, SLAB):
        cs = slice(c0, min(D, c0 + SLAB))
        xs = x[:, :, cs].cpu().numpy(); gs = g[:, :, cs].cpu().numpy()
        w_r = wr[cs].cpu().numpy(); w_i = wi[cs].cpu().numpy()
        b = None if bias is None else bias[cs].cpu().numpy()
        y_ref, _ = so.forward_closed(xs, w_r, w_i, b)
        gx_ref, gwr_ref, gwi_ref, gb_ref = so.backward_closed(xs, w_r, w_i, gs)
        got = dict(y=y[:, :, cs], gx=gx[:, :, cs], gwr=gwr[cs], gwi=gwi[cs])
        ref = dict(y=y_ref, gx=gx_ref, gwr=gwr_ref, gwi=gwi_ref)
        if gb is not None:
            got["gb"] = gb[cs]; ref["gb"] = gb_ref
        for k in got:
            a = got[k].cpu().numpy().astype(np.float64)
            num[k] = max(num[k], float(np.abs(a - ref[k]).max()))
            den[k] = max(den[k], float(np.abs(ref[k]).max()))
    return {k: (num[k] / den[k] if den[k] > 0 else num[k]) for k in num}


@pytest.mark.timeout(900)
def test_c5_wirtinger_filter_full_size_fwd_bwd(gpu):
    """BASELINE config 5: (64, 4096, 512), F = 256, through spectral_mix_with_filter, grad-check <= 1e-4
    against the CPU closed form on the full batch."""
    pkg = _pkg()
    B, N, D, F = 64, 4096, 512, 256
    torch.manual_seed(1234)
    filt = pkg.WirtingerSpectralFilter(D, F).to(gpu)
    with torch.no_grad():
        filt.weight.real.normal_(1.0, 0.5); filt.weight.imag.normal_(0.0, 0.5)
    gen = torch.Generator(device=gpu).manual_seed(1234)
    x = torch.randn(B, N, D, device=gpu, generator=gen).requires_grad_(True)
    g = torch.randn(B, N, D, device=gpu, generator=gen)
    y = pkg.spectral_mix_with_filter(x, filt)
    y.backward(g)
    torch.cuda.synchronize()
    e = _slab_check(x.detach(), g, filt.weight.real.detach(), filt.weight.imag.detach(), None,
                    y.detach(), x.grad, filt.weight.real.grad, filt.weight.imag.grad, None)
    assert e["y"] <= TOL_ACT and e["gx"] <= TOL_ACT, e
    assert e["gwr"] <= TOL_PARAM and e["gwi"] <= TOL_PARAM, e


@pytest.mark.timeout(900)
def test_c2_full_batch_parameter_gradients(gpu):
    """BASELINE config 2: (64, 4096, 256), F = 128, module API, all five outputs on the full batch."""
    pkg = _pkg()
    B, N, D, F = 64, 4096, 256, 128
    torch.manual_seed(1234)
    layer = pkg.SpectralMixingLayer(D, num_filters=F).to(gpu)
    with torch.no_grad():
        layer.weight_real.normal_(1.0, 0.5); layer.weight_imag.normal_(0.0, 0.5)
        layer.bias.normal_(0.0, 0.1)
    gen = torch.Generator(device=gpu).manual_seed(1234)
    x = torch.randn(B, N, D, device=gpu, generator=gen).requires_grad_(True)
    g = torch.randn(B, N, D, device=gpu, generator=gen)
    y = layer(x)
    y.backward(g)
    torch.cuda.synchronize()
    e = _slab_check(x.detach(), g, layer.weight_real.detach(), layer.weight_imag.detach(),
                    layer.bias.detach(), y.detach(), x.grad, layer.weight_real.grad,
                    layer.weight_imag.grad, layer.bias.grad)
    assert e["y"] <= TOL_ACT and e["gx"] <= TOL_ACT, e
    assert max(e["gwr"], e["gwi"], e["gb"]) <= TOL_PARAM, e


def test_cpu_resident_filter_raises(gpu):
    """A WirtingerSpectralFilter left on the CPU (or in float64) is refused with a Python error instead of
    handing a host pointer to a kernel."""
    pkg = _pkg()
    xf = torch.randn(2, 32, 16, device=gpu, dtype=torch.complex64)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        pkg.WirtingerSpectralFilter(16, 8)(xf)
    with pytest.raises(TypeError, match="float32"):
        pkg.WirtingerSpectralFilter(16, 8).to(gpu).double()(xf)
    with pytest.raises(ValueError, match="num_channels"):
        from tensor_cuda_fft_amd.wirtinger_ops import _FilterFn
        _FilterFn.apply(xf, torch.ones(8, 8, device=gpu), torch.zeros(8, 8, device=gpu))


def test_inference_does_not_save_the_spectrum(gpu):
    """Under torch.no_grad() (parameters still require grad) the forward neither writes the spectrum nor
    packs the filter: peak memory stays at input + output + workspace."""
    pkg = _pkg()
    layer = pkg.SpectralMixingLayer(256).to(gpu)
    x = torch.randn(16, 4096, 256, device=gpu)
    with torch.no_grad():
        layer(x)                                            # warm-up: tables, workspace
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = layer(x)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
    assert peak < y.numel() * 4 + (1 << 20), peak          # y only: no (B,k,D) complex64, no pack buffer
