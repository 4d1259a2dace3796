"""The reference's OWN self-tests for this path, re-run against the drop-in classes on the GPU.

  * `test_spectral_mixing_correctness`  -- reference fft_tensor/spectral_layers.py:258-317
  * `test_wirtinger_gradients`          -- reference fft_tensor/wirtinger_ops.py:205-378

Same shapes, same assertions and thresholds as the reference (its checks are properties, it holds no
numeric golden values -- SURVEY.md 8c); the transform under test is the native one (pruned_rfft /
SpectralMixingLayer / WirtingerGradient / WirtingerSpectralFilter of this package) instead of torch.fft.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pkg():
    import tensor_cuda_fft_amd as pkg
    return pkg


def test_spectral_mixing_correctness(gpu):
    """reference spectral_layers.py:258-317, tests 1-5."""
    pkg = _pkg()
    torch.manual_seed(0)
    B, T, D = 2, 128, 256
    x = torch.randn(B, T, D, device=gpu, requires_grad=True)

    # 1. round trip (:267-273): forward transform then inverse gives x back.  The native transform keeps
    #    bins [0, T/2): a band-limited x makes that the whole signal; identity filter W = (1, 2, 2, ...)
    #    undoes the one-sided 1/2, so the layer IS ifft(fft(x)).real here.
    k = T // 2
    spec = torch.fft.rfft(torch.randn(B, T, D, device=gpu), dim=1)
    spec[:, k:] = 0
    xb = torch.fft.irfft(spec, n=T, dim=1)
    ident = pkg.SpectralMixingLayer(D, num_filters=k).to(gpu)
    with torch.no_grad():
        ident.weight_real.fill_(2.0); ident.weight_real[:, 0] = 1.0
    error = torch.norm(ident(xb) - xb) / torch.norm(xb)
    assert error < 1e-5, f"FFT round-trip failed: {error}"

    # 2. Parseval (:275-284), on the native spectrum: sum |X_f|^2 over the full spectrum / T == sum x^2.
    #    Bins [0, T/2) come from pruned_rfft; a band-limited real signal has X_{T-f} = conj X_f and no Nyquist.
    xk = pkg.pruned_rfft(xb, k)
    energy_time = torch.sum(xb ** 2).item()
    energy_freq = (torch.sum(torch.abs(xk[:, :1]) ** 2) + 2 * torch.sum(torch.abs(xk[:, 1:]) ** 2)).item() / T
    ratio = energy_freq / energy_time
    assert abs(ratio - 1.0) < 0.01, f"Energy not preserved: {ratio}"

    # 3. gradient flow (:286-297)
    layer = pkg.SpectralMixingLayer(D, learnable=True).to(gpu)
    y = layer(x)
    y.sum().backward()
    grad_norm = torch.norm(x.grad).item()
    assert grad_norm > 0, "Gradients are zero"
    assert torch.isfinite(x.grad).all(), "Gradients contain NaN/Inf"
    assert abs(grad_norm - (B * T * D) ** 0.5) < 1e-2          # known answer of this very test: grad_x == 1

    # 4. identity preservation (:299-307)
    y_identity = pkg.SpectralMixingLayer(D, learnable=False).to(gpu)(x)
    assert (torch.norm(y_identity - x) / torch.norm(x)).item() < 1e-5

    # 5. domain legality (:309-315)
    assert not torch.is_complex(x) and not torch.is_complex(y), "Time domain must be real"
    assert torch.is_complex(xk), "Freq domain must be complex"
    assert layer.verify_energy_preservation(x.detach(), x.detach()) == pytest.approx(1.0, abs=1e-6)


def test_wirtinger_gradients(gpu):
    """reference wirtinger_ops.py:205-378, tests 1-4."""
    pkg = _pkg()
    torch.manual_seed(0)

    # 1. basic gradient flow (:219-248)
    x = torch.complex(torch.randn(2, 8, 16, device=gpu), torch.randn(2, 8, 16, device=gpu))
    weight_param = pkg.ComplexParameter((16, 4), init_mode="uniform").to(gpu)
    weight_broadcast = weight_param()[:, :4].T.unsqueeze(0)
    y = pkg.WirtingerGradient.apply(x[:, :4, :], weight_broadcast)
    torch.abs(y).sum().backward()
    assert weight_param.real.grad is not None, "Real gradient missing"
    assert weight_param.imag.grad is not None, "Imaginary gradient missing"
    assert torch.norm(weight_param.real.grad).item() > 0, "Real gradient is zero"
    assert torch.norm(weight_param.imag.grad).item() > 0, "Imaginary gradient is zero"

    # 2. phase learning (:250-292)
    target_phase = torch.randn(16, 4, device=gpu)
    target = torch.complex(torch.cos(target_phase), torch.sin(target_phase))
    filt = pkg.WirtingerSpectralFilter(16, 8).to(gpu)
    opt = torch.optim.Adam([{"params": filt.weight.real}, {"params": filt.weight.imag}], lr=0.1)
    initial_phase = filt.weight.phase()[:, :4].clone()
    for _ in range(50):
        opt.zero_grad()
        loss = torch.mean(torch.abs(filt.weight()[:, :4] - target) ** 2)
        loss.backward()
        opt.step()
    phase_change = torch.norm(filt.weight.phase()[:, :4] - initial_phase).item()
    assert phase_change > 0.1, f"Phase didn't change: {phase_change}"

    # 3. Wirtinger Function vs plain complex autograd (:294-333): the reference only compares magnitudes
    #    on different inputs; on the SAME inputs the two must agree exactly.
    x_w = torch.complex(torch.randn(2, 8, 16, device=gpu), torch.randn(2, 8, 16, device=gpu))
    weight_w = pkg.ComplexParameter((16, 4)).to(gpu)
    torch.abs(pkg.WirtingerGradient.apply(x_w[:, :4, :], weight_w()[:, :4].T.unsqueeze(0))).sum().backward()
    gw = (weight_w.real.grad.clone(), weight_w.imag.grad.clone())
    weight_w.real.grad = None; weight_w.imag.grad = None
    torch.abs(x_w[:, :4, :] * weight_w()[:, :4].T.unsqueeze(0)).sum().backward()
    assert torch.allclose(gw[0], weight_w.real.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gw[1], weight_w.imag.grad, rtol=1e-5, atol=1e-6)

    # 4. magnitude learning through the filter's parameters (:335-375)
    filt = pkg.WirtingerSpectralFilter(8, 16).to(gpu)
    initial_mag = filt.weight.magnitude().mean().item()
    target = torch.complex(torch.ones(8, 16, device=gpu), torch.zeros(8, 16, device=gpu))
    target[:, 8:] = 0.1
    opt = torch.optim.Adam([{"params": filt.weight.real, "lr": 0.1}, {"params": filt.weight.imag, "lr": 0.1}])
    for _ in range(20):
        opt.zero_grad()
        torch.mean(torch.abs(filt.weight() - target) ** 2).backward()
        opt.step()
    assert abs(filt.weight.magnitude().mean().item() - initial_mag) > 0.01, "Magnitude didn't change"


def test_filter_learns_through_the_native_path(gpu):
    """Beyond the reference's parameter-only loops: the same Adam loop with the gradients coming through
    the native transform (x -> fft -> WirtingerSpectralFilter -> ifft), loss on the time-domain output."""
    pkg = _pkg()
    torch.manual_seed(1)
    B, N, D, F = 4, 256, 16, 8
    x = torch.randn(B, N, D, device=gpu)
    teacher = pkg.WirtingerSpectralFilter(D, F).to(gpu)
    with torch.no_grad():
        teacher.weight.real.normal_(1.0, 0.5); teacher.weight.imag.normal_(0.0, 0.5)
    target = pkg.spectral_mix_with_filter(x, teacher).detach()
    student = pkg.WirtingerSpectralFilter(D, F).to(gpu)
    opt = torch.optim.Adam(student.parameters(), lr=0.05)
    first = None
    for _ in range(150):
        opt.zero_grad()
        loss = torch.mean((pkg.spectral_mix_with_filter(x, student) - target) ** 2)
        loss.backward()
        opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < 0.02 * first
    # the filter itself is recovered, real AND imaginary part (bin 0 only through its real part: the DC
    # bin of a real signal is real, so Im W[:, 0] never reaches the output)
    ws, wt = student.weight().detach(), teacher.weight().detach()
    assert (torch.norm(ws[:, 1:] - wt[:, 1:]) / torch.norm(wt[:, 1:])).item() < 0.15
    assert (torch.norm(ws[:, 0].real - wt[:, 0].real) / torch.norm(wt[:, 0].real)).item() < 0.15
