"""The oracle's restatements of the SURVEY 8(f) rows against fixtures produced by the reference modules
(tests/golden/make_golden.py): ports bit-exact (or to fp32 noise where the reference's own op order
differs between a module and its restated core), fp64 closed forms of the general transform consistent
with the ports.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import spectral_oracle as so

T = torch.from_numpy


def _sd(z):
    return {k[3:]: T(v) for k, v in z.items() if k.startswith("sd.")}


@pytest.mark.parametrize("name", ["F01_fixed_2x192x32", "F02_fixed_2x512x16", "F03_fixed_1x1024x8",
                                  "F04_fixed_2x100x16", "F05_fixed_2x100x16", "F06_fixed_2x300x9",
                                  "F07_fixed_1x8000x4"])
def test_fixed_block_port_matches_reference(name):
    z = load_golden(name)
    sd = _sd(z)
    x = T(z["x"])
    C = x.shape[2]
    cutoff = None if int(z["cutoff"]) < 0 else int(z["cutoff"])
    ln = torch.nn.functional.layer_norm
    h = ln(x, (C,), sd["ln.weight"], sd["ln.bias"], 1e-5)
    g_ctx = torch.sigmoid(h.mean(dim=1) @ sd["gate_ctx.weight"].T + sd["gate_ctx.bias"])
    y = so.causal_conv_port(h, sd["kernel"], sd["gain"], sd["gate_freq_logits"], g_ctx, cutoff,
                            int(z["transition_bins"]))
    x1 = x + y
    f = ln(x1, (C,), sd["ffn_ln.weight"], sd["ffn_ln.bias"], 1e-5)
    f = torch.nn.functional.gelu(f @ sd["ffn.0.weight"].T + sd["ffn.0.bias"]) @ sd["ffn.3.weight"].T \
        + sd["ffn.3.bias"]
    assert rel_err((x1 + f).numpy(), z["y"]) <= 1e-6


@pytest.mark.parametrize("name", ["T01_freqnative_2x192x16", "T02_freqnative_1x1024x8", "T03_freqnative_2x100x6",
                                  "T11_bicameral_2x192x16", "T12_bicameral_1x1024x8", "T13_bicameral_2x300x10"])
def test_spectrum_domain_twin_ports_match_reference(name):
    """FrequencyNativeBlock / BicameralBlock restated (oracle) against the reference modules' own runs: output
    and, through autograd of the port, the input gradient."""
    z = load_golden(name)
    sd = _sd(z)
    x = T(z["x"]).requires_grad_(True)
    cutoff = None if int(z["cutoff"]) < 0 else int(z["cutoff"])
    port = so.freq_native_block_port if "freqnative" in name else so.bicameral_block_port
    y = port(sd, x, cutoff, int(z["transition_bins"]))
    y.backward(T(z["g"]))
    assert rel_err(y.detach().numpy(), z["y"]) <= 1e-6
    assert rel_err(x.grad.numpy(), z["grad_x"]) <= 1e-5


def test_freqconv_port_matches_reference():
    z = load_golden("FC1_freqconv_2x33x8")
    y, gx, gk, gg = so.freqconv_port(T(z["x_freq"]), T(z["kernel_freq"]), T(z["gain"]), T(z["g"]))
    assert rel_err(y.numpy(), z["out"]) <= 1e-6 and rel_err(gx.numpy(), z["grad_x"]) <= 1e-6
    assert rel_err(gk.numpy(), z["grad_kernel"]) <= 1e-6 and rel_err(gg.numpy(), z["grad_gain"]) <= 1e-6


@pytest.mark.parametrize("name", ["P01_phase_2x512x32", "P02_phase_2x33x6", "P03_phase_1x2048x4"])
def test_phase_aware_port_and_closed_form(name):
    z = load_golden(name)
    sd = _sd(z)
    x = T(z["x"])
    y = so.phase_aware_port(x, sd["magnitude_filter"], sd["phase_filter"])
    assert rel_err(y.numpy(), z["y"]) <= 1e-6
    # the native op's formulation: W[d, f] = c_f m e^{ip} on every one-sided bin
    B, N, D = x.shape
    K = N // 2 + 1
    c = np.full(K, 2.0); c[0] = 1.0
    if N % 2 == 0:
        c[N // 2] = 1.0
    m, p = sd["magnitude_filter"].numpy().astype(np.float64), sd["phase_filter"].numpy().astype(np.float64)
    w_re, w_im = (m * np.cos(p))[:, None] * c[None], (m * np.sin(p))[:, None] * c[None]
    yc, _ = so.forward_closed_ex(x.numpy(), w_re, w_im, None, N, K)
    assert rel_err(yc, z["y"]) <= 2e-6


@pytest.mark.parametrize("name", ["M01_multi_2x1024x16", "M02_multi_2x50x8"])
def test_multiscale_port_matches_reference(name):
    z = load_golden(name)
    sd = _sd(z)
    x = T(z["x"])
    low, mid, high = so.multiscale_bands_port(x)
    lin = lambda t, n: t @ sd[n + ".weight"].T + sd[n + ".bias"]
    y = lin(torch.cat([lin(low, "low_freq"), lin(mid, "mid_freq"), lin(high, "high_freq")], -1), "fusion")
    assert rel_err(y.numpy(), z["y"]) <= 1e-6
    # the partition the drop-in relies on: the three bands add up to x
    assert rel_err((low + mid + high).numpy(), x.numpy()) <= 1e-6


@pytest.mark.parametrize("name", ["R01_rope_2x256x16", "R02_rope_2x40x8", "R03_rope_1x768x6"])
def test_rope_layer_port_matches_reference(name):
    z = load_golden(name)
    sd = _sd(z)
    x = T(z["x"])
    D = x.shape[2]
    ln = torch.nn.functional.layer_norm
    h = ln(x, (D,), sd["norm1.weight"], sd["norm1.bias"], 1e-5)
    x1 = x + so.complex_rope_mix_port(h, sd["rope.rotation"], sd["freq_filter"])
    h2 = ln(x1, (D,), sd["norm2.weight"], sd["norm2.bias"], 1e-5)
    lin = lambda t, n: t @ sd[n + ".weight"].T + sd[n + ".bias"]
    y = x1 + lin(torch.sigmoid(lin(h2, "glu.gate_proj")) * lin(h2, "glu.value_proj"), "glu.out_proj")
    assert rel_err(y.numpy(), z["y"]) <= 1e-6


@pytest.mark.parametrize("name", ["N01_fnet_2x256x8", "N02_fnet_2x30x5", "N03_fnet_1x1024x3", "N04_fnet_2x2048x5"])
def test_fnet_port_matches_reference(name):
    z = load_golden(name)
    assert rel_err(so.fnet_port(T(z["z"])).numpy(), z["out"]) <= 1e-6
    # backward of an unnormalised DFT: grad_z = conj(fft(conj(g)))
    gz = T(z["gz"])
    assert rel_err(so.fnet_port(gz.conj()).conj().resolve_conj().numpy(), z["grad_z"]) <= 1e-6


def test_closed_ex_agrees_with_causal_conv_port():
    """forward/backward_closed_ex (what the GPU tests compare the native op with) against autograd of the
    reference's op sequence, incl. zero-padded rows and the Nyquist bin."""
    torch.manual_seed(0)
    B, R, C, K = 2, 70, 4, 20
    n_fft = so.next_pow2(R + K - 1)                       # 128
    fb = n_fft // 2 + 1
    x = torch.randn(B, R, C, requires_grad=True)
    kern = torch.randn(K, dtype=torch.float64)
    gain = 1 + 0.3 * torch.randn(C, dtype=torch.float64)
    logits = torch.randn(fb, dtype=torch.float64)
    g = torch.randn(B, R, C)
    y = so.causal_conv_port(x.double(), kern, gain, logits, torch.ones(B, C, dtype=torch.float64), 40, 8)
    y.backward(g.double())
    kf = np.fft.rfft(np.pad(kern.numpy(), (0, n_fft - K)))
    c = np.full(fb, 2.0); c[0] = 1.0; c[-1] = 1.0
    mask = np.ones(fb); mask[32:40] = 0.5 * (1 + np.cos(np.pi * np.linspace(0, 1, 8))); mask[40:] = 0
    H = kf * c * (1 / (1 + np.exp(-logits.numpy()))) * mask
    W = gain.numpy()[:, None] * H[None]
    yc, _ = so.forward_closed_ex(x.detach().numpy(), W.real, W.imag, None, n_fft, fb)
    gxc, _, _, _ = so.backward_closed_ex(x.detach().numpy(), W.real, W.imag, g.numpy(), n_fft, fb)
    assert rel_err(yc, y.detach().numpy()) <= 1e-12
    assert rel_err(gxc, x.grad.numpy()) <= 1e-6          # x.grad is fp32 (x is an fp32 leaf)
