"""The oracle is only trusted because it reproduces what the reference produced (tests/golden)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names, load_golden, rel_err
from oracle import spectral_oracle as so

LAYER = [n for n in golden_names("layer") if "nolearn" not in n]


@pytest.mark.parametrize("name", LAYER)
def test_port_is_bit_exact_with_reference(name):
    z = load_golden(name)
    t = lambda a: torch.from_numpy(a)
    y, gx, gwr, gwi, gb = so.fwd_bwd_port(t(z["x"]), t(z["weight_real"]), t(z["weight_imag"]),
                                          t(z["bias"]), t(z["g"]))
    for got, key in ((y, "y"), (gx, "grad_x"), (gwr, "grad_w_real"), (gwi, "grad_w_imag"),
                     (gb, "grad_bias")):
        assert np.array_equal(got.numpy(), z[key]), key


@pytest.mark.parametrize("name", LAYER)
def test_closed_form_matches_reference(name):
    z = load_golden(name)
    y, _ = so.forward_closed(z["x"], z["weight_real"], z["weight_imag"], z["bias"])
    gx, gwr, gwi, gb = so.backward_closed(z["x"], z["weight_real"], z["weight_imag"], z["g"])
    assert rel_err(y, z["y"]) < 1e-6
    assert rel_err(gx, z["grad_x"]) < 1e-6
    assert rel_err(gwr, z["grad_w_real"]) < 1e-6
    assert rel_err(gwi, z["grad_w_imag"]) < 1e-6
    assert rel_err(gb, z["grad_bias"]) < 1e-6
    if "y_f64" in z:     # the reference's own fp64 evaluation: closed form agrees to fp64 round-off
        assert rel_err(y, z["y_f64"]) < 1e-12
        assert rel_err(gx, z["grad_x_f64"]) < 1e-12
        assert rel_err(gwr, z["grad_w_real_f64"]) < 1e-12
        assert rel_err(gwi, z["grad_w_imag_f64"]) < 1e-12


def test_unused_filter_columns_get_zero_grad():
    z = load_golden("G10_fgtn2_2x256x16")       # F=200 > N//2=128
    k = so.num_bins(256, 200)
    assert k == 128
    assert not z["grad_w_real"][:, k:].any() and not z["grad_w_imag"][:, k:].any()
    _, gwr, gwi, _ = so.backward_closed(z["x"], z["weight_real"], z["weight_imag"], z["g"])
    assert not gwr[:, k:].any() and not gwi[:, k:].any()


def test_k0_is_bias_only():
    z = load_golden("G07_k0_1x1x4")
    assert so.num_bins(1, 2) == 0
    assert np.allclose(z["y"], np.broadcast_to(z["bias"], z["y"].shape))


def test_known_answer_ysum():
    """spectral_layers.py:290-297 with default init: grad_x == 1 everywhere, norm = sqrt(B*N*D)."""
    z = load_golden("G13_ysum_2x128x256")
    assert np.allclose(z["grad_x"], 1.0, atol=1e-6)
    assert abs(np.linalg.norm(z["grad_x"]) - 256.0) < 1e-3


def test_nolearn_is_identity():
    z = load_golden("G11_nolearn_2x128x32")
    assert rel_err(z["y"], z["x"]) < 1e-6
    assert rel_err(z["grad_x"], z["g"]) < 1e-6


def test_wirtinger_port_and_backward():
    z = load_golden("G12_wirtinger_2x32x16")
    out = so.wirtinger_filter_port(torch.from_numpy(z["x_freq"]), torch.from_numpy(z["w_real"]),
                                   torch.from_numpy(z["w_imag"]))
    assert np.array_equal(out.numpy(), z["out"])
    k = so.num_bins(32, 8)
    w = (z["w_real"] + 1j * z["w_imag"])[:, :k].T[None]
    gx, gw = so.wirtinger_mul_backward(z["x_freq"][:, :k], w, z["g_freq"][:, :k])
    assert rel_err(gx, z["mul_grad_x"]) < 1e-6
    assert rel_err(gw, z["mul_grad_w"]) < 1e-6
    # split into the gradients of the .real / .imag Parameters
    assert rel_err(gw[0].real.T, z["grad_w_real"][:, :k]) < 1e-6
    assert rel_err(gw[0].imag.T, z["grad_w_imag"][:, :k]) < 1e-6


def test_energy_ratio_matches_reference_definition():
    x = np.random.default_rng(0).standard_normal((2, 8, 4)).astype(np.float32)
    assert abs(so.energy_ratio(x, x) - 1.0) < 1e-6


HALF = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "H*.npz")))
HALF_KEYS = ("y", "grad_x", "grad_ln_weight", "grad_ln_bias", "grad_w_real", "grad_w_imag", "grad_bias")


def test_half_block_cases_exist():
    assert len(HALF) >= 6


@pytest.mark.parametrize("name", HALF)
def test_block_half_port_is_bit_exact_with_reference(name):
    """x + spectral_mix(norm1(x)) of the reference block (spectral_layers.py:185) and its grads."""
    z = load_golden(name)
    t = lambda k: torch.from_numpy(z[k])
    out = so.block_half_port(t("x"), t("ln_weight"), t("ln_bias"), float(z["eps"]), t("weight_real"),
                             t("weight_imag"), t("bias"), t("g"))
    for key, got in zip(HALF_KEYS, out):
        assert np.array_equal(got.numpy(), z[key]), key
