"""Host-side mirror of the reference interface: constructors, state_dict, errors (no GPU)."""
import math

import pytest
import torch

import tensor_cuda_fft_amd as pkg
from tensor_cuda_fft_amd.distributed import shard_batch


def test_layer_constructor_and_state_dict_match_reference_contract():
    layer = pkg.SpectralMixingLayer(256)
    assert layer.embed_dim == 256 and layer.num_filters == 128 and layer.learnable
    sd = layer.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "weight_real": (256, 128), "weight_imag": (256, 128), "bias": (256,)}
    assert torch.all(layer.weight_real == 1) and torch.all(layer.weight_imag == 0)
    assert torch.all(layer.bias == 0) and layer._verify_gradients is True
    assert sum(p.numel() for p in layer.parameters()) == 65792            # BENCHMARKS.md:86
    assert pkg.SpectralMixingLayer(10, num_filters=3).num_filters == 3
    assert isinstance(layer.dropout, torch.nn.Dropout) and layer.dropout.p == 0.0


def test_non_learnable_has_no_parameters():
    layer = pkg.SpectralMixingLayer(32, learnable=False)
    assert layer.weight_real is None and layer.weight_imag is None and layer.bias is None
    assert len(layer.state_dict()) == 0
    x = torch.randn(2, 8, 32)
    assert torch.equal(layer(x), x)                                       # identity, any device


def test_errors_mirror_the_reference():
    layer = pkg.SpectralMixingLayer(16)
    with pytest.raises(AssertionError, match="Expected embed_dim=16, got 8"):
        layer(torch.randn(1, 4, 8))
    with pytest.raises(ValueError):                                       # 2-D input cannot unpack
        layer(torch.randn(4, 16))
    with pytest.raises(RuntimeError, match="no CPU implementation"):      # never a silent CPU path
        layer(torch.randn(1, 4, 16))


def test_energy_ratio_method():
    layer = pkg.SpectralMixingLayer(4)
    x = torch.randn(2, 8, 4)
    assert abs(layer.verify_energy_preservation(x, x) - 1.0) < 1e-6
    assert abs(layer.verify_energy_preservation(x, 2 * x) - 4.0) < 1e-5


def test_block_keeps_reference_attribute_names():
    blk = pkg.SpectralMLPBlock(32, mlp_ratio=2, dropout=0.0)
    keys = set(blk.state_dict())
    assert {"spectral_mix.weight_real", "spectral_mix.weight_imag", "spectral_mix.bias",
            "norm1.weight", "norm2.bias", "mlp.0.weight", "mlp.3.bias"} <= keys
    assert blk.mlp[0].out_features == 64


def test_block_fusion_gate():
    """The fused first half is only taken for fp32 GPU input with inactive dropout; a CPU tensor goes to the
    composition, whose native layer then refuses it (no CPU path anywhere)."""
    blk = pkg.SpectralMLPBlock(32, mlp_ratio=2, dropout=0.1)
    x = torch.randn(2, 64, 32)
    assert blk.fuse_norm and not blk._fusable(x)                 # CPU tensor
    sm = blk.spectral_mix
    assert sm._fused_dropout_p() == pytest.approx(0.1)           # training: dropout goes into the native op
    sm.fuse_dropout = False
    assert sm._fused_dropout_p() == 0.0
    sm.fuse_dropout = True
    blk.eval()
    assert sm._fused_dropout_p() == 0.0
    blk.train()
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        blk.eval()(x)
    assert not blk._fusable(torch.randn(2, 64, 16))              # wrong width -> reference's errors


def test_hybrid_attention_keeps_reference_attribute_names():
    m = pkg.HybridSpectralAttention(32, num_heads=4, window_size=16, dropout=0.0)
    assert {"spectral.weight_real", "spectral.weight_imag", "spectral.bias", "qkv.weight", "qkv.bias",
            "proj.weight", "proj.bias", "norm.weight", "norm.bias"} == set(m.state_dict())
    assert (m.embed_dim, m.num_heads, m.window_size) == (32, 4, 16)


def test_complex_parameter_init_modes():
    torch.manual_seed(0)
    p = pkg.ComplexParameter((64, 32), "xavier")
    bound = math.sqrt(3.0 / 96)
    assert p.real.abs().max() <= bound and p.imag.abs().max() <= bound
    p = pkg.ComplexParameter((64, 32), "uniform")
    assert torch.allclose(p.magnitude(), torch.ones(64, 32), atol=1e-5)
    p = pkg.ComplexParameter((8, 4), "ones")
    assert torch.all(p.real == 1) and torch.all(p.imag == 0) and p().dtype == torch.complex64
    assert torch.all(p.phase() == 0)
    p = pkg.ComplexParameter((1000, 10), "kaiming")
    assert abs(p.real.std().item() - math.sqrt(2.0 / 1000)) < 0.005
    with pytest.raises(ValueError, match="Unknown init_mode: nope"):
        pkg.ComplexParameter((2, 2), "nope")


def test_wirtinger_filter_contract():
    f = pkg.WirtingerSpectralFilter(16, 8)
    assert set(f.state_dict()) == {"weight.real", "weight.imag"}
    assert f.num_channels == 16 and f.num_frequencies == 8
    with pytest.raises(AssertionError):
        f(torch.zeros(1, 4, 8, dtype=torch.complex64))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        f(torch.zeros(1, 4, 16, dtype=torch.complex64))


@pytest.mark.parametrize("B,world", [(512, 8), (10, 4), (3, 8), (64, 1)])
def test_shard_batch_partitions_exactly(B, world):
    rows = []
    for r in range(world):
        s = shard_batch(B, r, world)
        rows += list(range(B))[s]
    assert rows == list(range(B))
    sizes = [len(range(B)[shard_batch(B, r, world)]) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("name,cls", [("T01_freqnative_2x192x16", "FrequencyNativeBlock"),
                                      ("T11_bicameral_2x192x16", "BicameralBlock"),
                                      ("F01_fixed_2x192x32", "FixedSpectralBlock")])
def test_fft_lm_blocks_keep_the_reference_state_dict(name, cls):
    """Constructor arguments, parameter names and shapes of the fft_lm blocks (reference fft_lm/train_fixed_full.py:
    427-495, fft_lm/frequency_native.py:251-294, fft_lm/bicameral.py:40-132): a reference checkpoint loads unchanged."""
    import torch
    import tensor_cuda_fft_amd as pkg
    from conftest import load_golden
    z = load_golden(name)
    blk = getattr(pkg, cls)(z["x"].shape[2], seq_len=int(z["seq_len"]), kernel_len=int(z["kernel_len"]),
                            transition_bins=int(z["transition_bins"]), dropout=0.0)
    sd = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd.")}
    assert {k: tuple(v.shape) for k, v in blk.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    blk.load_state_dict(sd)                                    # strict


def test_transform_pair_refuses_cpu_tensors_and_bad_arguments():
    """functional.rfft / irfft have no CPU path (the product fails loudly without the GPU) and check their
    arguments before touching the library."""
    import torch
    from tensor_cuda_fft_amd import functional as Fn
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        Fn.rfft(torch.zeros(1, 8, 2))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        Fn.irfft(torch.zeros(1, 5, 2, dtype=torch.complex64))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        import tensor_cuda_fft_amd as pkg
        pkg.PhaseShift(4, 8)(torch.zeros(1, 8, 4, dtype=torch.complex64))


def test_slow_plans_warn_once_for_large_problems():
    """a large problem on the DFT-product plan (n_fft % 256 != 0) or on band groups says so, once."""
    import warnings
    from tensor_cuda_fft_amd import _lib, functional as Fn
    pytest.importorskip("ctypes")
    if not __import__("os").path.exists(_lib.LIB_PATH):
        pytest.skip("libsmx.so not built")
    Fn._slow_plan_warned.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Fn._note_plan(_lib.plan(64, 4001, 256, 128), 64, 4001, 256, 4001)
        Fn._note_plan(_lib.plan(64, 4001, 256, 128), 64, 4001, 256, 4001)      # second time: silent
        Fn._note_plan(_lib.plan(64, 4000, 256, 128), 64, 4000, 256, 4000)      # 16 | N: sixteen-row decimation, silent
        Fn._note_plan(_lib.plan(2, 100, 8, 4), 2, 100, 8, 100)                  # small: silent
        Fn._note_plan(_lib.plan(64, 4096, 256, 128), 64, 4096, 256, 4096)       # streaming plan: silent
        Fn._note_plan(_lib.plan(64, 8704, 2048, 1024), 64, 8704, 2048, 8704)    # band groups (L = 34)
    assert len(w) == 2 and "not a multiple of 256" in str(w[0].message) and "band groups" in str(w[1].message)
    Fn._slow_plan_warned.clear()
