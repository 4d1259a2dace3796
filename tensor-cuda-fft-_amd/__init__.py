"""MI355X-native spectral mixing: drop-in for `fft_tensor.spectral_layers.SpectralMixingLayer`
and `fft_tensor.wirtinger_ops` of fricker2025-star/Tensor-Cuda-FFT-.

Import as `tensor_cuda_fft_amd` (the shim at the repo root maps that name onto this directory,
whose on-disk name is not a Python identifier).
"""
from .spectral_layers import HybridSpectralAttention, SpectralMixingLayer, SpectralMLPBlock
from .wirtinger_ops import (ComplexParameter, WirtingerGradient, WirtingerSpectralFilter,
                            spectral_mix_with_filter)
from .functional import (DropoutState, hermitian_scale, irfft, pruned_rfft, rfft, rfft_bins, seq_fft,
                         spectral_block_mix, spectral_filter, spectral_mix)
from .spectral_enhancements import MultiScaleSpectralFeatures, PhaseAwareSpectralMixing
from .complex_rope import ComplexRoPE, ComplexRoPESpectralLayer, GatedLinearUnit
from .frequency_ops import FrequencyAttention
from .fixed_spectral import FixedSpectralBlock, FrequencyConvFunc, causal_spectral_conv
from .frequency_native import BicameralBlock, FrequencyNativeBlock, PhaseShift, SpectralFFN, SpectralLayerNorm
from .distributed import GradSync, attach_grad_sync, all_reduce_grads, shard_batch

__all__ = [
    "SpectralMixingLayer", "SpectralMLPBlock", "HybridSpectralAttention", "ComplexParameter", "WirtingerGradient",
    "WirtingerSpectralFilter", "spectral_mix_with_filter", "spectral_mix", "spectral_block_mix",
    "pruned_rfft", "DropoutState", "spectral_filter", "rfft_bins", "seq_fft", "hermitian_scale",
    "PhaseAwareSpectralMixing", "MultiScaleSpectralFeatures", "ComplexRoPE", "GatedLinearUnit",
    "ComplexRoPESpectralLayer", "FrequencyAttention", "FixedSpectralBlock", "FrequencyConvFunc",
    "causal_spectral_conv", "rfft", "irfft", "FrequencyNativeBlock", "BicameralBlock", "PhaseShift", "SpectralFFN",
    "SpectralLayerNorm",
    "GradSync", "attach_grad_sync", "all_reduce_grads", "shard_batch",
]
__version__ = "0.2.0"
