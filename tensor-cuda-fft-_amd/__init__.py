"""MI355X-native spectral mixing: drop-in for `fft_tensor.spectral_layers.SpectralMixingLayer`
and `fft_tensor.wirtinger_ops` of fricker2025-star/Tensor-Cuda-FFT-.

Import as `tensor_cuda_fft_amd` (the shim at the repo root maps that name onto this directory,
whose on-disk name is not a Python identifier).
"""
from .spectral_layers import HybridSpectralAttention, SpectralMixingLayer, SpectralMLPBlock
from .wirtinger_ops import (ComplexParameter, WirtingerGradient, WirtingerSpectralFilter,
                            spectral_mix_with_filter)
from .functional import DropoutState, pruned_rfft, spectral_block_mix, spectral_mix
from .distributed import GradSync, attach_grad_sync, all_reduce_grads, shard_batch

__all__ = [
    "SpectralMixingLayer", "SpectralMLPBlock", "HybridSpectralAttention", "ComplexParameter", "WirtingerGradient",
    "WirtingerSpectralFilter", "spectral_mix_with_filter", "spectral_mix", "spectral_block_mix",
    "pruned_rfft", "DropoutState",
    "GradSync", "attach_grad_sync", "all_reduce_grads", "shard_batch",
]
__version__ = "0.1.2"
