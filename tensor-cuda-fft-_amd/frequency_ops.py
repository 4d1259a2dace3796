"""Mirror of the one member of `fft_tensor.frequency_ops` that touches the sequence transform:
`FrequencyAttention.fnet_attention` (reference fft_tensor/frequency_ops.py:188-204) -- a full complex FFT
along the sequence axis of a complex (B, N, D) tensor, here the native transform (functional.seq_fft).
The rest of that module (FrequencyMatMul, ComplexSemanticEmbedding, ...) works on SparseSpectralTensor
weights and is out of scope."""
from __future__ import annotations

import torch

from .functional import seq_fft


class FrequencyAttention:
    @staticmethod
    def fnet_attention(x_freq: torch.Tensor) -> torch.Tensor:
        """torch.fft.fft(x_freq, dim=1) for complex64 (B, N, D) on a ROCm device."""
        return seq_fft(x_freq)
