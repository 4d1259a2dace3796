"""Drop-in mirrors of the twins of FixedSpectralBlock that work ON the spectrum between the two transforms
(SURVEY 8f-2): `fft_lm.frequency_native.{PhaseShift, SpectralLayerNorm, SpectralFFN, FrequencyNativeBlock}`
(reference fft_lm/frequency_native.py:22-77, :203-239, :124-200, :242-362) and `fft_lm.bicameral.BicameralBlock`
(reference fft_lm/bicameral.py:26-278).  Same constructors, attribute names and state_dict keys, so reference
checkpoints load unchanged.

What is native here: the two transforms every call makes -- rfft of the zero-padded (B, T, C) activations and
irfft cropped back to T rows (functional.rfft / functional.irfft: smx_rfft_ex / smx_irfft_ex, differentiable) --,
the gate chain between them (kernel spectrum x gain x frequency gate x context gate x cutoff mask:
functional.spectral_gate = smx_spectral_gate_*, one launch each way), SpectralLayerNorm (smx_spectral_ln_*), the
feed-forward's PhaseShift and residual on (2, B, F, C) planes (smx_planar_*), the per-(bin, channel) complex multiply
of PhaseShift elsewhere (smx_cmul / smx_cmul_grad_w) and BicameralBlock's depthwise time path (smx_dwconv3_*).
What stays in torch: the nn.Linear layers (fp32 GEMMs), the (B, T, C) LayerNorms and residual adds around them.

PhaseShift: the reference splits z into (abs, angle), adds the learned rotation, scales the magnitude and rebuilds
the number (:62-77).  |z| m e^{i (arg z + r)} = z * (m e^{i r}) for every z (including 0), so the mirror multiplies
by the (F, C) complex factor  m e^{i r},  m = 1 + 0.1 tanh(magnitude_logits),  r = pi tanh(phase_weights)  --
one broadcast complex multiply with the Wirtinger backward, no abs / angle / exp over the (B, F, C) tensor.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from .fixed_spectral import FrequencyConvFunc, _kernel_dft, cutoff_mask, next_pow2
from .wirtinger_ops import WirtingerGradient


def _unit(phase: torch.Tensor) -> torch.Tensor:
    return torch.complex(torch.cos(phase), torch.sin(phase))


class PhaseShift(nn.Module):
    """Learned per-(bin, channel) phase rotation and near-unity magnitude factor (reference :22-77)."""

    def __init__(self, d_model: int, n_freqs: int):
        super().__init__()
        self.d_model = d_model
        self.n_freqs = n_freqs
        self.phase_weights = nn.Parameter(torch.randn(n_freqs, d_model) * 0.01)
        self.magnitude_logits = nn.Parameter(torch.zeros(n_freqs, d_model))

    def factor(self, bins: int) -> torch.Tensor:
        rot = torch.tanh(self.phase_weights[:bins]) * math.pi                        # :66
        mag = 1.0 + 0.1 * torch.tanh(self.magnitude_logits[:bins])                   # :70
        return (mag * _unit(rot)).to(torch.complex64)

    def forward(self, z_freq: torch.Tensor) -> torch.Tensor:
        z = z_freq.to(torch.complex64)
        return WirtingerGradient.apply(z, self.factor(z.size(1)).unsqueeze(0))       # (B, F, C) x (1, F, C)


class SpectralLayerNorm(nn.Module):
    """Magnitudes normalised across channels per (batch, bin), phases kept (reference :203-239)."""

    def __init__(self, d_model: int, n_freqs: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(n_freqs, d_model))
        self.beta = nn.Parameter(torch.zeros(n_freqs, d_model))

    def forward(self, x_freq: torch.Tensor) -> torch.Tensor:
        bins = x_freq.size(1)
        if (x_freq.is_cuda and x_freq.dtype == torch.complex64 and x_freq.dim() == 3
                and self.gamma.dtype == torch.float32 and Fn.spectral_layer_norm_supported(x_freq.size(2))):
            # one native launch each way instead of ~12 / ~25 elementwise passes over the (B, F, C) spectrum
            return Fn.spectral_layer_norm(x_freq, self.gamma[:bins], self.beta[:bins], self.eps)
        mag = x_freq.abs()
        mean = mag.mean(dim=-1, keepdim=True)
        var = mag.var(dim=-1, keepdim=True, unbiased=False)
        scaled = (mag - mean) * torch.rsqrt(var + self.eps) * self.gamma[:bins] + self.beta[:bins]    # :228-233
        return scaled * _unit(x_freq.angle())                                        # :236 (a zero keeps angle 0)


def _linear_re_im(lin: nn.Linear, z: torch.Tensor) -> torch.Tensor:
    """lin applied to the real and the imaginary part separately, bias on both (reference :167-172, :187-189)."""
    h = lin(torch.view_as_real(z).movedim(-1, 0))                                    # (2, B, F, H)
    return torch.complex(h[0], h[1])


class SpectralFFN(nn.Module):
    """Feed-forward that stays in the frequency domain (reference :124-200)."""

    def __init__(self, d_model: int, n_freqs: int, expansion: int = 2, dropout: float = 0.1):
        super().__init__()
        self.d_model = d_model
        self.n_freqs = n_freqs
        hidden = d_model * expansion
        self.ln = SpectralLayerNorm(d_model, n_freqs)
        self.w1 = nn.Linear(d_model, hidden)
        self.activation = PhaseShift(hidden, n_freqs)
        self.w2 = nn.Linear(hidden, d_model)
        self.dropout_p = dropout
        for lin in (self.w1, self.w2):
            nn.init.normal_(lin.weight, mean=0.0, std=0.01)
            nn.init.zeros_(lin.bias)

    def _planar_ok(self, x_freq: torch.Tensor) -> bool:
        return (x_freq.is_cuda and x_freq.dtype == torch.complex64 and x_freq.dim() == 3
                and self.w1.weight.dtype == torch.float32 and Fn.spectral_layer_norm_supported(x_freq.size(2)))

    def residual(self, x_freq: torch.Tensor) -> torch.Tensor:
        """x_freq + self(x_freq) (the block's use, reference :355-356).  On the device the feed-forward runs on
        (2, B, F, C) float32 planes -- what "the same Linear on the real and on the imaginary part" (:167-172, :187-189)
        wants: SpectralLayerNorm writes the planes, both nn.Linear see contiguous rows, PhaseShift is a native planar
        complex multiply, and the residual add folds the planes back into a complex tensor -- no (de)interleaving
        copies and no complex elementwise passes in between."""
        if not self._planar_ok(x_freq):
            return x_freq + self(x_freq)
        bins = x_freq.size(1)
        p = Fn.spectral_layer_norm(x_freq, self.ln.gamma[:bins], self.ln.beta[:bins], self.ln.eps, planar=True)
        h = self.w1(p)                                                                # (2, B, F, H), bias on both planes
        fac = self.activation.factor(bins)                                            # (F, H) complex, :62-70
        h = Fn.planar_cmul(h, fac.real.contiguous(), fac.imag.contiguous())           # :175
        if self.training and self.dropout_p > 0:                                      # :178-182: one mask for both planes
            h = h * F.dropout(torch.ones_like(h[0]), p=self.dropout_p, training=True)
        return Fn.add_planar(x_freq, self.w2(h))

    def forward(self, x_freq: torch.Tensor) -> torch.Tensor:
        h = self.activation(_linear_re_im(self.w1, self.ln(x_freq)))
        if self.training and self.dropout_p > 0:                                      # :178-182: on the magnitude
            keep = F.dropout(torch.ones_like(h.real), p=self.dropout_p, training=True)
            h = h * keep
        return _linear_re_im(self.w2, h)


def _kernel_spectrum(kernel: torch.Tensor, n_fft: int) -> torch.Tensor:
    """rfft of the zero-padded taps (reference :320-322): a (bins x taps) matrix-vector product, differentiable.
    The DC and Nyquist bins of a real sequence are real: exact +0 imaginary parts, as an r2c transform returns."""
    cm, sm = _kernel_dft(n_fft, kernel.shape[0], kernel.device)
    h_im = sm @ kernel
    real_bin = torch.zeros(h_im.shape[0], dtype=torch.bool, device=h_im.device)
    real_bin[0] = True
    if n_fft % 2 == 0:
        real_bin[n_fft // 2] = True
    return torch.complex(cm @ kernel, torch.where(real_bin, torch.zeros_like(h_im), h_im))


class FrequencyNativeBlock(nn.Module):
    """Pre-norm block whose mixing, gating and feed-forward all happen on the spectrum (reference :242-362)."""

    def __init__(self, d_model: int, seq_len: int, kernel_len: int, transition_bins: int, dropout: float = 0.1):
        super().__init__()
        self.d_model = d_model
        self.seq_len = seq_len
        self.kernel_len = kernel_len
        self.transition_bins = int(max(1, transition_bins))
        self.max_freq_bins = next_pow2(seq_len + kernel_len - 1) // 2 + 1
        self.ln = nn.LayerNorm(d_model)
        self.kernel = nn.Parameter(torch.zeros(kernel_len))
        nn.init.normal_(self.kernel, mean=0.0, std=0.001)
        self.gain = nn.Parameter(torch.ones(d_model))
        self.gate_freq_logits = nn.Parameter(torch.ones(self.max_freq_bins) * 2.0)
        self.gate_ctx = nn.Linear(d_model, d_model)
        nn.init.zeros_(self.gate_ctx.weight)
        nn.init.constant_(self.gate_ctx.bias, 2.0)
        self.ffn = SpectralFFN(d_model, self.max_freq_bins, expansion=2, dropout=dropout)
        self.drop = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, cutoff: "int | None" = None) -> torch.Tensor:
        residual = x
        x = self.ln(x)
        T = x.shape[1]
        n_fft = next_pow2(T + self.kernel_len - 1)                                   # :308-311
        x_freq = Fn.rfft(x, n_fft)                                                   # :314-317, zero-padded load
        bins = x_freq.size(1)
        g_freq = torch.sigmoid(self.gate_freq_logits[:bins])                         # :331
        g_ctx = torch.sigmoid(self.gate_ctx(x.mean(dim=1)))                          # :334-335
        mask = cutoff_mask(cutoff, bins, self.transition_bins, x_freq.device)        # :341-350
        if x_freq.is_cuda and x_freq.shape[2] % 2 == 0 and self.gain.dtype == torch.float32:
            # :325 (FrequencyConvFunc, with its hand-written gradients), :338 and :351 in one native launch each way.
            # SpectralLayerNorm takes arg() of every bin, including the ones the mask has just zeroed (:223, :236),
            # where it is 0 or pi by the SIGN of the zeros: the kernel multiplies in the reference's order and treats a
            # real factor as torch does (promoted to complex), so every masked bin -- the DC and Nyquist rows with their
            # exactly-zero imaginary parts included -- ends with the signs it has there (fixtures T02, T03).
            y_freq = Fn.spectral_gate(x_freq, _kernel_spectrum(self.kernel, n_fft), self.gain, g_freq, g_ctx, mask,
                                      reference_gain_grad=True)
        else:
            k_freq = _kernel_spectrum(self.kernel, n_fft)
            y_freq = FrequencyConvFunc.apply(x_freq, k_freq, self.gain)              # :325
            y_freq = y_freq * (g_freq.view(1, -1, 1) * g_ctx.unsqueeze(1))           # :338
            if mask is not None:
                y_freq = y_freq * mask.view(1, -1, 1)                                # :351
                # the DC and Nyquist rows in the reference's order of multiplications (the sign of their zero
                # imaginary parts depends on it; B x 2 x C numbers)
                sel = [0, n_fft // 2]
                r = x_freq[:, sel] * k_freq[sel].view(1, -1, 1) * self.gain.view(1, 1, -1)            # :95
                r = r * g_freq[sel].view(1, -1, 1) * g_ctx.unsqueeze(1)                               # :338
                y_freq[:, sel] = r * mask[sel].view(1, -1, 1)                                         # :351
        y_freq = self.ffn.residual(y_freq)                                           # :355-356
        y = Fn.irfft(y_freq, n_fft, T)                                               # :359-360, cropped store
        return residual + self.drop(y)


class BicameralBlock(nn.Module):
    """Frequency path (global, follows the curriculum cutoff) + time path (depthwise causal Conv1d, always full
    bandwidth), fused by learned weights and a cross-talk projection (reference fft_lm/bicameral.py:26-278)."""

    def __init__(self, d_model: int, seq_len: int, kernel_len: int, transition_bins: int, dropout: float = 0.1):
        super().__init__()
        self.d_model = d_model
        self.seq_len = seq_len
        self.kernel_len = kernel_len
        self.transition_bins = int(max(1, transition_bins))
        self.ln = nn.LayerNorm(d_model)
        self.max_freq_bins = next_pow2(seq_len + kernel_len - 1) // 2 + 1
        # frequency path
        self.kernel_freq = nn.Parameter(torch.zeros(kernel_len))
        nn.init.normal_(self.kernel_freq, mean=0.0, std=0.001)
        self.gain_freq = nn.Parameter(torch.ones(d_model))
        self.gate_freq_logits = nn.Parameter(torch.ones(self.max_freq_bins) * 2.0)
        self.gate_ctx_freq = nn.Linear(d_model, d_model)
        nn.init.zeros_(self.gate_ctx_freq.weight)
        nn.init.constant_(self.gate_ctx_freq.bias, 2.0)
        self.phase_activation = PhaseShift(d_model, self.max_freq_bins)
        # time path
        self.conv1d = nn.Conv1d(d_model, d_model, kernel_size=3, padding=1, groups=d_model)
        nn.init.normal_(self.conv1d.weight, mean=0.0, std=0.01)
        nn.init.zeros_(self.conv1d.bias)
        self.gate_time = nn.Linear(d_model, d_model)
        nn.init.zeros_(self.gate_time.weight)
        nn.init.constant_(self.gate_time.bias, 2.0)
        # fusion
        self.alpha_freq = nn.Parameter(torch.tensor(0.5))
        self.alpha_time = nn.Parameter(torch.tensor(0.5))
        self.cross_interact = nn.Linear(d_model * 2, d_model)
        nn.init.normal_(self.cross_interact.weight, mean=0.0, std=0.01)
        nn.init.zeros_(self.cross_interact.bias)
        hidden = d_model * 2
        self.ffn_ln = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(nn.Linear(d_model, hidden), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden, d_model))
        for m in self.ffn:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                nn.init.zeros_(m.bias)
        self.drop = nn.Dropout(dropout)

    def frequency_path(self, x: torch.Tensor, pooled: torch.Tensor, cutoff) -> torch.Tensor:
        T = x.shape[1]
        n_fft = next_pow2(T + self.kernel_len - 1)                                   # :166-168
        x_freq = Fn.rfft(x, n_fft)                                                   # :170-171
        # kernel spectrum x gain x frequency gate x context gate (:179-186): one (B, F, C) multiply
        k_freq = _kernel_spectrum(self.kernel_freq, n_fft)
        bins = x_freq.size(1)
        per_f = k_freq * torch.sigmoid(self.gate_freq_logits[:bins])
        per_bc = self.gain_freq.unsqueeze(0) * torch.sigmoid(self.gate_ctx_freq(pooled))
        if x_freq.is_cuda and x_freq.shape[2] % 2 == 0 and per_bc.dtype == torch.float32:
            y_freq = Fn.spectral_gate(x_freq, per_f, q=per_bc)
        else:
            y_freq = x_freq * (per_f.view(1, -1, 1) * per_bc.unsqueeze(1))
        mask = cutoff_mask(cutoff, bins, self.transition_bins, y_freq.device)        # :193-202
        if mask is None:
            y_freq = self.phase_activation(y_freq)                                   # :189
        else:
            # :189 and :203 in one pass: the cutoff mask (real, per bin) folded into PhaseShift's (bin, channel) factor
            fac = self.phase_activation.factor(bins) * mask.view(-1, 1)
            y_freq = WirtingerGradient.apply(y_freq.to(torch.complex64), fac.unsqueeze(0))
        return Fn.irfft(y_freq, n_fft, T)                                            # :206-207

    def time_path(self, x: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dtype == torch.float32 and self.conv1d.weight.dtype == torch.float32:
            # one native launch on (B, T, C): shift, depthwise three-tap convolution and the time gate (:214-227) --
            # through MIOpen the Conv1d and its two transposes were most of this block's time
            return Fn.causal_dwconv3(x, self.conv1d.weight, self.conv1d.bias, torch.sigmoid(self.gate_time(pooled)))
        xc = x.transpose(1, 2)
        shifted = F.pad(xc[:, :, :-1], (1, 0))                                       # :221
        y = self.conv1d(shifted).transpose(1, 2)                                     # :222-223
        return y * torch.sigmoid(self.gate_time(pooled)).unsqueeze(1)                # :226-227

    def forward(self, x: torch.Tensor, cutoff: "int | None" = None) -> torch.Tensor:
        residual = x
        x = self.ln(x)
        pooled = x.mean(dim=1)
        y_spectral = self.frequency_path(x, pooled, cutoff)
        y_time = self.time_path(x, pooled)
        a_f, a_t = torch.sigmoid(self.alpha_freq), torch.sigmoid(self.alpha_time)    # :240-246
        total = a_f + a_t + 1e-8
        C = x.shape[-1]
        native = (x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0 and x.numel() > 0
                  and self.cross_interact.weight.dtype == torch.float32 and not (self.training and self.drop.p > 0))
        if native:
            # :254-255 without the (B, T, 2C) concatenation: the two halves of the cross-talk weight on the two paths,
            # the second product accumulated into the first; then :261-268 (weighted paths + 0.1 x cross-talk +
            # residual) in one native launch each way
            Wc = self.cross_interact.weight
            cross = torch.addmm(F.linear(y_spectral.reshape(-1, C), Wc[:, :C], self.cross_interact.bias),
                                y_time.reshape(-1, C), Wc[:, C:].t()).view_as(x)
            out = Fn.mix_paths(residual, y_spectral, y_time, cross, torch.stack([a_f / total, a_t / total]), 0.1)
        else:
            y = (a_f / total) * y_spectral + (a_t / total) * y_time                  # :261
            y = y + 0.1 * self.cross_interact(torch.cat([y_spectral, y_time], dim=-1))   # :254-265
            out = residual + self.drop(y)
        return out + self.drop(self.ffn(self.ffn_ln(out)))                           # :272-273
