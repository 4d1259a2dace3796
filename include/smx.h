/* smx.h -- C ABI of the MI355X spectral-mixing library (libsmx.so).
 *
 * The reference (fricker2025-star/Tensor-Cuda-FFT-) is pure PyTorch and has NO FFI for this path;
 * its boundary is the Python nn.Module API.  These entry points are what a native binding of
 * that API binds to; each one names the reference lines it replaces (paths relative to the
 * reference root).  See INTEGRATION.md for the reference-side ctypes stub.
 *
 * Conventions
 *  - Every buffer is DEVICE memory owned by the caller (fp32, contiguous, row-major).
 *    Complex tensors are interleaved (re,im) fp32 pairs == torch.complex64.
 *  - x, y, g, grad_x : (B, N, D).  w_re, w_im : (D, F).  bias : (D).
 *    k = min(F, N/2) kept bins (integer division).  Spectra xk/gk : (B, k, D) complex.
 *  - All work is enqueued on `stream` (a hipStream_t; NULL = default stream).  No host sync,
 *    except the one-time twiddle-table upload the first time a given N is seen on a device.
 *    That upload is REFUSED (SMX_ERR_UNSUPPORTED, nothing enqueued) while `stream` is being
 *    captured into a hipGraph: call smx_prepare(N), or run one eager call of the shape, first.
 *    The table cache holds "table_cache_entries" (default 256) sequence lengths per process;
 *    beyond that the least recently used ones OF THE CALLING DEVICE that no call in flight is using are freed
 *    after a device synchronise (smx_tables_epoch() changes).
 *  - Return value: 0 on success, negative SMX_ERR_* otherwise; text via smx_last_error()
 *    (thread-local).  Nothing throws across this boundary.
 *  - Thread-safe; re-entrant across streams and devices (uses the calling thread's current device).
 */
#ifndef SMX_H_
#define SMX_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_VERSION 303            /* 0.3.3: smx_spectral_gate_* (0.3.2: smx_diag_clock; conv / cfft workspaces behind the sync area) */

#define SMX_OK 0
#define SMX_ERR_INVALID (-1)       /* bad shape / null pointer / misaligned buffer */
#define SMX_ERR_UNSUPPORTED (-2)
#define SMX_ERR_HIP (-3)           /* a HIP runtime call failed; see smx_last_error() */
#define SMX_ERR_WORKSPACE (-4)     /* workspace too small; see smx_workspace_bytes() */

/* which kernels a shape is routed to */
#define SMX_PATH_DECIMATED 1       /* N % 256 == 0, D even, k <= 512: fused Stockham radix-16x16 */
#define SMX_PATH_DIRECT 2          /* everything else: pruned DFT as matrix products, O(N k) per column */
#define SMX_PATH_DECIM16 3         /* n_fft % 16 == 0 (not % 256), D even, k <= 256 (rows <= n_fft zero-padded): one
                                      16-point transform per residue + O(N k / 16) accumulation, x read once,
                                      y written once (k_fused16; fused dropout and every phase split included);
                                      synthesis alone runs the DFT products of SMX_PATH_DIRECT, row_scale is refused */

typedef struct smx_plan {
  int path;        /* SMX_PATH_*                                                      */
  int k;           /* kept bins                                                       */
  int L;           /* decimation factor N/256            (decimated path)             */
  int bands;       /* 1: k <= 128, 2: k <= 256, 4: k > 256 (decimated path)           */
  int nsplit;      /* residue chunks; 1 = single fused launch per direction           */
  int workgroups;  /* workgroups of the transform launch                              */
  int groups;      /* band groups of 512 bins: 1 unless k > 512 (one launch per group) */
} smx_plan;

int smx_version(void);
const char* smx_last_error(void);

/* Tuning knobs, process-wide DEFAULTS: "nsplit" (0 = auto), "placement" (workgroup -> tile map: 0 b-major,
 * 1 rotated residues, 2 XCD-aware = default, 3 | a << 8 | b << 16 = XCD-aware with the residue rotation
 * (a l2 + b d_tile) mod L, used by tools/rot_scan.py), "round" (workgroups per launch of the streaming
 * kernels, default 512 = one resident round; 0 = a single launch), "force_direct" (0/1), "full8", "fourstep",
 * "fs_bgroups", "fold_gradw", "tiled_dft", "decim16", "conv1", "st_plain" (A/B switches of DESIGN.md), "table_cache_entries" (twiddle-table cache bound). */
int smx_set_option(const char* name, int value);

/* The same knobs as an argument of the calling context: every call THIS THREAD makes between
 * smx_options_push(opts) and the matching smx_options_pop() plans with *opts instead of the process-wide
 * defaults (nestable, 8 deep).  Two users of one process can therefore run different plans, and a plan query /
 * workspace size / forward / backward sequence made under one push is consistent by construction.
 * smx_options_default fills *out with the current defaults (start from it, change what you need).
 * smx_options_epoch() changes whenever a process-wide default changes: anything memoised per shape outside a
 * push (plan, workspace size) is valid for one epoch.  smx_tables_epoch() changes whenever twiddle tables are
 * evicted from the cache: a shape "prepared" for stream capture before that may need smx_prepare again. */
typedef struct smx_options {
  int nsplit, placement, round, force_direct, full8, fourstep, fs_bgroups;
  int fold_gradw;   /* 0 (default): separate parameter-gradient reduction launch (k_gradw); 1: smx_backward with
                       SMX_PHASE_ALL on the single-launch plan reduces them inside the transform launch --
                       bit-identical, measured slower on MI355X, kept as an A/B switch */
  int decim16;      /* 1 (default): SMX_PATH_DECIM16 is used where it applies; 0: DFT products (A/B, tests) */
  int conv1;        /* 1 (default): smx_conv_* run ONE launch per direction for n_fft = 512, 1024, 2048
                       (k_conv1) from 48 (batch row, 32-channel tile) items on -- as 256-thread
                       workgroups on 16 channels up to 768 items, 512-thread workgroups on 32 channels above;
                       2 / 3: wherever the shape allows it with 32- / 16-channel workgroups; 0: the three launches
                       of the four-step form (A/B, tests).  The layout of x_spectra differs between the forms:
                       forward and backward of one call pair must run under the same value */
  int st_plain;     /* rows (of the 16 a thread writes per tile) that the streaming kernels store with the DEFAULT
                       write-back policy instead of the streaming hint: -1 (default) = by the size of the output
                       tensor, about 64 MiB of it (4 rows up to 320 MiB, 2 up to 768 MiB, 1 up to 1.5 GiB, else 0);
                       0, 1, 2, 4 = that many (A/B).  DESIGN.md section 4, round 4 */
} smx_options;
int smx_options_default(smx_options* out);
int smx_options_push(const smx_options* opts);
int smx_options_pop(void);
unsigned long long smx_options_epoch(void);
unsigned long long smx_tables_epoch(void);

/* Compile-time switches of this binary that change what the kernels compute ("" for the shipped build).
 * A name starting with SMX_AB_ marks a timing-ablation build that returns wrong results by design. */
const char* smx_build_flags(void);

/* Diagnostic (no reference counterpart): the shader clock the chip sustains right now.  One wavefront reads
 * s_memtime (core-clock ticks) and s_memrealtime (the constant 100 MHz counter) around `spin` dependent FMAs and
 * stores both differences: out2[0] / out2[1] * 0.1 = GHz.  bench.py enqueues it right behind its timed region and
 * stamps the figure on its line, so a slow box can be told from a slow build (the fused launches follow the clock:
 * DESIGN.md section 4).  out2: two 64-bit words in device memory. */
int smx_diag_clock(unsigned long long* out2, int spin, void* stream);

int smx_plan_query(int B, int N, int D, int F, smx_plan* out);
int smx_workspace_bytes(int B, int N, int D, int F, size_t* out);
int smx_prepare(int N);

/* y = real(ifft(pad_k(W .* fft(x)[:k]))) + bias
 *   replaces SpectralMixingLayer.forward, fft_tensor/spectral_layers.py:88-116.
 *   bias may be NULL.  xk_save (B,k,D) complex may be NULL; when given it receives fft(x)[:, :k, :]
 *   (what backward needs -- the reference keeps the whole x_freq alive instead, :88).
 *   conj_w != 0 multiplies by conj(W): with bias == NULL that is grad_x of the same layer. */
int smx_forward(const float* x, const float* w_re, const float* w_im, const float* bias, float* y,
                float* xk_save, void* workspace, size_t workspace_bytes, int B, int N, int D, int F,
                int conj_w, void* stream);

/* Autograd backward of the lines above for upstream gradient g:
 *   grad_x = real(ifft(pad_k(conj(W) .* fft(g)[:k]))),
 *   grad_w_real[d,f] = Re P, grad_w_imag[d,f] = -Im P, P[f,d] = (1/N) sum_b X conj(G)   (f < k, else 0),
 *   grad_bias[d] = sum_{b,n} g.
 * phases (bit mask, SMX_PHASE_ALL = one fused launch when the plan allows):
 *   SMX_PHASE_SPECTRUM  transform g, filter, per-batch-row products X conj(G)   (into the workspace)
 *   SMX_PHASE_INVERSE   inverse transform to grad_x                             (needs SPECTRUM first)
 *   SMX_PHASE_PARAMS    reduce the products to gw_re / gw_im / gbias           (needs SPECTRUM first)
 *   Separate calls on the same workspace let a multi-GPU caller run PARAMS + the all-reduce of the
 *   parameter gradients on a side stream while INVERSE runs on the main one.
 * gw_re / gw_im / gbias may be NULL together (input gradient only; PARAMS is then a no-op, but pass
 * them to the SPECTRUM call too when PARAMS will follow). */
#define SMX_PHASE_SPECTRUM 1
#define SMX_PHASE_INVERSE 2
#define SMX_PHASE_PARAMS 4
#define SMX_PHASE_ALL 7
/* With option fold_gradw = 1 and SMX_PHASE_ALL on the single-launch plan (k <= 256) the parameter gradients are
 * reduced INSIDE the transform launch (reduction workgroups appended to its grid), which needs a few flag words of the workspace -- its first
 * 64 KiB are reserved for them in every layout -- to be zero when the launch starts.  The library clears them itself (one small hipMemsetAsync per call) unless the caller
 * ORs SMX_PHASE_SYNC_CLEAN into `phases`, vouching that the last thing that touched this workspace was
 * an smx_forward / smx_block_forward of the same (B, D) that was given the workspace (it leaves those words zero),
 * or a completed smx_backward (it leaves them zero as well), and that nothing else wrote to those 64 KiB.  The Python autograd functions hand
 * forward's workspace to backward and set the flag. */
#define SMX_PHASE_SYNC_CLEAN 8
int smx_backward(const float* g, const float* xk, const float* w_re, const float* w_im,
                 float* grad_x, float* gw_re, float* gw_im, float* gbias, void* workspace,
                 size_t workspace_bytes, int B, int N, int D, int F, int phases, void* stream);

/* xk = fft(x, dim=1)[:, :k, :]   (spectral_layers.py:88 restricted to the bins :94-101 keep) */
int smx_spectrum(const float* x, float* xk, void* workspace, size_t workspace_bytes, int B, int N,
                 int D, int F, void* stream);

/* Parameter gradients from two saved spectra (same formulas as smx_backward). */
int smx_grad_w(const float* xk, const float* gk, float* gw_re, float* gw_im, float* gbias, int B,
               int N, int D, int F, void* stream);

/* WirtingerSpectralFilter.forward, fft_tensor/wirtinger_ops.py:170-203:
 *   out[b,n,d] = n < k ? x_freq[b,n,d] * (w_re + i w_im)[d,n] : 0      (complex (B,N,D) in/out)
 *   conj_w != 0 gives the filter's grad_x (wirtinger_ops.py:71). */
int smx_wfilter_forward(const float* x_freq, const float* w_re, const float* w_im, float* out, int B,
                        int N, int D, int F, int conj_w, void* stream);
/* grad of the filter weights: sum_b g * conj(x) on bins < k (wirtinger_ops.py:77-80), split into
 * the gradients of ComplexParameter.real / .imag (wirtinger_ops.py:132-134); columns >= k are zero. */
int smx_wfilter_grad_w(const float* x_freq, const float* g_freq, float* gw_re, float* gw_im, int B,
                       int N, int D, int F, void* stream);

/* WirtingerGradient, fft_tensor/wirtinger_ops.py:45-50 and :67-82, for w broadcast over the
 * leading dimension: x (batch, inner) complex, w (inner) complex.
 *   smx_cmul:        out = x * w          (conj_w: x * conj(w)  == grad_x)
 *   smx_cmul_grad_w: gw  = sum_batch g * conj(x) */
int smx_cmul(const float* x, const float* w, float* out, long long batch, long long inner,
             int conj_w, void* stream);
int smx_cmul_grad_w(const float* x, const float* g, float* gw, long long batch, long long inner,
                    void* stream);

/* ---- general shapes: zero-padded rows, explicit bin count, Nyquist bin ------------------------------
 * The same transform for the callers either side of the layer:
 *   y[:, :rows] = real(ifft_{n_fft}( pad_k( W .* fft_{n_fft}(zero-pad(x))[:k] ) ))[:, :rows] + bias
 * x, y, g, grad_x : (B, rows, D), rows <= n_fft -- rows beyond `rows` are zeros on the way in and are
 * not written on the way out: the causal FFT convolution of fft_lm.FixedSpectralBlock (reference
 * fft_lm/train_fixed_full.py:507-519 zero-pad to a power of two, :553-555 crop).
 * k <= min(F, n_fft/2 + 1) kept bins, spectra (B, k, D); k = n_fft/2 + 1 is the full one-sided spectrum
 * incl. the Nyquist bin, of which -- as of DC -- only Re(W X) reaches the real output.  The convention
 * stays the layer's (one-sided spectrum, real part: bins 1 .. n_fft/2 - 1 count once); torch.fft.irfft
 * semantics (reference spectral_enhancements.py:164, train_fixed_full.py:553) = the same call with W
 * doubled on those bins, which the Python callers do (functional.hermitian_weights).
 * With rows = n_fft and k = min(F, n_fft/2) these are smx_forward / smx_backward / smx_spectrum (no
 * fused dropout here).  filter_pack as below. */
typedef struct smx_shape {
  int B;        /* batch                                    */
  int rows;     /* rows of x / y / g / grad_x               */
  int D;        /* channels                                 */
  int F;        /* columns of w_re / w_im (D, F)            */
  int n_fft;    /* transform length, >= rows                */
  int k;        /* kept bins, <= min(F, n_fft/2 + 1)        */
} smx_shape;
int smx_plan_query_ex(const smx_shape* shape, smx_plan* out);
int smx_workspace_bytes_ex(const smx_shape* shape, size_t* out);
/* row_scale (may be NULL): (B, D) real factors, W_eff[b,d,f] = W[d,f] * row_scale[b,d] -- the context gate
 * of FixedSpectralBlock (train_fixed_full.py:532-536: g_ctx multiplies every bin of a (batch, channel)
 * column) applied inside the filter stage instead of as a pass over y.  Backward given the same row_scale
 * returns gw_re / gw_im with the factor included and, in grad_row_scale (B, D; may be NULL),
 * d/d row_scale = sum_n g * y0 with y0 the unscaled output.  Available where smx_row_scale_supported()
 * says so (every decimated plan except the band groups); SMX_ERR_UNSUPPORTED otherwise. */
int smx_row_scale_supported(const smx_shape* shape);
int smx_forward_ex(const smx_shape* shape, const float* x, const float* w_re, const float* w_im,
                   const float* bias, float* y, float* xk_save, void* workspace, size_t workspace_bytes,
                   int conj_w, float* filter_pack, const float* row_scale, void* stream);
int smx_backward_ex(const smx_shape* shape, const float* g, const float* xk, const float* w_re,
                    const float* w_im, float* grad_x, float* gw_re, float* gw_im, float* gbias,
                    void* workspace, size_t workspace_bytes, int phases, const float* filter_pack,
                    const float* row_scale, float* grad_row_scale, void* stream);
/* xk = rfft(zero-pad(x), n_fft)[:, :k, :]   (reference train_fixed_full.py:515-519,
 * spectral_enhancements.py:147, :237; frequency_ops.py:201 via the channel-pair packing) */
int smx_spectrum_ex(const smx_shape* shape, const float* x, float* xk, void* workspace,
                    size_t workspace_bytes, void* stream);

/* torch.fft.fft(z, dim=1) of a COMPLEX (B, rows, D/2) tensor handed over as real (B, rows, D): a channel pair
 * is one complex channel, so the packed spectrum the kernels form is the answer -- out (B, n_fft, D) real =
 * (B, n_fft, D/2) complex, every bin (FrequencyAttention.fnet_attention, fft_tensor/frequency_ops.py:188-204).
 * n_fft = 256 L with L in {2, 4, 5..32, 36..64 step 4, 72..128 step 8, 144..256 step 16} and even D (shape->F, k are ignored; workspace from
 * smx_cfft_workspace_bytes): SMX_ERR_UNSUPPORTED otherwise -- compose it from smx_spectrum_ex there
 * (Z[f] = A[f] + i B[f], Z[n_fft - f] = conj A[f] + i conj B[f]), as functional.seq_fft does. */
int smx_cfft_workspace_bytes(const smx_shape* shape, size_t* out);
int smx_cfft_ex(const smx_shape* shape, const float* z, float* out, void* workspace, size_t workspace_bytes,
                void* stream);

/* The differentiable transform pair of the blocks that work ON the spectrum between the two transforms
 * (reference fft_lm/frequency_native.py:316-317 / :355-356 FrequencyNativeBlock, fft_lm/bicameral.py:170-171 /
 * :203-204 BicameralBlock, fft_tensor/spectral_enhancements.py:152 / :166): with theta = 2 pi f n / n_fft and
 * c_f = 2 for 0 < f < n_fft/2 when `hermitian`, else 1,
 *   smx_rfft_ex   spec[b, f, d] = scale * c_f * sum_{n < rows} x[b, n, d] e^{-i theta}              f < k
 *   smx_irfft_ex  y[b, n, d]    = scale * sum_{f < k} c_f * Re( spec[b, f, d] e^{+i theta} )        n < rows
 * (the imaginary parts of the DC and Nyquist rows do not reach y, as in torch.fft.irfft).
 *   torch.fft.rfft(x, n_fft, dim=1)[:, :k]      = smx_rfft_ex(scale 1, hermitian 0)  (= smx_spectrum_ex)
 *   torch.fft.irfft(spec, n_fft, dim=1)[:, :rows] = smx_irfft_ex(scale 1/n_fft, hermitian 1), k = n_fft/2 + 1
 *   backward of rfft:  grad_x = smx_irfft_ex(grad_spec, scale 1, hermitian 0)
 *   backward of irfft: grad_spec = smx_rfft_ex(grad_y, scale 1/n_fft, hermitian 1)
 * spec is (B, k, D) complex64 interleaved, 16-byte aligned; workspace as smx_workspace_bytes_ex(shape).
 * shape.F is only required to be >= k. */
int smx_rfft_ex(const smx_shape* shape, const float* x, float* spec, float scale, int hermitian,
                void* workspace, size_t workspace_bytes, void* stream);
int smx_irfft_ex(const smx_shape* shape, const float* spec, float* y, float scale, int hermitian,
                 void* workspace, size_t workspace_bytes, void* stream);

/* The causal FFT convolution of fft_lm.FixedSpectralBlock with its own kernels (reference
 * fft_lm/train_fixed_full.py:507-555; twin backward fft_lm/frequency_native.py:107-121):
 *   y[b, n, c] = row_scale[b, c] * irfft( rfft(zero-pad(x[b, :, c]), n_fft) * H, n_fft )[n],   n < rows
 * H (n_fft/2 + 1 complex: h_re, h_im) is shared by every channel -- kernel spectrum x frequency gate x cutoff
 * mask -- and row_scale (B, D; may be NULL) carries gain x context gate.  A real kernel convolves the packed
 * channel pair as it convolves each channel, so the packed spectrum is multiplied by the Hermitian extension of
 * H directly: no (D, F) filter, no unpack, no one-sided spectrum or gradient slab.
 * forward writes the tile spectra of x into x_spectra (save_bytes of smx_conv_workspace_bytes; may be NULL when
 * no backward follows); backward needs them and returns
 *   grad_x, grad_row_scale (B, D; may be NULL) = sum_n g * y0 (y0 = output before row_scale), and
 *   grad_h_re / grad_h_im (n_fft/2 + 1 each; may be NULL together) = dL/dRe H, dL/dIm H: with
 *   P[f] = sum over (b, channel pairs) of Zg[f] (sigma conj Zx[f] + delta Zx[-f]) the Hermitian part
 *   Q[f] = (P[f] + conj P[n_fft - f]) / 2 is sum_c s_c conj(X_c) G_c and dL/dH[f] = c_f Q[f] / n_fft, c_f = 2 (1 at
 *   DC and Nyquist, whose imaginary parts do not reach the output and get gradient 0).
 * n_fft = 512, 1024, 2048 (fft_lm's default: seq_len 1024 + 128 taps) run ONE launch per direction (k_conv1, option
 * "conv1"): with rows <= n_fft / 2 the n_fft-point spectrum splits by parity of the bin into two half-length transforms
 * of the same rows, and one workgroup holds both; more rows are first folded onto the lower half (x[n] +/- x[n + n_fft/2]
 * at the load, two output rows per value at the store).  x is read once, y written once; x_spectra then holds the packed
 * spectrum in that kernel's own layout (same save_bytes).  Other lengths: three launches through a workspace.
 * Shapes: shape->{B, rows, D, n_fft}; F and k are ignored.  Available for n_fft = 512 ... 65536 (powers
 * of two) with even D (smx_conv_supported); other lengths go through smx_forward_ex with W[c, f] = c_f H[f] gain[c] and
 * row_scale. */
int smx_conv_supported(const smx_shape* shape);
int smx_conv_workspace_bytes(const smx_shape* shape, size_t* workspace_bytes, size_t* save_bytes);
int smx_conv_forward(const smx_shape* shape, const float* x, const float* h_re, const float* h_im,
                     const float* row_scale, float* y, float* x_spectra, void* workspace,
                     size_t workspace_bytes, void* stream);
int smx_conv_backward(const smx_shape* shape, const float* g, const float* x_spectra, const float* h_re,
                      const float* h_im, const float* row_scale, float* grad_x, float* grad_h_re,
                      float* grad_h_im, float* grad_row_scale, void* workspace, size_t workspace_bytes,
                      void* stream);

/* The response the block hands to smx_conv_*, in one launch (reference fft_lm/train_fixed_full.py:511-513 k_freq,
 * :529 frequency gate, :540-551 cutoff mask):
 *   H[f] = rfft(zero-pad(kernel[0 .. taps), n_fft))[f] * sigmoid(gate_logits[f]) * mask[f],   f <= n_fft / 2
 * gate_logits (at least n_fft/2 + 1 entries) and mask (n_fft/2 + 1) may each be NULL (= factor 1).
 * backward: from grad_h_re / grad_h_im (n_fft/2 + 1, e.g. smx_conv_backward's) the gradients of the taps
 * (grad_kernel, taps; may be NULL) and of the logits (grad_gate_logits, n_logits entries, zero from n_fft/2 + 1 on;
 * may be NULL).  Deterministic (fixed summation order). */
int smx_conv_response(int n_fft, int taps, const float* kernel, const float* gate_logits, const float* mask,
                      float* h_re, float* h_im, void* stream);
int smx_conv_response_backward(int n_fft, int taps, int n_logits, const float* kernel, const float* gate_logits,
                               const float* mask, const float* grad_h_re, const float* grad_h_im, float* grad_kernel,
                               float* grad_gate_logits, void* stream);

/* The filter of PhaseAwareSpectralMixing (reference fft_tensor/spectral_enhancements.py:147-164: magnitude * m[d],
 * phase + p[d] on every bin, then irfft) as the (D, k) arrays smx_forward_ex takes:
 *   W[d, f] = c_f * magnitude[d] * exp(i * phase[d]),  f < k <= n_fft/2 + 1;  c_f = 2, 1 at DC and at the Nyquist bin of
 *   an even n_fft (torch.fft.irfft's weights).
 * backward: grad_magnitude[d] = sum_f c_f (gW_re cos p + gW_im sin p), grad_phase[d] = m[d] sum_f c_f (gW_im cos p -
 * gW_re sin p) from the (D, row_pitch) gradient arrays of smx_backward_ex (either output may be NULL). */
int smx_phase_filter(const float* magnitude, const float* phase, int D, int k, int n_fft, float* w_re, float* w_im,
                     void* stream);
int smx_phase_filter_backward(const float* magnitude, const float* phase, const float* grad_w_re, const float* grad_w_im,
                              int D, int k, int n_fft, int row_pitch, float* grad_magnitude, float* grad_phase,
                              void* stream);

/* First half of SpectralMLPBlock.forward, fft_tensor/spectral_layers.py:185 (with :154-158, :162):
 *   y = x + SpectralMixingLayer(LayerNorm(x; ln_w, ln_b, eps))            (dropout inactive)
 * ln_w / ln_b (D) may be NULL (elementwise_affine=False).  ln_stats (B,N,2) receives (mean, rstd) per
 * row and xk_save (B,k,D) complex the spectrum of the normalised input; backward needs both plus x.
 * On the decimated single-launch plan the normalisation is applied while x is loaded and x is added
 * while y is stored (HBM traffic 16 B/sample instead of 28); other plans run the same arithmetic
 * unfused.  y must not alias x.  smx_block_supported(D) == 0 -> SMX_ERR_UNSUPPORTED. */
int smx_block_supported(int D);
int smx_block_forward(const float* x, const float* ln_w, const float* ln_b, float eps,
                      const float* w_re, const float* w_im, const float* bias, float* y,
                      float* xk_save, float* ln_stats, void* workspace, size_t workspace_bytes,
                      int B, int N, int D, int F, void* stream);

/* Autograd backward of smx_block_forward for upstream gradient g:
 *   grad_h = smx_backward(g)            (filter gradients as there, from xk = spectrum of LayerNorm(x))
 *   grad_x = g + LayerNorm'(grad_h)     (torch.nn.LayerNorm backward; written over grad_h in place)
 *   g_ln_w[d] = sum_{b,n} grad_h * xhat,  g_ln_b[d] = sum_{b,n} grad_h     (either may be NULL)
 * phases as in smx_backward; the LayerNorm backward (grad_x, g_ln_w, g_ln_b) belongs to
 * SMX_PHASE_INVERSE. */
int smx_block_backward(const float* g, const float* x, const float* ln_stats, const float* ln_w,
                       const float* xk, const float* w_re, const float* w_im, float* grad_x,
                       float* g_ln_w, float* g_ln_b, float* gw_re, float* gw_im, float* gbias,
                       void* workspace, size_t workspace_bytes, int B, int N, int D, int F,
                       int phases, void* stream);

/* Training-mode dropout of SpectralMixingLayer.forward (nn.Dropout on y + bias, spectral_layers.py:68, :118)
 * fused into the same launches: the "_dropout" twins take the drop probability p in [0, 1) (quantised to
 * 1/65536; p = 0 is the plain call) and rng_state, a DEVICE pointer to two 64-bit words (seed, call
 * counter).  The mask is a counter-based function of (rng_state, batch row, element index): the forward
 * call applies it to y (after the bias, before the block's residual) scaled by 1/(1-p), the backward call
 * given the SAME two words applies it to g.  It is this library's generator, not torch's: same
 * distribution, different bits.  smx_rng_next copies `state` to `saved` (hand `saved` to forward and
 * backward) and advances the counter, all on the device, so a captured hipGraph draws a new mask at
 * every replay.  With more than 512 kept bins the same mask is one more pass inside the call instead of part of a tile
 * store / load: forward on y; backward stages the masked g in grad_x (four-step and eight-band plans: grad_x must then
 * be given, and the eight-band plan wants the SPECTRUM and INVERSE phases in one call) or in a (B, N, D) row copy the
 * workspace holds (band-group plan, smx_plan.groups > 1 -- which for that reason cannot transform in place).
 * Zero-padded rows (smx_*_ex with rows < n_fft) refuse p > 0 with SMX_ERR_UNSUPPORTED.
 * filter_pack (may be NULL): a (k, D) complex64 device buffer, k = min(F, N/2), 16-byte aligned.  The
 * forward call fills it with the filter in the layout its kernels read (pack[f, d] = W[d, f]; without
 * it that copy goes to the workspace); the backward call given the same buffer -- and unchanged weights
 * -- skips its own packing launch.  A forward call may skip it too (constant weights, e.g. inference):
 * OR SMX_FILTER_PACK_READY into conj_w and pass the buffer an earlier forward call filled.  (Two cases
 * never pack and leave the buffer untouched: problems below 8 Mi samples -- the launch would cost more
 * than it saves -- and one band (k <= 128), where every workgroup stages its own 32-channel slice of
 * (D, F) through LDS.) */
#define SMX_FILTER_PACK_READY 2
int smx_rng_next(void* state, void* saved, void* stream);
int smx_forward_dropout(const float* x, const float* w_re, const float* w_im, const float* bias,
                        float* y, float* xk_save, void* workspace, size_t workspace_bytes, int B,
                        int N, int D, int F, int conj_w, float dropout_p, const void* rng_state,
                        float* filter_pack, void* stream);
int smx_backward_dropout(const float* g, const float* xk, const float* w_re, const float* w_im,
                         float* grad_x, float* gw_re, float* gw_im, float* gbias, void* workspace,
                         size_t workspace_bytes, int B, int N, int D, int F, int phases,
                         float dropout_p, const void* rng_state, const float* filter_pack,
                         void* stream);
int smx_block_forward_dropout(const float* x, const float* ln_w, const float* ln_b, float eps,
                              const float* w_re, const float* w_im, const float* bias, float* y,
                              float* xk_save, float* ln_stats, void* workspace,
                              size_t workspace_bytes, int B, int N, int D, int F, float dropout_p,
                              const void* rng_state, float* filter_pack, void* stream);
int smx_block_backward_dropout(const float* g, const float* x, const float* ln_stats,
                               const float* ln_w, const float* xk, const float* w_re,
                               const float* w_im, float* grad_x, float* g_ln_w, float* g_ln_b,
                               float* gw_re, float* gw_im, float* gbias, void* workspace,
                               size_t workspace_bytes, int B, int N, int D, int F, int phases,
                               float dropout_p, const void* rng_state, const float* filter_pack,
                               void* stream);

/* The time path of fft_lm's BicameralBlock on the block's own (B, T, C) layout (round 4):
 *   y[b, t, c] = scale[b, c] * (bias[c] + w[c,0] x[b, t-2, c] + w[c,1] x[b, t-1, c] + w[c,2] x[b, t, c] * [t <= T-2])
 * replaces fft_lm/bicameral.py:214-223 (transpose, F.pad(x[:, :, :-1], (1, 0)), depthwise nn.Conv1d(kernel_size = 3,
 * padding = 1, groups = C), transpose back: the tap on x[t] is absent from the last row because the shifted sequence
 * dropped x[T-1]) and :226-227 (the time gate, as scale (B, C); NULL = 1), and their autograd backward.
 * w = conv1d.weight viewed as (C, 3); bias (C) or NULL.  Backward: grad_x (B, T, C) or NULL; grad_w (C, 3), grad_bias
 * (C), grad_scale (B, C): any may be NULL.  Fixed-order two-stage sums (bitwise reproducible); workspace from
 * smx_dwconv3_workspace_bytes, 256-byte aligned.  Pointers of (B, T, C) tensors 4-byte aligned (16-byte aligned and
 * C % 4 == 0 selects the vector kernels). */
int smx_dwconv3_workspace_bytes(int B, int T, int C, size_t* out);
int smx_dwconv3_forward(const float* x, const float* w, const float* bias, const float* scale, float* y, int B, int T,
                        int C, void* stream);
int smx_dwconv3_backward(const float* g, const float* x, const float* w, const float* bias, const float* scale,
                         float* grad_x, float* grad_w, float* grad_bias, float* grad_scale, void* workspace,
                         size_t workspace_bytes, int B, int T, int C, void* stream);

/* SpectralLayerNorm of fft_lm's FrequencyNativeBlock (round 4), replaces fft_lm/frequency_native.py:203-239 and its
 * autograd backward: per (batch row, bin) the magnitudes of the C channels are normalised (mean, biased variance, eps),
 * scaled by gamma[f, c], shifted by beta[f, c], and put back on the phases of z.
 *   z, out, g, grad_z: (B, F, C) complex64 (interleaved), 8-byte aligned; gamma, beta, grad_gamma, grad_beta: (F, C)
 *   float32 (rows of the module's (n_freqs, C) parameters: pass their base pointers, F <= n_freqs).
 * C <= 1024 (smx_spectral_ln_supported).  grad_z, grad_gamma, grad_beta may each be NULL.  Sums over the batch in a
 * fixed order (bitwise reproducible), no workspace. */
int smx_spectral_ln_supported(int C);
/* planar != 0: out (forward) / g (backward) are (2, B, F, C) float32 -- plane 0 the real parts, plane 1 the imaginary
 * parts -- instead of interleaved complex: the layout SpectralFFN's nn.Linear layers take ("the same weights on the real
 * and on the imaginary part", frequency_native.py:167-172), so no (de)interleaving copy stands between them. */
int smx_spectral_ln_forward(const float* z, const float* gamma, const float* beta, float eps, float* out, int planar,
                            int B, int F, int C, void* stream);
int smx_spectral_ln_backward(const float* g, const float* z, const float* gamma, const float* beta, float eps,
                             float* grad_z, float* grad_gamma, float* grad_beta, int planar, int B, int F, int C,
                             void* stream);
/* The planar side of SpectralFFN (frequency_native.py:167-189, :355-356), h / out / g (2, B, F, C) float32 planes:
 *   smx_planar_cmul_*: out = h (f_re + i f_im)[f, c] -- PhaseShift (:62-77) between the two Linear layers -- and its
 *     backward: grad_h = g conj(f), grad_f_re / grad_f_im (F, C) summed over the batch in a fixed order (any may be NULL);
 *   smx_planar_add: y = a + (p0 + i p1), y and a (n,) complex64 (a may be NULL): the residual around the feed-forward with
 *     the planar result folded in;  smx_planar_split: the two planes of a complex tensor (the backward of that fold). */
int smx_planar_cmul_forward(const float* h, const float* f_re, const float* f_im, float* out, int B, int F, int C,
                            void* stream);
int smx_planar_cmul_backward(const float* g, const float* h, const float* f_re, const float* f_im, float* grad_h,
                             float* grad_f_re, float* grad_f_im, int B, int F, int C, void* stream);
int smx_planar_add(const float* a, const float* planar, float* y, long long n, void* stream);
int smx_planar_split(const float* g, float* planar, long long n, void* stream);

/* The gate chain between the two transforms of fft_lm's twin blocks, one launch each way.  Replaces reference
 * fft_lm/frequency_native.py:95 (FrequencyConvFunc.forward: x k gain), :338 (times the frequency gate and the context
 * gate), :351 (times the cutoff mask) and their backward (:108-117 hand-written, the rest autograd); fft_lm/bicameral.py
 * :179-186 is the same product without u and m.
 *   forward:  y[b,f,c] = ((((x[b,f,c] a[f]) u[c]) p[f]) q[b,c]) m[f], multiplied in this (the reference's) order;
 *             x, y (B, F, C) complex64, a (F) complex64, u (C), p (F), q (B, C), m (F) float32; u, p, q, m may be NULL (= 1).
 *   backward: grad_x = g conj(a) u p q m (may be NULL) and three reductions the caller finishes with a few tiny products:
 *             s1 (F) complex64 = sum_{b,c} g conj(x) u q    -> grad_a = p m s1 (:111), grad_p = m Re(conj(a) s1)
 *             rc (B, C) = sum_f p m Re(conj(a) g conj(x))   -> grad_q = u rc; grad_u = sum_b q rc where autograd is meant
 *             rp (B, C) = sum_f p m Re(a g x)               -> the reference's own grad_gain = sum_b q rp (:115, un-conjugated)
 *             (s1, rc, rp may be NULL).  Sums in a fixed order: bitwise reproducible.
 * C must be even; x, y, g, grad_x 16-byte aligned; workspace as smx_spectral_gate_workspace_bytes says, 256-byte aligned. */
int smx_spectral_gate_workspace_bytes(int B, int F, int C, size_t* out);
int smx_spectral_gate_forward(const float* x, const float* a, const float* u, const float* p, const float* q,
                              const float* m, float* y, int B, int F, int C, void* stream);
int smx_spectral_gate_backward(const float* g, const float* x, const float* a, const float* u, const float* p,
                               const float* q, const float* m, float* grad_x, float* s1, float* rc, float* rp,
                               void* workspace, size_t workspace_bytes, int B, int F, int C, void* stream);

/* BicameralBlock's fusion line (reference fft_lm/bicameral.py:237-268: weighted paths + 0.1 x cross-talk + residual):
 *   forward:  out = r + w[0] a + w[1] b + c3 c   (n float32 each; w two floats in device memory; c may be NULL)
 *   backward: grad_a = w[0] g, grad_b = w[1] g, grad_c = c3 g (each may be NULL), grad_w[0] = sum g a, grad_w[1] = sum g b
 *             (fixed-order sums; may be NULL); grad_r is g itself.
 * n % 4 == 0, tensors 16-byte aligned; workspace as smx_mix_workspace_bytes says, 256-byte aligned. */
int smx_mix_workspace_bytes(size_t* out);
int smx_mix_forward(const float* r, const float* a, const float* b, const float* c, const float* w, float c3, float* out,
                    long long n, void* stream);
int smx_mix_backward(const float* g, const float* a, const float* b, const float* w, float c3, float* grad_a, float* grad_b,
                     float* grad_c, float* grad_w, void* workspace, size_t workspace_bytes, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMX_H_ */
