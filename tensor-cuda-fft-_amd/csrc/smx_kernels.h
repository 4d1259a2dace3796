// smx_kernels.h -- launch-side declarations shared by the kernel translation units and smx_api.
#pragma once
#include <hip/hip_runtime.h>
#include "smx_core.h"

namespace smx {

struct DecimArgs {
  const float* in;      // x (forward) or g (backward), (B,N,D) f32
  float* out;           // y or grad_x, (B,N,D) f32; may be null (spectrum only)
  const cf* tw;         // w_N^n, n < N
  const cf* bt;         // w_N^{16 s' r}, [L][32]
  const cf* tq;         // w_N^{q e}, [N/16][16] (N % 256 == 0): the inter-pass twiddles c^q of row-group t and residue
                        // r at row e = t L + r, read by the streaming loops instead of raising c to its powers
  Geom g;
  FilterArgs fa;
  const cf* v16;        // sixteen-row decimation (g.P != 0): V[s'' + 16][t'] = w_P^{s'' t'}, 32 x 16
  const cf* b16;        // ... and beta[tau][s'' + 16] = w_P^{16 s'' tau}, T x 32
  int placement;        // workgroup placement: 0 b-major, 1 + rotated residues, 2 XCD-aware (default)
  int accumulate;       // four-band kernels only: out += instead of out = (band groups after the first)
  int round;            // workgroups per launch of the streaming kernels (0 = all in one launch)
  int st_plain;         // 0, 1, 2 or 4: a thread's first rows of every tile stored write-back, the others streaming
  int bid0;             // first workgroup index of this launch (set by the launchers)
  // split path only
  int nsplit, lc;       // residues are cut into nsplit chunks of lc
  int sum_in_f;         // 1: k_split_f sums the chunk partials itself (no k_split_sum launch)
  cf* ws_z;             // [B*ndt][nsplit][16 NB][256] partial packed spectra
  cf* ws_zs;            // [B*ndt][16 NB][256] partial spectra summed over the chunks
  cf* ws_s;             // [B*ndt][16 NB][256] filtered packed spectra
  const float* out_scale;  // k_fs_b: (B, D) factors applied to the two channels at the store, or null
  ConvArgs ca;          // rank-one filter on the four-step path (launch_fs_conv)
  const cf* conv_src;   // launch_fs_conv: where the columns are read (ws_f itself, or the saved spectra of x)
  int fs_bgroups;       // four-step backward: batch groups whose slab rows are summed inside k_fs_f (0 = per row)
  cf* ws_f;             // four-step path: [B*ndt][L][16][256] per-residue tile spectra (in place: filtered)
  // dropout (training): mask regenerated from (rng[0], rng[1]) = (seed, call counter) in device memory;
  // forward launches apply it to what they store, backward launches to the g they load
  unsigned drop_thr;    // round(p * 65536); 0 = none
  float drop_scale;     // 65536 / (65536 - drop_thr)
  const unsigned long long* rng;
  // parameter gradients folded into the backward launch (launch_fused, mode 1): `n_cons` reduction workgroups are
  // appended to the grid of the (last round of the) transform launch.  They only ever wait for workgroups with
  // a smaller index -- which the dispatcher has already started -- so nothing can deadlock, whatever else is
  // resident.  sync[dt * B + b] is raised by transform workgroup (b, dt) once its slab rows have landed; the last
  // reduction workgroup to finish leaves the area zero again.
  unsigned* sync;       // sync_words(B, D) words, zero on entry: cleared by the library (hipMemsetAsync) unless the
                        // caller vouches for it (SMX_PHASE_SYNC_CLEAN); the fused FORWARD launch clears it as well
  int n_cons;           // reduction workgroups appended (0 = parameter gradients by launch_gradw_slab)
  float* gw_re;         // (D, F)
  float* gw_im;         // (D, F)
  float* gbias;         // (D) or null
  // fused block (launch_fused_block only): in = x, LayerNorm folded into the load, + x at the store
  const cf* ln_stats;   // (B,N) (mean, rstd)
  const float* ln_w;    // (D) or null (= 1)
  const float* ln_b;    // (D) or null (= 0)
};

// one flag per transform workgroup + the count of finished reduction workgroups
static inline size_t sync_words(int B, int D) { return (size_t)B * ((D + DT - 1) / DT) + 1; }
constexpr int GWT_BINS = 8;        // bins per appended reduction workgroup
// reduction workgroups launch_fused appends for (D, F): ceil(D/32) * (ceil(F/GWT_BINS) + (bias ? 1 : 0))
int gradw_tail_blocks(int D, int F, bool bias);
// sixteen-row decimation (N % 16 == 0, N % 256 != 0, k <= 128 nb): one launch per direction, modes as launch_fused
hipError_t launch_fused16(const DecimArgs& a, int nb, int mode, hipStream_t s);
// ... its residue-split form for few (batch row, d-tile) pairs: (A) partial spectra per chunk of tiles, then
// launch_split_f (shared with the 256-point plan), then (B) the inverse per chunk; (B) with nsplit == 1 is also the
// inverse half of a phase-split backward (from the spectrum launch_fused16 with out == NULL parked in ws_s)
hipError_t launch_split16_a(const DecimArgs& a, int nb, bool drop_in, hipStream_t s);
hipError_t launch_split16_b(const DecimArgs& a, int nb, bool drop_out, hipStream_t s);
// fused single-launch path (nsplit == 1)
hipError_t launch_fused(const DecimArgs& a, int nb, int mode, hipStream_t s);
// synthesis from a given one-sided spectrum (fa.xk_in, fa.sp_scale, fa.sp_herm): fused inverse, or the packed
// spectrum parked for launch_split_b when out == NULL
hipError_t launch_synth(const DecimArgs& a, int nb, hipStream_t s);
hipError_t launch_synth8(const DecimArgs& a, hipStream_t s);       // N = 2048, every bin: eight bands, one launch
// full spectrum at N = 2048 (eight bands, L == 8): one launch per direction, no dropout / residue split
hipError_t launch_full8(const DecimArgs& a, int mode, hipStream_t s);
// four-step path (more than 512 bins, L in {5..16, 18..32 even, 64, 128, 256}): tile spectra -> workspace / column filter / inverse
hipError_t launch_fs_a(const DecimArgs& a, hipStream_t s);
hipError_t launch_fs_f(const DecimArgs& a, int mode, hipStream_t s);     // mode 4: columns from fa.xk_in (synthesis)
hipError_t launch_fs_b(const DecimArgs& a, hipStream_t s);
int conv_column_blocks(int L);      // the same for launch_fs_conv (ConvArgs::r_part)
int fs_column_blocks(int L);        // grid.y of the column launch = rows of FilterArgs::gsc_part per workgroup
// two-level columns: L = L1 L2, L2 in {4, 8, 16} threads per column pair, 9 <= L1 <= 16 (L >= 36)
bool fs_two_level(int L, int* L1, int* L2);
hipError_t launch_fs_big_general(const DecimArgs& a, int mode, int l1, int l2, hipStream_t s);   // L1 = 9 ... 15
// rank-one filter (causal convolution of fft_lm) on the four-step path: column launch, dir 0 forward / 1 backward
// (backward also reduces P -> dL/dH (gh_re, gh_im: N/2 + 1 each) and (R1, R2) -> grad_scale (B, D))
hipError_t launch_fs_conv(const DecimArgs& a, int dir, float* gh_re, float* gh_im, float* grad_scale,
                          hipStream_t s);
hipError_t launch_conv_reduce(const DecimArgs& a, float* gh_re, float* gh_im, float* grad_scale, int ny,
                              float rscale, hipStream_t s, int nwg = 0, int nj = 16);
// the same filter in ONE launch per direction (smx_conv1.hip): n_fft = 512, 1024, 2048 (rows above n_fft / 2 folded).
// dir 0: a.ws_f = where the packed spectrum of x is kept for backward (or null); dir 1: a.ca.xs = that spectrum,
// partial sums as launch_fs_conv with one row of (R1, R2) per workgroup
bool conv1_supported(int N, int R);
// nj = channel pairs per workgroup: 16 (512 threads) or 8 (256 threads, two workgroups per CU)
int conv1_workgroups(int B, int D, int nj);
hipError_t launch_conv1(const DecimArgs& a, int nj, int dir, float* gh_re, float* gh_im, float* grad_scale,
                        hipStream_t s);
// the filter's own response H = rfft(zero-pad(kernel), N) * sigmoid(logits) * mask (f <= N/2) and its backward
hipError_t launch_conv_response(const float* kernel, const float* logits, const float* mask, const cf* tw, int N,
                                int K, float* h_re, float* h_im, hipStream_t s);
hipError_t launch_conv_response_bwd(const float* kernel, const float* logits, const float* mask, const cf* tw, int N,
                                    int K, int n_logits, const float* gh_re, const float* gh_im, float* grad_kernel,
                                    float* grad_logits, hipStream_t s);
// W[d, f] = c_f m[d] exp(i p[d]) (PhaseAwareSpectralMixing) and the gradients of m, p from grad_W (row pitch ld)
hipError_t launch_phase_filter(const float* m, const float* ph, int D, int k, int n_fft, float* w_re, float* w_im,
                               hipStream_t s);
hipError_t launch_phase_filter_bwd(const float* m, const float* ph, const float* gw_re, const float* gw_im, int D, int k,
                                   int n_fft, int ld, float* g_m, float* g_p, hipStream_t s);
// forward of y = x + mix(LayerNorm(x)) in one launch (nsplit == 1 only)
hipError_t launch_fused_block(const DecimArgs& a, int nb, hipStream_t s);
// three-launch path: partial forward / combine+filter / inverse
hipError_t launch_split_a(const DecimArgs& a, int nb, bool drop_in, hipStream_t s);
hipError_t launch_split_f(const DecimArgs& a, int nb, int mode, hipStream_t s);
hipError_t launch_split_b(const DecimArgs& a, int nb, bool drop_out, hipStream_t s);

// elementwise dropout for the plans without a fused epilogue (direct path): out = mask * scale * in
hipError_t launch_dropout_rows(const float* in, float* out, int B, long long row_elems, unsigned thr,
                               float scale, const unsigned long long* rng, hipStream_t s);
// saved = state; state[1] += 1      (one tiny launch: the generator advances on the device, so a captured
// hipGraph draws a fresh mask at every replay)
hipError_t launch_rng_next(unsigned long long* state, unsigned long long* saved, hipStream_t s);

// generic (any N, any k) kernels
struct DirectArgs {
  int B, N, D, F, k;    // k = number of bins of this launch
  const cf* tw;
  // bin i of the launch is frequency f0 + i * fstep (default: the first k bins)
  int f0 = 0, fstep = 1;
  // rows of the spectrum buffers: 0 = compact (B, k, D), row = i; otherwise (B, rows, D), row = frequency
  int rows = 0;
  int accumulate = 0;   // synthesis: y += instead of y =
  int R = 0;            // rows present in x / y (0 = N): zero-padded input, cropped output (Geom::R)
  __host__ __device__ int rows_present() const { return R ? R : N; }
};
hipError_t launch_direct_spectrum(const float* x, cf* xk, const DirectArgs& a, hipStream_t s);
hipError_t launch_direct_filter(const cf* xk, const float* w_re, const float* w_im, int conj_w,
                                cf* sk, const DirectArgs& a, hipStream_t s);
hipError_t launch_direct_synth(const cf* sk, const float* bias, float* y, const DirectArgs& a,
                               hipStream_t s);
hipError_t launch_scale_bins(const cf* in, cf* out, int B, int k, int D, int N, float scale, int hermitian,
                             hipStream_t s);
// large problems on the direct plan go through LDS-tiled kernels (k_tiled_spectrum / k_tiled_synth) inside
// the two launchers above; option "tiled_dft" = 0 keeps the literal fp64 kernels (A/B, tests)
void set_tiled_dft(int on);
// band-group edge bins: spectrum of the few bins f0 + i fstep with the rows spread over the grid;
// part = edge_chunks(B,N,D) * B * k * D * 2 doubles of scratch
int edge_chunks(int B, int N, int D);
hipError_t launch_edge_spectrum(const float* x, cf* xk, double* part, const DirectArgs& a,
                                hipStream_t s);
// y += the edge bins' part of the synthesis; sk compact (B, k, D), already scaled by 1/N
hipError_t launch_edge_synth_acc(const cf* sk, float* y, const DirectArgs& a, hipStream_t s);
// band-group edge bins (frequencies f0 + i fstep): products X conj(G) / N into the rows of the
// (B, k_total, D) grad slab, X from the saved spectrum (rows = frequency), G compact (B, k, D)
hipError_t launch_edge_slab(const cf* xk, const cf* ge, cf* slab, int k_total, const DirectArgs& a,
                            hipStream_t s);

// wt (k, D) complex <- (w_re, w_im) (D, F): the filter transposed and interleaved for the unpack phase
hipError_t launch_pack_w(const float* w_re, const float* w_im, cf* wt, int D, int F, int k,
                         hipStream_t s);

// parameter-gradient reductions (deterministic: fixed order over the batch)
hipError_t launch_gradw_slab(const cf* pslab, const float* gb_part, float* gw_re, float* gw_im,
                             float* gbias, int B, int D, int F, int k, hipStream_t s);
hipError_t launch_gradw_spectra(const cf* xk, const cf* gk, float* gw_re, float* gw_im,
                                float* gbias, int B, int N, int D, int F, int k, hipStream_t s);

// complex-in / complex-out filter of wirtinger_ops.WirtingerSpectralFilter and the x*w Function
hipError_t launch_wfilter(const cf* xf, const float* w_re, const float* w_im, int conj_w, cf* out,
                          int B, int N, int D, int F, int k, hipStream_t s);
hipError_t launch_wfilter_gradw(const cf* xf, const cf* gf, float* gw_re, float* gw_im, int B,
                                int N, int D, int F, int k, hipStream_t s);
hipError_t launch_cmul(const cf* x, const cf* w, int conj_w, cf* out, long long batch,
                       long long inner, hipStream_t s);
hipError_t launch_cmul_gradw(const cf* x, const cf* g, cf* gw, long long batch, long long inner,
                             hipStream_t s);

// ---- depthwise causal 3-tap convolution on (B, T, C): BicameralBlock's time path (smx_time.hip) ----------------------
size_t dwconv3_workspace_bytes(int B, int T, int C);        // partial sums of the backward: [B ceil(T/32)][5][C] floats
hipError_t launch_dwconv3_fwd(const float* x, const float* w, const float* bias, const float* scale, float* y, int B,
                              int T, int C, hipStream_t s);
hipError_t launch_dwconv3_bwd(const float* g, const float* x, const float* w, const float* bias, const float* scale,
                              float* gx, float* gw, float* gbias, float* gscale, float* part, int B, int T, int C,
                              hipStream_t s);

// ---- SpectralLayerNorm on a (B, F, C) complex spectrum, gamma / beta rows per bin (smx_time.hip) ----------------------
bool spectral_ln_supported(int C);         // C <= 1024: the row lives in one wavefront's registers
// planar: out / g are (2, B, F, C) float32 planes (real, imaginary) instead of interleaved complex
hipError_t launch_spectral_ln_fwd(const cf* z, const float* gamma, const float* beta, float eps, cf* out, int planar,
                                  int B, int F, int C, hipStream_t s);
hipError_t launch_spectral_ln_bwd(const cf* g, const cf* z, const float* gamma, const float* beta, float eps, cf* gz,
                                  float* ggamma, float* gbeta, int planar, int B, int F, int C, hipStream_t s);
// the planar side of SpectralFFN: complex factor per (bin, channel) on (2, B, F, C) planes; planar <-> interleaved
hipError_t launch_pcmul_fwd(const float* h, const float* fr, const float* fi, float* out, int B, int F, int C,
                            hipStream_t s);
hipError_t launch_pcmul_bwd(const float* g, const float* h, const float* fr, const float* fi, float* gh, float* gfr,
                            float* gfi, int B, int F, int C, hipStream_t s);
hipError_t launch_add_planar(const cf* a, const float* p, cf* y, long long n, hipStream_t s);   // y = a + (p0 + i p1)
hipError_t launch_to_planar(const cf* g, float* p, long long n, hipStream_t s);
// BicameralBlock's fusion line: out = r + w[0] a + w[1] b + c3 c  (smx_time.hip)
size_t mix_workspace_bytes();
hipError_t launch_mix_fwd(const float* r, const float* a, const float* b, const float* c, const float* w, float c3,
                          float* out, long long n, hipStream_t s);
hipError_t launch_mix_bwd(const float* g, const float* a, const float* b, const float* w, float c3, float* ga, float* gb,
                          float* gc, float* gw, float* part, long long n, hipStream_t s);
// the gate chain of the twin blocks: y = ((((x a[f]) u[c]) p[f]) q[b,c]) m[f]  (smx_time.hip)
size_t gate_workspace_bytes(int B, int F, int C);
hipError_t launch_gate_fwd(const cf* x, const cf* a, const float* u, const float* p, const float* q, const float* m, cf* y,
                           int B, int F, int C, hipStream_t s);
hipError_t launch_gate_bwd(const cf* g, const cf* x, const cf* a, const float* u, const float* p, const float* q,
                           const float* m, cf* gx, cf* s1, float* rc, float* rp, cf* part, int B, int F, int C,
                           hipStream_t s);

// ---- LayerNorm row kernels of the fused block (smx_block.hip) -------------------------------------
constexpr int LN_MAX_BLOCKS = 2048;      // most rows of the grad_gamma / grad_beta partial buffer
int ln_num_blocks(long long rows);       // blocks (= partial rows) launch_ln_bwd uses for `rows` rows
constexpr int LN_MAX_D = 4096;           // widest row held in registers (D % 4 == 0)
constexpr int LN_MAX_D_ODD = 1024;       // same for the scalar variant (D % 4 != 0)
bool ln_supported(int D);
hipError_t launch_ln_stats(const float* x, cf* stats, long long rows, int D, float eps,
                           hipStream_t s);
hipError_t launch_ln_apply(const float* x, const cf* stats, const float* gamma, const float* beta,
                           float* h, long long rows, int D, hipStream_t s);
hipError_t launch_add_rows(float* y, const float* x, size_t total, hipStream_t s);
// in place over grad_h; part: (ln_num_blocks(rows), 2, D) floats of scratch
hipError_t launch_ln_bwd(float* gh_dx, const float* x, const float* g, const cf* stats,
                         const float* gamma, float* part, float* g_gamma, float* g_beta,
                         long long rows, int D, hipStream_t s);

}  // namespace smx
