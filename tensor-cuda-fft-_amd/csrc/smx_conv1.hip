// smx_conv1.hip -- rank-one filter (fft_lm's causal FFT convolution) in ONE launch per direction for
// n_fft = 512, 1024, 2048 (rows <= n_fft / 2 as they are, more rows folded onto the lower half first).
// Arithmetic and the description: end of smx_core.h.
//
// Replaces for those lengths: reference fft_lm/train_fixed_full.py:515-519 (rfft of the zero-padded sequence),
// :521-551 (response, gates), :553-555 (irfft, crop) and their autograd backward; supersedes the three launches
// k_fs_a / k_fs_conv / k_fs_b of smx_fourstep.hip there (option "conv1" = 0 keeps those).
#include "smx_launch.h"

namespace smx {

namespace {

template <int CTRL>
__device__ __forceinline__ float c1_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// One value per lane out of sixteen: lane j of a DPP row (= channel pair j of a row group) returns the sum over the
// row's 16 lanes of a[j].  A butterfly that halves the number of live values at every step -- partner lane ^ 15
// (row_mirror), ^ 7 (row_half_mirror), ^ 3, ^ 1 (quad_perm), lane bit 3, 2, 1, 0 choosing which half it keeps --
// 15 DPP adds and 30 selects for 16 sums instead of 64 DPP adds, and no lane-15-only store afterwards.
template <int CTRL, int CNT>
__device__ __forceinline__ void c1_fold(const float* in, float* out, bool hi) {
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    const float keep = hi ? in[k + CNT] : in[k], send = hi ? in[k] : in[k + CNT];
    out[k] = keep + c1_dpp<CTRL>(send);
  }
}
__device__ __forceinline__ float c1_row_transpose_sum(const float (&a)[16], int j) {
  float b[8], c[4], d[2], e[1];
  c1_fold<0x140, 8>(a, b, (j & 8) != 0);
  c1_fold<0x141, 4>(b, c, (j & 4) != 0);
  c1_fold<0x1B, 2>(c, d, (j & 2) != 0);
  c1_fold<0xB1, 1>(d, e, (j & 1) != 0);
  return e[0];
}

// the same over the EIGHT lanes of a channel-pair group (NJ = 8: a DPP row holds two row groups, which must not mix):
// partner lane ^ 7, ^ 3, ^ 1; lane j returns the sum of a[j & 7] over its group
__device__ __forceinline__ float c1_half_row_transpose_sum(const float* a, int j) {
  float c[4], d[2], e[1];
  c1_fold<0x141, 4>(a, c, (j & 4) != 0);
  c1_fold<0x1B, 2>(c, d, (j & 2) != 0);
  c1_fold<0xB1, 1>(d, e, (j & 1) != 0);
  return e[0];
}

// LDS map (complex elements, EXJ = 16 x 16 x NJ = one team's exchange buffer): E[p][buf] = lds + (2 p + buf) EXJ during
// the forward loop; Hs = lds + 4 EXJ; inverse loop: E[p] = lds + 2 p EXJ, C = lds + EXJ; backward sums between the
// loops: Pbuf = lds + EXJ (N), Rbuf = lds + 3 EXJ (one entry per thread).
template <int NJ> constexpr int c1_exj() { return 16 * 16 * NJ; }
template <int LP, int NJ> constexpr int c1_lds_elems() { return 4 * c1_exj<NJ>() + 512 * LP; }

// FOLD (rows > N / 2): the tile is x[n] + x[n + N/2] for the even team, x[n] - x[n + N/2] for the odd one; the upper
// rows (nh, padded: hh.R = rows - N/2 of them exist) are prefetched beside the lower ones (nx, all present)
template <int LP, int R, bool PAD, int NJ, bool FOLD>
__device__ __forceinline__ void c1_fwd_tiles(cf (&acc)[16 * LP], cf (&nx)[16], cf (&nh)[FOLD ? 16 : 1], cf* lds,
                                             const float* __restrict__ xb, const Geom& h, const Geom& hh,
                                             const cf (&twe)[LP], const cf (&tw2e)[LP], int N, int p, int t, int j) {
  if constexpr (R < LP) {
    constexpr bool PLO = PAD && !FOLD;           // the lower rows are all there once rows > N / 2
    const float* xh = xb + (size_t)h.N * h.D;
    cf v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = nx[u];
    if constexpr (FOLD) c1_fold_in(v, nh, p);
    if constexpr (R + 1 < LP) {
      load_part_tile<0, 8, PLO, false>(xb, h, t, R + 1, nx);
      if constexpr (FOLD) load_part_tile<0, 8, PAD, false>(xh, hh, t, R + 1, nh);
    }
    cf* E = lds + (2 * p + (R & 1)) * c1_exj<NJ>();
    c1_fwd_phase1<LP, NJ>(v, twe[R], tw2e[R], E, p, t, j);
    __syncthreads();
    if constexpr (R + 1 < LP) {
      load_part_tile<8, 8, PLO, false>(xb, h, t, R + 1, nx);
      if constexpr (FOLD) load_part_tile<8, 8, PAD, false>(xh, hh, t, R + 1, nh);
    }
    c1_fwd_phase2<LP, R, NJ>(acc, E, t, j);
    c1_fwd_tiles<LP, R + 1, PAD, NJ, FOLD>(acc, nx, nh, lds, xb, h, hh, twe, tw2e, N, p, t, j);
  }
}
template <int LP, int R, bool PAD, int NJ, bool FOLD>
__device__ __forceinline__ void c1_inv_tiles(const cf (&acc)[16 * LP], cf* lds, float* __restrict__ yb, const Geom& h,
                                             const cf (&twe)[LP], const cf (&tw2e)[LP], int N, int p, int t, int j,
                                             int lt, bool valid, float sa, float sb) {
  if constexpr (R < LP) {
    cf v[16];
    cf* E = lds + 2 * p * c1_exj<NJ>();
    cf* C = lds + c1_exj<NJ>();
    c1_inv_phase1<LP, R, NJ>(acc, v, E, t, j);
    __syncthreads();
    c1_inv_phase2<LP, NJ>(v, twe[R], tw2e[R], E, p, t, j);
    c1_comb_write<NJ>(v, C, p, lt);
    __syncthreads();
    c1_comb_store<PAD, NJ, FOLD>(v, C, yb, h, p, t, lt, R, valid, sa, sb);
    c1_inv_tiles<LP, R + 1, PAD, NJ, FOLD>(acc, lds, yb, h, twe, tw2e, N, p, t, j, lt, valid, sa, sb);
  }
}

// DIR 0: y = s * conv(x);  a.ws_f = where the packed spectrum of x is kept (or null)
// DIR 1: grad_x = s * conv^T(g), P partials -> a.ca.p_part[wg][N], (R1, R2) -> a.ca.r_part[wg][NJ]
// NJ channel pairs per workgroup: 16 (512 threads, one workgroup per CU) or 8 (256 threads on 16 channels, 64-byte row
// segments, two workgroups per CU); wg = b ceil(D / (2 NJ)) + d-tile either way.
// FOLD: N / 2 < rows <= N (PAD: rows < N); otherwise rows <= N / 2 (PAD: rows < N / 2).
template <int LP, int DIR, bool PAD, int NJ, bool FOLD>
__global__ __launch_bounds__(c1_tpb<NJ>(), NJ == 16 ? 1 : 2) void k_conv1(const DecimArgs a) {
  __shared__ cf lds[c1_lds_elems<LP, NJ>()];
  constexpr int EXJ = c1_exj<NJ>(), TS = 16 * NJ, DTJ = 2 * NJ;
  const Geom& g = a.g;                         // the n_fft geometry (N = 512 LP, R rows)
  Geom h = g;                                  // tile geometry of the two half-length transforms
  h.N = g.N / 2; h.L = LP;
  Geom hh = h;                                 // ... of the upper rows n + N / 2 (FOLD): rows - N / 2 of them exist
  hh.R = g.R - h.N;
  const int N = g.N;
  const int tid = threadIdx.x, p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ;
  const int ndt = (g.D + DTJ - 1) / DTJ;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, LP, a.placement);
  const int b = w.b, wg = b * ndt + w.dt, d = w.dt * DTJ + 2 * j;
  const bool valid = d < g.D;
  const int dc = valid ? d : g.D - 2;
  const float* xb = a.in + (size_t)b * g.R * g.D + dc;
  cf* Hs = lds + 4 * EXJ;

  cf acc[16 * LP];
  cf nx[16], nh[FOLD ? 16 : 1];
  // this thread's inter-pass twiddles of its LP tiles, w_N^e and w_N^{2e} with e = LP t + R: requested FIRST (ahead of
  // every tile load in the in-order vector-memory queue), used by the forward tiles and again by the inverse ones
  cf twe[LP], tw2e[LP];
#pragma unroll
  for (int R = 0; R < LP; ++R) { twe[R] = a.tw[LP * t + R]; tw2e[R] = a.tw[2 * (LP * t + R)]; }
  load_tile<PAD && !FOLD, false>(xb, h, t, 0, nx);      // (cached: both teams read the same rows)
  if constexpr (FOLD) load_tile<PAD, false>(xb + (size_t)h.N * h.D, hh, t, 0, nh);
  float sa = 1.f, sb = 1.f;
  if (a.ca.sc) { sa = a.ca.sc[(size_t)b * g.D + dc]; sb = a.ca.sc[(size_t)b * g.D + dc + 1]; }
  c1_stage_h<NJ>(a.ca, N, g.inv_n, Hs, tid);
  c1_fwd_tiles<LP, 0, PAD, NJ, FOLD>(acc, nx, nh, lds, xb, h, hh, twe, tw2e, N, p, t, j);
  c1_pin(acc);
  c1_residues<LP, -1>(acc);
  c1_pin(acc);

  if constexpr (DIR == 0) {
    cf* xsave = a.ws_f ? a.ws_f + (size_t)wg * (16 * LP) * c1_tpb<NJ>() : nullptr;
    c1_mid_fwd<LP, NJ>(acc, Hs, xsave, p, t, tid);
    __syncthreads();                           // every thread is done with the forward exchange buffers
  } else {
    __syncthreads();                           // Pbuf / Rbuf reuse the forward exchange buffers
    cf* Pbuf = lds + EXJ;
    cf* Rbuf = lds + 3 * EXJ;
    const float sig = valid ? 0.5f * (sa + sb) : 0.f, del = valid ? 0.5f * (sa - sb) : 0.f;
    cf rr;
    c1_mid_bwd<LP, NJ>(acc, Hs, a.ca.xs + (size_t)wg * (16 * LP) * c1_tpb<NJ>(), sig, del, p, t, j, tid, rr,
                       [&](int grp, const float (&px)[16], const float (&py)[16]) {
                         if constexpr (NJ == 16) {
                           Pbuf[c1_bin(p, t, c1_group_slot<LP>(grp, j))] =
                               mk(c1_row_transpose_sum(px, j), c1_row_transpose_sum(py, j));
                         } else {                // eight lanes per row group: the low and the high eight slots apart
                           Pbuf[c1_bin(p, t, c1_group_slot<LP>(grp, j))] =
                               mk(c1_half_row_transpose_sum(px, j), c1_half_row_transpose_sum(py, j));
                           Pbuf[c1_bin(p, t, c1_group_slot<LP>(grp, 8 + j))] =
                               mk(c1_half_row_transpose_sum(px + 8, j), c1_half_row_transpose_sum(py + 8, j));
                         }
                       });
    if (!valid) rr = mk(0.f, 0.f);
    Rbuf[tid] = rr;
    __syncthreads();
    cf* pp = a.ca.p_part + (size_t)wg * N;
    for (int f = tid; f < N; f += c1_tpb<NJ>()) pp[f] = Pbuf[f];
    if (tid < NJ) {                            // (R1, R2) of channel pair j = tid: the 32 (half, row group) threads
      cf s = mk(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 32; ++i) s = cadd(s, Rbuf[i * NJ + tid]);
      a.ca.r_part[(size_t)wg * NJ + tid] = s;
    }
  }
  if (a.out == nullptr) return;
  c1_pin(acc);
  c1_residues<LP, +1>(acc);
  c1_pin(acc);
  c1_inv_tiles<LP, 0, PAD, NJ, FOLD>(acc, lds, a.out + (size_t)b * g.R * g.D + d, h, twe, tw2e, N, p, t, j, lt, valid, sa,
                                     sb);
}

// ---- the filter's own response (reference fft_lm/train_fixed_full.py:511-513, :529, :540-551) -------------------
//   H[f] = rfft(zero-pad(kernel, n_fft))[f] * sigmoid(gate_logits[f]) * mask[f],   f <= n_fft / 2
// K taps against the exact twiddle table (index f t mod N): one thread per bin.  Replaces two matrix-vector
// products with a cached DFT matrix, a sigmoid, a slice and three multiplications (seven launches of about 5 us).
// kf[f] = sum_t kernel[t] w_N^{f t} for the 32 bins f0 .. f0 + 31 of a 256-thread block: thread (bin = tid & 31,
// tap group = tid >> 5) sums every eighth tap (a chain of K / 8 table loads instead of K), the eight partial sums
// meet in LDS in a fixed order.  Every thread of the block must call it; thread tid < 32 gets the result.
__device__ __forceinline__ cf c1_kf32(const float* __restrict__ kernel, const cf* __restrict__ tw, int N, int K, int f0,
                                      cf* red) {
  const int tid = threadIdx.x, fl = tid & 31, tg = tid >> 5, f = f0 + fl;
  float re = 0.f, im = 0.f;
  const unsigned fm = (unsigned)(f % N);
  unsigned idx = (unsigned)(((unsigned long long)fm * (unsigned)tg) % (unsigned)N);
  const unsigned step = (unsigned)(((unsigned long long)fm * 8u) % (unsigned)N);
  for (int t = tg; t < K; t += 8) {
    const cf w = tw[idx];
    const float k = kernel[t];
    re = fmaf(k, w.x, re);
    im = fmaf(k, w.y, im);
    idx += step;
    if (idx >= (unsigned)N) idx -= (unsigned)N;
  }
  red[tid] = mk(re, im);
  __syncthreads();
  cf acc = mk(0.f, 0.f);
  if (tid < 32) {
#pragma unroll
    for (int g = 0; g < 8; ++g) acc = cadd(acc, red[g * 32 + tid]);
  }
  __syncthreads();
  return acc;
}
__device__ __forceinline__ float c1_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void k_conv_response(const float* __restrict__ kernel,
                                                      const float* __restrict__ logits,
                                                      const float* __restrict__ mask, const cf* __restrict__ tw, int N,
                                                      int K, float* __restrict__ h_re, float* __restrict__ h_im) {
  __shared__ cf red[256];
  const int f0 = blockIdx.x * 32, f = f0 + (int)threadIdx.x;
  const cf kf = c1_kf32(kernel, tw, N, K, f0, red);
  if (threadIdx.x >= 32 || f > N / 2) return;
  float sg = logits ? c1_sigmoid(logits[f]) : 1.f;
  if (mask) sg *= mask[f];
  h_re[f] = kf.x * sg;
  h_im[f] = kf.y * sg;
}
// backward: block t < K -> grad_kernel[t] = sum_f Re(dkf[f] conj(w_N^{f t})), dkf = (gh_re, gh_im) sigmoid mask;
// blocks K ... -> grad_logits[f] = (gh_re kf_re + gh_im kf_im) mask sigmoid (1 - sigmoid) for 32 bins each, zero from
// n_fft / 2 + 1 on.  Fixed summation order (per thread ascending f, then a fixed tree over the block).
__global__ __launch_bounds__(256) void k_conv_response_bwd(const float* __restrict__ kernel,
                                                          const float* __restrict__ logits,
                                                          const float* __restrict__ mask, const cf* __restrict__ tw,
                                                          int N, int K, int n_logits,
                                                          const float* __restrict__ gh_re,
                                                          const float* __restrict__ gh_im,
                                                          float* __restrict__ grad_kernel,
                                                          float* __restrict__ grad_logits) {
  __shared__ cf red[256];
  const int fb = N / 2 + 1, tid = threadIdx.x;
  if ((int)blockIdx.x >= K) {                         // the gate logits, 32 per block
    const int f0 = ((int)blockIdx.x - K) * 32, f = f0 + tid;
    if (f0 >= fb) {                                   // past the spectrum: zeros (uniform per block)
      if (tid < 32 && f < n_logits) grad_logits[f] = 0.f;
      return;
    }
    const cf kf = c1_kf32(kernel, tw, N, K, f0, red);
    if (tid >= 32 || f >= n_logits) return;
    float gl = 0.f;
    if (f < fb) {
      const float sg = c1_sigmoid(logits[f]);
      gl = (gh_re[f] * kf.x + gh_im[f] * kf.y) * sg * (1.f - sg);
      if (mask) gl *= mask[f];
    }
    grad_logits[f] = gl;
    return;
  }
  if (grad_kernel == nullptr) return;
  const int t = blockIdx.x;
  float acc = 0.f;
  for (int f = tid; f < fb; f += 256) {
    float sg = logits ? c1_sigmoid(logits[f]) : 1.f;
    if (mask) sg *= mask[f];
    const cf w = tw[(unsigned)(((unsigned long long)f * (unsigned)t) % (unsigned)N)];
    acc += sg * (gh_re[f] * w.x + gh_im[f] * w.y);    // Re((a + i b)(cos + i sin)) with w = cos - i sin
  }
  float* redf = reinterpret_cast<float*>(red);
  redf[tid] = acc;
  __syncthreads();
#pragma unroll
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) redf[tid] += redf[tid + st];
    __syncthreads();
  }
  if (tid == 0) grad_kernel[t] = redf[0];
}

// ---- the filter of PhaseAwareSpectralMixing (reference fft_tensor/spectral_enhancements.py:147-164) ------------------
//   W[d, f] = c_f m[d] exp(i p[d]),  f < k;  c_f = 2, 1 at DC and at the Nyquist bin of an even n_fft (irfft's weights)
// -- "magnitude times m, phase plus p" on every bin is one complex constant per channel.  One launch instead of six
// elementwise torch launches on (D, k) tensors; one more for the gradients of m and p instead of about ten.
__global__ void k_phase_filter(const float* __restrict__ m, const float* __restrict__ ph, int D, int k, int n_fft,
                               float* __restrict__ w_re, float* __restrict__ w_im) {
  const long long total = (long long)D * k;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i / k), f = (int)(i % k);
    const float c = (f == 0 || (n_fft % 2 == 0 && 2 * f == n_fft)) ? 1.f : 2.f;
    float sn, cs;
    sincosf(ph[d], &sn, &cs);
    const float a = c * m[d];
    w_re[i] = a * cs;
    w_im[i] = a * sn;
  }
}
// g_m[d] = sum_f c_f (gw_re cos p + gw_im sin p),  g_p[d] = m[d] sum_f c_f (gw_im cos p - gw_re sin p);  gw (D, ld)
__global__ __launch_bounds__(256) void k_phase_filter_bwd(const float* __restrict__ m, const float* __restrict__ ph,
                                                         const float* __restrict__ gw_re,
                                                         const float* __restrict__ gw_im, int D, int k, int n_fft,
                                                         int ld, float* __restrict__ g_m, float* __restrict__ g_p) {
  __shared__ float ra[256], rb[256];
  const int d = blockIdx.x, tid = threadIdx.x;
  float sr = 0.f, si = 0.f;                      // sum_f c_f gw_re, sum_f c_f gw_im (fixed order per thread, then a tree)
  for (int f = tid; f < k; f += 256) {
    const float c = (f == 0 || (n_fft % 2 == 0 && 2 * f == n_fft)) ? 1.f : 2.f;
    sr = fmaf(c, gw_re[(size_t)d * ld + f], sr);
    si = fmaf(c, gw_im[(size_t)d * ld + f], si);
  }
  ra[tid] = sr; rb[tid] = si;
  __syncthreads();
#pragma unroll
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) { ra[tid] += ra[tid + st]; rb[tid] += rb[tid + st]; }
    __syncthreads();
  }
  if (tid == 0) {
    float sn, cs;
    sincosf(ph[d], &sn, &cs);
    if (g_m) g_m[d] = ra[0] * cs + rb[0] * sn;
    if (g_p) g_p[d] = m[d] * (rb[0] * cs - ra[0] * sn);
  }
}

}  // namespace

hipError_t launch_conv_response(const float* kernel, const float* logits, const float* mask, const cf* tw, int N,
                                int K, float* h_re, float* h_im, hipStream_t s) {
  hipLaunchKernelGGL(k_conv_response, dim3((N / 2 + 1 + 31) / 32), dim3(256), 0, s, kernel, logits, mask, tw, N, K,
                     h_re, h_im);
  return hipGetLastError();
}
hipError_t launch_conv_response_bwd(const float* kernel, const float* logits, const float* mask, const cf* tw, int N,
                                    int K, int n_logits, const float* gh_re, const float* gh_im, float* grad_kernel,
                                    float* grad_logits, hipStream_t s) {
  const int lb = grad_logits ? (n_logits + 31) / 32 : 0;
  hipLaunchKernelGGL(k_conv_response_bwd, dim3(K + lb), dim3(256), 0, s, kernel, logits, mask, tw, N, K, n_logits,
                     gh_re, gh_im, grad_kernel, grad_logits);
  return hipGetLastError();
}

hipError_t launch_phase_filter(const float* m, const float* ph, int D, int k, int n_fft, float* w_re, float* w_im,
                               hipStream_t s) {
  const long long total = (long long)D * k;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_phase_filter, dim3(blocks ? blocks : 1), dim3(256), 0, s, m, ph, D, k, n_fft, w_re, w_im);
  return hipGetLastError();
}
hipError_t launch_phase_filter_bwd(const float* m, const float* ph, const float* gw_re, const float* gw_im, int D, int k,
                                   int n_fft, int ld, float* g_m, float* g_p, hipStream_t s) {
  hipLaunchKernelGGL(k_phase_filter_bwd, dim3(D), dim3(256), 0, s, m, ph, gw_re, gw_im, D, k, n_fft, ld, g_m, g_p);
  return hipGetLastError();
}

namespace {
template <int LP, int NJ, bool FOLD>
void launch_conv1_f(const DecimArgs& a, int dir, bool pad, dim3 grid, dim3 block, hipStream_t s) {
  if (dir == 0 && pad) hipLaunchKernelGGL((k_conv1<LP, 0, true, NJ, FOLD>), grid, block, 0, s, a);
  else if (dir == 0) hipLaunchKernelGGL((k_conv1<LP, 0, false, NJ, FOLD>), grid, block, 0, s, a);
  else if (pad) hipLaunchKernelGGL((k_conv1<LP, 1, true, NJ, FOLD>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_conv1<LP, 1, false, NJ, FOLD>), grid, block, 0, s, a);
}
template <int LP, int NJ>
void launch_conv1_t(const DecimArgs& a, int dir, hipStream_t s) {
  const dim3 grid(conv1_workgroups(a.g.B, a.g.D, NJ)), block(c1_tpb<NJ>());
  const bool fold = 2 * a.g.R > a.g.N;                    // rows beyond N / 2: folded onto the lower half
  const bool pad = fold ? a.g.R < a.g.N : 2 * a.g.R < a.g.N;
  if (fold) launch_conv1_f<LP, NJ, true>(a, dir, pad, grid, block, s);
  else launch_conv1_f<LP, NJ, false>(a, dir, pad, grid, block, s);
}
template <int NJ>
hipError_t launch_conv1_nj(const DecimArgs& a, int dir, hipStream_t s) {
  switch (a.g.N) {
    case 512: launch_conv1_t<1, NJ>(a, dir, s); break;
    case 1024: launch_conv1_t<2, NJ>(a, dir, s); break;
    case 2048: launch_conv1_t<4, NJ>(a, dir, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

bool conv1_supported(int N, int R) { return (N == 512 || N == 1024 || N == 2048) && R >= 1 && R <= N; }

int conv1_workgroups(int B, int D, int nj) { return B * ((D + 2 * nj - 1) / (2 * nj)); }

hipError_t launch_conv1(const DecimArgs& a0, int nj, int dir, float* gh_re, float* gh_im, float* grad_scale,
                        hipStream_t s) {
  DecimArgs a = a0;
  a.bid0 = 0;
  const hipError_t e = nj == 8 ? launch_conv1_nj<8>(a, dir, s) : launch_conv1_nj<16>(a, dir, s);
  if (e != hipSuccess) return e;
  if (dir == 1)                                   // (R1, R2) arrive / N, one row of nj per workgroup
    return launch_conv_reduce(a, gh_re, gh_im, grad_scale, 1, 0.5f, s, conv1_workgroups(a.g.B, a.g.D, nj), nj);
  return hipGetLastError();
}

}  // namespace smx
