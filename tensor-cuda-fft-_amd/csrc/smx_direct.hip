// smx_direct.hip -- shape-generic kernels (any N incl. odd / non power of two, any k <= N/2),
// the parameter-gradient reductions, and the complex-in/complex-out Wirtinger filter kernels.
//
// The direct path evaluates the pruned DFT sums literally, O(N k) per column, with fp64
// accumulation and an exact (f*n mod N) index into the fp64-generated twiddle table.  It is the
// path for shapes the decimated kernels do not take (N % 256 != 0, odd D); its building blocks also
// serve the edge bins of the band-group plans (k > 512).
//
// Reference lines: fft_tensor/spectral_layers.py:88-116; fft_tensor/wirtinger_ops.py:45-50,67-82,170-203.
#include "smx_kernels.h"

namespace smx {

constexpr int DB = 64;   // channels per block (contiguous, coalesced)
constexpr int RB = 4;    // bins / rows per block

// A real sequence has a real DC and (even N) Nyquist bin: the decimated kernels get an exact +0 there by
// construction (Z - Z), a DFT product gets rounding dust of either sign -- which decides arg() of a bin that a
// later mask zeroes (reference fft_lm/frequency_native.py:236 takes the angle of such bins).  Store exact zeros.
__device__ __forceinline__ cf real_bin(cf v, int f, int N) {
  if (f == 0 || 2 * f == N) v.y = 0.f;
  return v;
}

// Xk[b,f,d] = sum_n x[b,n,d] w_N^{f n}
__global__ __launch_bounds__(DB * RB) void k_direct_spectrum(const float* __restrict__ x,
                                                            cf* __restrict__ xk, DirectArgs a) {
  const int d = blockIdx.x * DB + (threadIdx.x % DB);
  const int fi = blockIdx.y * RB + (threadIdx.x / DB);
  const int b = blockIdx.z;
  if (d >= a.D || fi >= a.k) return;
  const int f = a.f0 + fi * a.fstep;
  const int R = a.rows_present();             // rows n >= R are zero padding
  const float* xp = x + (size_t)b * R * a.D + d;
  double re = 0.0, im = 0.0;
  int idx = 0, n = 0;
  for (; n + 8 <= R; n += 8) {                // eight rows in flight per thread
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = xp[(size_t)(n + u) * a.D];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const cf w = a.tw[idx];
      re += (double)(v[u] * w.x);
      im += (double)(v[u] * w.y);
      idx += f;
      if (idx >= a.N) idx -= a.N;
    }
  }
  for (; n < R; ++n) {
    const float v = xp[(size_t)n * a.D];
    const cf w = a.tw[idx];
    re += (double)(v * w.x);
    im += (double)(v * w.y);
    idx += f;
    if (idx >= a.N) idx -= a.N;
  }
  const size_t row = a.rows ? (size_t)b * a.rows + f : (size_t)b * a.k + fi;
  xk[row * a.D + d] = real_bin(mk((float)re, (float)im), f, a.N);
}

// Sk[b,f,d] = W[d,f] (or conj) * Xk[b,f,d] / N
__global__ void k_direct_filter(const cf* __restrict__ xk, const float* __restrict__ w_re,
                                const float* __restrict__ w_im, int conj_w, cf* __restrict__ sk,
                                DirectArgs a) {
  const size_t total = (size_t)a.B * a.k * a.D;
  const float inv_n = (float)(1.0 / (double)a.N);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % a.D);
    const int fi = (int)((i / a.D) % a.k);
    const int f = a.f0 + fi * a.fstep;
    cf w = mk(w_re[(size_t)d * a.F + f], w_im[(size_t)d * a.F + f]);
    if (conj_w) w = cconj(w);
    // the input may live in the rows of a (B, rows, D) buffer; the output is always compact
    const size_t in = a.rows ? (((size_t)(i / a.D / a.k)) * a.rows + f) * a.D + d : i;
    sk[i] = cscale(cmul(w, xk[in]), inv_n);
  }
}

// y[b,n,d] = bias[d] + sum_{f<k} Re( Sk[b,f,d] * conj(w_N^{f n}) )
__global__ __launch_bounds__(DB * RB) void k_direct_synth(const cf* __restrict__ sk,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ y, DirectArgs a) {
  const int d = blockIdx.x * DB + (threadIdx.x % DB);
  const int n = blockIdx.y * RB + (threadIdx.x / DB);
  const int b = blockIdx.z;
  const int R = a.rows_present();
  if (d >= a.D || n >= R) return;
  const cf* sp = sk + (size_t)b * a.k * a.D + d;
  double acc = 0.0;
  const int step = (int)(((long long)a.fstep * n) % a.N);
  int idx = (int)(((long long)a.f0 * n) % a.N);
  for (int fi = 0; fi < a.k; ++fi) {
    const cf s = sp[(size_t)fi * a.D];
    const cf w = a.tw[idx];
    acc += (double)(s.x * w.x + s.y * w.y);      // Re(s * conj(w))
    idx += step;
    if (idx >= a.N) idx -= a.N;
  }
  float* yp = y + ((size_t)b * R + n) * a.D + d;
  const float v = (float)acc + (bias ? bias[d] : 0.f);
  *yp = a.accumulate ? *yp + v : v;
}

bool use_tiled(const DirectArgs& a);
hipError_t launch_tiled_spectrum(const float* x, cf* xk, const DirectArgs& a, hipStream_t s);
hipError_t launch_tiled_synth(const cf* sk, const float* bias, float* y, const DirectArgs& a, hipStream_t s);

hipError_t launch_direct_spectrum(const float* x, cf* xk, const DirectArgs& a, hipStream_t s) {
  if (a.k == 0 || a.B == 0) return hipSuccess;
  if (use_tiled(a)) return launch_tiled_spectrum(x, xk, a, s);
  dim3 grid((a.D + DB - 1) / DB, (a.k + RB - 1) / RB, a.B);
  hipLaunchKernelGGL(k_direct_spectrum, grid, dim3(DB * RB), 0, s, x, xk, a);
  return hipGetLastError();
}

hipError_t launch_direct_filter(const cf* xk, const float* w_re, const float* w_im, int conj_w,
                                cf* sk, const DirectArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.k * a.D;
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_direct_filter, dim3(blocks), dim3(256), 0, s, xk, w_re, w_im, conj_w, sk, a);
  return hipGetLastError();
}

hipError_t launch_direct_synth(const cf* sk, const float* bias, float* y, const DirectArgs& a,
                               hipStream_t s) {
  if (a.B == 0 || a.N == 0) return hipSuccess;
  if (use_tiled(a)) return launch_tiled_synth(sk, bias, y, a, s);
  dim3 grid((a.D + DB - 1) / DB, (a.rows_present() + RB - 1) / RB, a.B);
  hipLaunchKernelGGL(k_direct_synth, grid, dim3(DB * RB), 0, s, sk, bias, y, a);
  return hipGetLastError();
}

// ---- edge bins of the band-group plans: a few bins, every row -----------------------------------
// X[b, f_i, d] = sum_n x[b,n,d] w_N^{f_i n} for the handful of bins f_i = f0 + i fstep.  Unlike
// k_direct_spectrum (one thread per bin walks a whole column) the ROWS are spread over the grid --
// blocks of 64 channels x 4 row lanes per (batch row, row chunk), eight rows in flight per thread --
// so x streams through once at memory speed.  Per-chunk partial sums go to `part` as doubles and are
// added in chunk order by k_edge_sum (deterministic).
constexpr int EB_MAX = 4;    // most bins accumulated per pass over x
template <int EB>
__global__ __launch_bounds__(256) void k_edge_partial(const float* __restrict__ x,
                                                      double* __restrict__ part, DirectArgs a,
                                                      int nch, int rows_per_chunk, int bin0) {
  __shared__ double red[4][EB][2][64];
  const int dl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dl, ch = blockIdx.y, b = blockIdx.z;
  constexpr int nb = EB;
  const int n0 = ch * rows_per_chunk, n1 = min(a.rows_present(), n0 + rows_per_chunk);
  double re[EB], im[EB];
  int idx[EB], stp[EB];
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    re[i] = 0.0; im[i] = 0.0;
    const long long f = a.f0 + (long long)(bin0 + i) * a.fstep;
    idx[i] = (int)((f * (n0 + rl)) % a.N);
    stp[i] = (int)((f * 4) % a.N);
  }
  if (d < a.D) {
    const float* xp = x + (size_t)b * a.rows_present() * a.D + d;
    int n = n0 + rl;
    for (; n + 4 * 7 < n1; n += 4 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(xp + (size_t)(n + 4 * u) * a.D);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          const cf w = a.tw[idx[i]];
          re[i] += (double)(v[u] * w.x); im[i] += (double)(v[u] * w.y);
          idx[i] += stp[i]; if (idx[i] >= a.N) idx[i] -= a.N;
        }
    }
    for (; n < n1; n += 4) {
      const float v = xp[(size_t)n * a.D];
#pragma unroll
      for (int i = 0; i < EB; ++i) {
        const cf w = a.tw[idx[i]];
        re[i] += (double)(v * w.x); im[i] += (double)(v * w.y);
        idx[i] += stp[i]; if (idx[i] >= a.N) idx[i] -= a.N;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < EB; ++i) { red[rl][i][0][dl] = re[i]; red[rl][i][1][dl] = im[i]; }
  __syncthreads();
  if (rl == 0 && d < a.D) {
    for (int i = 0; i < nb; ++i) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) { sr += red[r2][i][0][dl]; si += red[r2][i][1][dl]; }
      double* o = part + ((((size_t)ch * a.B + b) * a.k + bin0 + i) * a.D + d) * 2;
      o[0] = sr; o[1] = si;
    }
  }
}

__global__ void k_edge_sum(const double* __restrict__ part, cf* __restrict__ xk, DirectArgs a, int nch) {
  const size_t total = (size_t)a.B * a.k * a.D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    double sr = 0.0, si = 0.0;
    for (int c = 0; c < nch; ++c) { sr += part[(c * total + i) * 2]; si += part[(c * total + i) * 2 + 1]; }
    const int d = (int)(i % a.D);
    const int fi = (int)((i / a.D) % a.k);
    const size_t b = i / a.D / a.k;
    const size_t row = a.rows ? b * a.rows + (a.f0 + fi * a.fstep) : b * a.k + fi;
    xk[row * a.D + d] = real_bin(mk((float)sr, (float)si), a.f0 + fi * a.fstep, a.N);
  }
}

int edge_chunks(int B, int N, int D) {
  const long long base = (long long)B * ((D + 63) / 64);
  long long nch = (1024 + base - 1) / base;
  const long long cap = (N + 255) / 256;            // at least 256 rows per chunk
  if (nch > cap) nch = cap;
  return (int)(nch < 1 ? 1 : nch);
}

hipError_t launch_edge_spectrum(const float* x, cf* xk, double* part, const DirectArgs& a,
                                hipStream_t s) {
  if (a.k == 0 || a.B == 0) return hipSuccess;
  const int nch = edge_chunks(a.B, a.N, a.D);
  const int rpc = ((a.rows_present() + nch - 1) / nch + 3) & ~3;  // multiple of the 4 row lanes
  const dim3 grid((a.D + 63) / 64, nch, a.B), block(256);
  for (int bin0 = 0; bin0 < a.k;) {
    const int nb = a.k - bin0 >= 4 ? 4 : a.k - bin0 >= 2 ? 2 : 1;
    if (nb == 4) hipLaunchKernelGGL((k_edge_partial<4>), grid, block, 0, s, x, part, a, nch, rpc, bin0);
    else if (nb == 2) hipLaunchKernelGGL((k_edge_partial<2>), grid, block, 0, s, x, part, a, nch, rpc, bin0);
    else hipLaunchKernelGGL((k_edge_partial<1>), grid, block, 0, s, x, part, a, nch, rpc, bin0);
    bin0 += nb;
  }
  const size_t total = (size_t)a.B * a.k * a.D;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_edge_sum, dim3(blocks), dim3(256), 0, s, part, xk, a, nch);
  return hipGetLastError();
}

// y[b,n,d] += sum_i Re( S[b,i,d] conj(w_N^{f_i n}) ): eight rows per thread (stride 4 rows), the table
// index advanced incrementally, so the 64-bit modulo is paid once per thread and bin.
constexpr int ES_ROWS = 8;
__global__ __launch_bounds__(256) void k_edge_synth_acc(const cf* __restrict__ sk, float* __restrict__ y,
                                                        DirectArgs a) {
  const int dl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dl, b = blockIdx.z;
  const int n0 = blockIdx.y * (4 * ES_ROWS) + rl;
  const int R = a.rows_present();
  if (d >= a.D || n0 >= R) return;
  float acc[ES_ROWS];
#pragma unroll
  for (int u = 0; u < ES_ROWS; ++u) acc[u] = 0.f;
  const cf* sp = sk + (size_t)b * a.k * a.D + d;
  for (int fi = 0; fi < a.k; ++fi) {
    const long long f = a.f0 + (long long)fi * a.fstep;
    int idx = (int)((f * n0) % a.N);
    const int stp = (int)((f * 4) % a.N);
    const cf sv = sp[(size_t)fi * a.D];
#pragma unroll
    for (int u = 0; u < ES_ROWS; ++u) {
      const cf w = a.tw[idx];
      acc[u] += sv.x * w.x + sv.y * w.y;       // Re(s conj(w))
      idx += stp; if (idx >= a.N) idx -= a.N;
    }
  }
  float* yp = y + ((size_t)b * R + n0) * a.D + d;
  float old[ES_ROWS];
#pragma unroll
  for (int u = 0; u < ES_ROWS; ++u) old[u] = n0 + 4 * u < R ? yp[(size_t)(4 * u) * a.D] : 0.f;
#pragma unroll
  for (int u = 0; u < ES_ROWS; ++u)
    if (n0 + 4 * u < R) yp[(size_t)(4 * u) * a.D] = old[u] + acc[u];
}

hipError_t launch_edge_synth_acc(const cf* sk, float* y, const DirectArgs& a, hipStream_t s) {
  if (a.k == 0 || a.B == 0) return hipSuccess;
  dim3 grid((a.D + 63) / 64, (a.rows_present() + 4 * ES_ROWS - 1) / (4 * ES_ROWS), a.B);
  hipLaunchKernelGGL(k_edge_synth_acc, grid, dim3(256), 0, s, sk, y, a);
  return hipGetLastError();
}

// slab[b, f, d] = X[b, f, d] conj(G[b, i, d]) / N for the edge bins f = f0 + i fstep
__global__ void k_edge_slab(const cf* __restrict__ xk, const cf* __restrict__ ge, cf* __restrict__ slab,
                            int k_total, DirectArgs a) {
  const size_t total = (size_t)a.B * a.k * a.D;
  const float inv_n = (float)(1.0 / (double)a.N);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % a.D);
    const int fi = (int)((i / a.D) % a.k);
    const size_t b = i / a.D / a.k;
    const size_t row = (b * k_total + (a.f0 + fi * a.fstep)) * a.D + d;
    slab[row] = cscale(cmulc(xk[row], ge[i]), inv_n);
  }
}

hipError_t launch_edge_slab(const cf* xk, const cf* ge, cf* slab, int k_total, const DirectArgs& a,
                            hipStream_t s) {
  const size_t total = (size_t)a.B * a.k * a.D;
  if (total == 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_edge_slab, dim3(blocks), dim3(256), 0, s, xk, ge, slab, k_total, a);
  return hipGetLastError();
}

// wt[f, d] = (w_re[d, f], w_im[d, f]) for f < k: the filter in the layout the unpack phase reads it in
// (FilterArgs::wt).  32 x 32 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void k_pack_w(const float* __restrict__ w_re,
                                                const float* __restrict__ w_im, cf* __restrict__ wt,
                                                int D, int F, int k) {
  __shared__ float tre[32][33], tim[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int f0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  for (int dy = ty; dy < 32; dy += 8) {
    const int d = d0 + dy, f = f0 + tx;
    const bool ok = d < D && f < k;
    tre[dy][tx] = ok ? w_re[(size_t)d * F + f] : 0.f;
    tim[dy][tx] = ok ? w_im[(size_t)d * F + f] : 0.f;
  }
  __syncthreads();
  for (int fy = ty; fy < 32; fy += 8) {
    const int f = f0 + fy, d = d0 + tx;
    if (f < k && d < D) wt[(size_t)f * D + d] = mk(tre[tx][fy], tim[tx][fy]);
  }
}

// out[b, f, d] = in[b, f, d] * scale * (hermitian && 0 < f < N/2 ? 2 : 1): the bin weights of the transform
// pair smx_rfft_ex / smx_irfft_ex on the plans that do not apply them in their own kernels (in == out allowed)
__global__ void k_scale_bins(const cf* __restrict__ in, cf* __restrict__ out, long long total, int k, int D, int N,
                             float scale, int hermitian) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)((i / D) % k);
    const float w = (hermitian && f != 0 && 2 * f != N) ? 2.f * scale : scale;
    out[i] = cscale(in[i], w);
  }
}
hipError_t launch_scale_bins(const cf* in, cf* out, int B, int k, int D, int N, float scale, int hermitian,
                             hipStream_t s) {
  const long long total = (long long)B * k * D;
  if (total == 0) return hipSuccess;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(k_scale_bins, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, in, out,
                     total, k, D, N, scale, hermitian);
  return hipGetLastError();
}

hipError_t launch_pack_w(const float* w_re, const float* w_im, cf* wt, int D, int F, int k,
                         hipStream_t s) {
  if (k == 0 || D == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_w, dim3((k + 31) / 32, (D + 31) / 32), dim3(256), 0, s, w_re, w_im, wt, D, F, k);
  return hipGetLastError();
}

// ---- parameter gradients ---------------------------------------------------------------------
// grad_w_real[d,f] = sum_b Re P[b,f,d] ; grad_w_imag[d,f] = -sum_b Im P[b,f,d] ; columns >= k zero.
// Block = one bin f x 32 channels x 8 batch groups: every thread sums a contiguous run of batch
// rows (independent loads, all in flight), then the 8 partial sums are added in group order through
// LDS.  The order of additions is fixed, so the result is bitwise reproducible.  k*D/32 blocks.
constexpr int GW_G = 8;
template <bool FROM_SPECTRA>
__global__ __launch_bounds__(256) void k_gradw(const cf* __restrict__ p0, const cf* __restrict__ p1,
                                               const float* __restrict__ gb_part,
                                               float* __restrict__ gw_re, float* __restrict__ gw_im,
                                               float* __restrict__ gbias, int B, int D, int F, int k,
                                               float inv_n) {
  __shared__ float pre[GW_G][32], pim[GW_G][32];
  const int tx = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int f = blockIdx.y, d = blockIdx.x * 32 + tx;
  const bool bias_row = (f == F);            // extra grid row: grad_bias[d] = sum_b Re G[b,0,d]
  const int per = (B + GW_G - 1) / GW_G;
  const int b0 = grp * per, b1 = min(B, b0 + per);
  float re = 0.f, im = 0.f;
  if (d < D && (bias_row ? (!FROM_SPECTRA || k > 0) : f < k)) {
    const size_t o = bias_row ? (size_t)d : (size_t)f * D + d;
    const size_t bs = bias_row && !FROM_SPECTRA ? (size_t)D : (size_t)k * D;
    auto term = [&](int b) -> cf {
      const size_t a = o + (size_t)b * bs;
      if (bias_row) return FROM_SPECTRA ? mk(p1[a].x, 0.f) : mk(gb_part[a], 0.f);
      return FROM_SPECTRA ? cmulc(p0[a], p1[a]) : p0[a];
    };
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      cf v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = term(b + u);
#pragma unroll
      for (int u = 0; u < 8; ++u) { re += v[u].x; im += v[u].y; }
    }
    for (; b < b1; ++b) { const cf pr = term(b); re += pr.x; im += pr.y; }
  }
  pre[grp][tx] = re; pim[grp][tx] = im;
  __syncthreads();
  if (grp == 0 && d < D) {
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < GW_G; ++g2) { sr += pre[g2][tx]; si += pim[g2][tx]; }
    if (bias_row) {
      gbias[d] = sr;
    } else {
      if (FROM_SPECTRA) { sr *= inv_n; si *= inv_n; }
      gw_re[(size_t)d * F + f] = sr;
      gw_im[(size_t)d * F + f] = -si;
    }
  }
}

// The slab reduction with NBINS bins per block (round 3): every thread keeps NBINS running sums and has
// (32 / NBINS) rows x NBINS bins = 32 loads in flight, so a block moves 4 x (C2) to 8 x the bytes per latency of the
// one-bin blocks of k_gradw<false> above -- which waited 67 % of their wave-cycles at C5 (20.0 us for 68 MB; now 15.1 us
// = 4.5 TB/s; C2: 6.0 -> 4.8 us).  Same
// additions in the same order (a thread's rows ascending, the 8 group sums in group order): bit-identical results.
template <int NBINS>
__global__ __launch_bounds__(256) void k_gradw_slab(const cf* __restrict__ pslab, const float* __restrict__ gb_part,
                                                    float* __restrict__ gw_re, float* __restrict__ gw_im,
                                                    float* __restrict__ gbias, int B, int D, int F, int k) {
  __shared__ float pre[NBINS][GW_G][32], pim[NBINS][GW_G][32];
  const int tx = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + tx, nfb = (F + NBINS - 1) / NBINS;
  const bool bias_row = (int)blockIdx.y == nfb;
  const int f0 = blockIdx.y * NBINS;
  const int per = (B + GW_G - 1) / GW_G;
  const int b0 = grp * per, b1 = min(B, b0 + per);
  float re[NBINS], im[NBINS];
#pragma unroll
  for (int i = 0; i < NBINS; ++i) { re[i] = 0.f; im[i] = 0.f; }
  if (d < D) {
    if (bias_row) {
      const float* gp = gb_part + d;
      int b = b0;
      for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gp[(size_t)(b + u) * D];
#pragma unroll
        for (int u = 0; u < 8; ++u) re[0] += v[u];
      }
      for (; b < b1; ++b) re[0] += gp[(size_t)b * D];
    } else {
      const cf* sp = pslab + (size_t)f0 * D + d;
      const size_t bs = (size_t)k * D;
      const int nb = min(NBINS, k - f0);                 // bins of this block that exist (<= 0: all zero)
      constexpr int RB = 32 / NBINS;
      int b = b0;
      for (; b + RB <= b1; b += RB) {
        cf v[RB][NBINS];
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
          for (int i = 0; i < NBINS; ++i) v[u][i] = i < nb ? sp[(size_t)(b + u) * bs + (size_t)i * D] : mk(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
          for (int i = 0; i < NBINS; ++i) { re[i] += v[u][i].x; im[i] += v[u][i].y; }
      }
      for (; b < b1; ++b)
#pragma unroll
        for (int i = 0; i < NBINS; ++i)
          if (i < nb) { const cf v = sp[(size_t)b * bs + (size_t)i * D]; re[i] += v.x; im[i] += v.y; }
    }
  }
#pragma unroll
  for (int i = 0; i < NBINS; ++i) { pre[i][grp][tx] = re[i]; pim[i][grp][tx] = im[i]; }
  __syncthreads();
  if (d < D && grp < NBINS) {                            // thread (tx, grp) finishes bin f0 + grp
    const int f = f0 + grp;
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < GW_G; ++g2) { sr += pre[grp][g2][tx]; si += pim[grp][g2][tx]; }
    if (bias_row) {
      if (grp == 0) gbias[d] = sr;
    } else if (f < F) {
      gw_re[(size_t)d * F + f] = sr;
      gw_im[(size_t)d * F + f] = -si;
    }
  }
}

hipError_t launch_gradw_slab(const cf* pslab, const float* gb_part, float* gw_re, float* gw_im,
                             float* gbias, int B, int D, int F, int k, hipStream_t s) {
  constexpr int NBINS = 4;        // (8 bins per block measured the same at C5: 15.3 against 15.1 us)
  dim3 grid((D + 31) / 32, (F + NBINS - 1) / NBINS + (gbias ? 1 : 0));
  hipLaunchKernelGGL((k_gradw_slab<NBINS>), grid, dim3(256), 0, s, pslab, gb_part, gw_re, gw_im, gbias, B, D, F, k);
  return hipGetLastError();
}

hipError_t launch_gradw_spectra(const cf* xk, const cf* gk, float* gw_re, float* gw_im,
                                float* gbias, int B, int N, int D, int F, int k, hipStream_t s) {
  dim3 grid((D + 31) / 32, F + (gbias ? 1 : 0));
  hipLaunchKernelGGL((k_gradw<true>), grid, dim3(256), 0, s, xk, gk, (const float*)nullptr, gw_re,
                     gw_im, gbias, B, D, F, k, (float)(1.0 / (double)N));
  return hipGetLastError();
}

// ---- wirtinger_ops.WirtingerSpectralFilter (complex in, complex out) ----------------------------
// out[b,n,d] = n < k ? x[b,n,d] * W[d,n] : 0       (conj_w: multiply by conj(W) -> grad_x of the filter)
// One workgroup per row (b, n): rows n >= k are a streaming zero fill, rows n < k a 16-byte-per-lane
// multiply -- no per-element division.  The kernel is a 1 GiB write at C5 (64, 4096, 512): HBM-bound.
__global__ __launch_bounds__(256) void k_wfilter(const cf* __restrict__ xf, const float* __restrict__ w_re,
                                                 const float* __restrict__ w_im, int conj_w,
                                                 cf* __restrict__ out, int N, int D, int F, int k,
                                                 long long rows) {
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row % N);
    cf* o = out + (size_t)row * D;
    if (n >= k) {
      if ((D & 1) == 0 && ((uintptr_t)o & 15) == 0) {
        f32x4 z; z.x = z.y = z.z = z.w = 0.f;
        for (int d = threadIdx.x * 2; d < D; d += 512)
          __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(o + d));
      } else {
        for (int d = threadIdx.x; d < D; d += 256) o[d] = mk(0.f, 0.f);
      }
      continue;
    }
    const cf* xi = xf + (size_t)row * D;
    for (int d = threadIdx.x; d < D; d += 256) {
      cf w = mk(w_re[(size_t)d * F + n], w_im[(size_t)d * F + n]);
      if (conj_w) w = cconj(w);
      o[d] = cmul(xi[d], w);
    }
  }
}

// grad_w[d,f] = sum_b g[b,f,d] * conj(x[b,f,d]), f < k ; real part -> grad of .real, imag -> grad of .imag
// Block = one bin f x 32 channels x 8 batch groups (the layout of k_gradw): every thread sums a run of
// batch rows with its loads in flight, the groups are added in fixed order -> deterministic.
__global__ __launch_bounds__(256) void k_wfilter_gradw(const cf* __restrict__ xf,
                                                       const cf* __restrict__ gf,
                                                       float* __restrict__ gw_re,
                                                       float* __restrict__ gw_im, int B, int N,
                                                       int D, int F, int k) {
  __shared__ float pre[GW_G][32], pim[GW_G][32];
  const int tx = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int f = blockIdx.y, d = blockIdx.x * 32 + tx;
  const int per = (B + GW_G - 1) / GW_G;
  const int b0 = grp * per, b1 = min(B, b0 + per);
  float re = 0.f, im = 0.f;
  if (d < D && f < k) {
    const size_t o = (size_t)f * D + d, bs = (size_t)N * D;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
      cf gv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { gv[u] = gf[o + (size_t)(b + u) * bs]; xv[u] = xf[o + (size_t)(b + u) * bs]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const cf pr = cmulc(gv[u], xv[u]); re += pr.x; im += pr.y; }
    }
    for (; b < b1; ++b) { const cf pr = cmulc(gf[o + (size_t)b * bs], xf[o + (size_t)b * bs]); re += pr.x; im += pr.y; }
  }
  pre[grp][tx] = re; pim[grp][tx] = im;
  __syncthreads();
  if (grp == 0 && d < D) {
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < GW_G; ++g2) { sr += pre[g2][tx]; si += pim[g2][tx]; }
    gw_re[(size_t)d * F + f] = f < k ? sr : 0.f;
    gw_im[(size_t)d * F + f] = f < k ? si : 0.f;
  }
}

hipError_t launch_wfilter(const cf* xf, const float* w_re, const float* w_im, int conj_w, cf* out,
                          int B, int N, int D, int F, int k, hipStream_t s) {
  const long long rows = (long long)B * N;
  if (rows == 0 || D == 0) return hipSuccess;
  const int blocks = (int)(rows < (1ll << 20) ? rows : (1ll << 20));
  hipLaunchKernelGGL(k_wfilter, dim3(blocks), dim3(256), 0, s, xf, w_re, w_im, conj_w, out, N, D, F, k, rows);
  return hipGetLastError();
}

hipError_t launch_wfilter_gradw(const cf* xf, const cf* gf, float* gw_re, float* gw_im, int B,
                                int N, int D, int F, int k, hipStream_t s) {
  dim3 grid((D + 31) / 32, F);
  hipLaunchKernelGGL(k_wfilter_gradw, grid, dim3(256), 0, s, xf, gf, gw_re, gw_im, B, N, D, F, k);
  return hipGetLastError();
}

// ---- wirtinger_ops.WirtingerGradient: out = x * w with w broadcast over the leading batch -------
// grid.y walks the batch, grid.x the inner index: no modulo per element; two complex per lane when aligned
__global__ __launch_bounds__(256) void k_cmul(const cf* __restrict__ x, const cf* __restrict__ w, int conj_w,
                                              cf* __restrict__ out, long long batch, long long inner) {
  const bool vec = (inner & 1) == 0 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0;
  for (long long b = blockIdx.y; b < batch; b += gridDim.y) {
    const cf* xb = x + b * inner;
    cf* ob = out + b * inner;
    if (vec) {
      for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < inner; i += (long long)gridDim.x * 512) {
        float x0, x1, x2, x3, w0, w1, w2, w3;
        ld4(reinterpret_cast<const float*>(xb + i), x0, x1, x2, x3);
        ld4(reinterpret_cast<const float*>(w + i), w0, w1, w2, w3);
        const cf a = conj_w ? cmulc(mk(x0, x1), mk(w0, w1)) : cmul(mk(x0, x1), mk(w0, w1));
        const cf c = conj_w ? cmulc(mk(x2, x3), mk(w2, w3)) : cmul(mk(x2, x3), mk(w2, w3));
        st4(reinterpret_cast<float*>(ob + i), a.x, a.y, c.x, c.y);
      }
    } else {
      for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < inner; i += (long long)gridDim.x * 256)
        ob[i] = conj_w ? cmulc(xb[i], w[i]) : cmul(xb[i], w[i]);
    }
  }
}

// gw[i] = sum_b g[b,i] * conj(x[b,i]): 32 inner elements x 8 batch groups per block, fixed-order sum
__global__ __launch_bounds__(256) void k_cmul_gradw(const cf* __restrict__ x, const cf* __restrict__ g,
                                                    cf* __restrict__ gw, long long batch, long long inner) {
  __shared__ float pre[GW_G][32], pim[GW_G][32];
  const int tx = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long per = (batch + GW_G - 1) / GW_G;
  const long long b0 = grp * per, b1 = b0 + per < batch ? b0 + per : batch;
  for (long long i0 = (long long)blockIdx.x * 32; i0 < inner; i0 += (long long)gridDim.x * 32) {
    const long long i = i0 + tx;
    float re = 0.f, im = 0.f;
    if (i < inner) {
      long long b = b0;
      for (; b + 4 <= b1; b += 4) {
        cf gv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { gv[u] = g[(b + u) * inner + i]; xv[u] = x[(b + u) * inner + i]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const cf pr = cmulc(gv[u], xv[u]); re += pr.x; im += pr.y; }
      }
      for (; b < b1; ++b) { const cf pr = cmulc(g[b * inner + i], x[b * inner + i]); re += pr.x; im += pr.y; }
    }
    __syncthreads();
    pre[grp][tx] = re; pim[grp][tx] = im;
    __syncthreads();
    if (grp == 0 && i < inner) {
      float sr = 0.f, si = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < GW_G; ++g2) { sr += pre[g2][tx]; si += pim[g2][tx]; }
      gw[i] = mk(sr, si);
    }
  }
}

hipError_t launch_cmul(const cf* x, const cf* w, int conj_w, cf* out, long long batch,
                       long long inner, hipStream_t s) {
  if (batch * inner == 0) return hipSuccess;
  const long long bx = (inner + 511) / 512;
  dim3 grid((unsigned)(bx < 4096 ? bx : 4096), (unsigned)(batch < 65535 ? batch : 65535));
  hipLaunchKernelGGL(k_cmul, grid, dim3(256), 0, s, x, w, conj_w, out, batch, inner);
  return hipGetLastError();
}

hipError_t launch_cmul_gradw(const cf* x, const cf* g, cf* gw, long long batch, long long inner,
                             hipStream_t s) {
  if (inner == 0) return hipSuccess;
  const long long bx = (inner + 31) / 32;
  hipLaunchKernelGGL(k_cmul_gradw, dim3((unsigned)(bx < (1 << 20) ? bx : (1 << 20))), dim3(256), 0, s, x, g, gw,
                     batch, inner);
  return hipGetLastError();
}

// ---- tiled pruned DFT for the shapes the decimated kernels do not take -----------------------------
// N % 256 != 0 (1000, 1500, 4000, 128 ...) or an odd channel count: the transform is evaluated as the
// matrix product it is -- X (k x D) = Wf (k x N) x (N x D) per batch row and y (N x D) = Re(conj Wf^T S) --
// with workgroup tiles staged through LDS and 4 x 4 register tiles per thread, fp32 FMAs with two-level
// accumulation (128-row chunks summed into a second accumulator).  O(N k) per column like the literal
// kernels above, but x / S stream through LDS once per 32 bins / rows and the twiddles are built once per
// tile instead of once per thread: ~100x faster at (64, 4000, 256).  The table index (f n mod N) is exact.
constexpr int TD_BINS = 32;      // bins (spectrum) or rows (synthesis) per workgroup tile
constexpr int TD_CH = 128;       // channels per workgroup tile
constexpr int TD_K = 32;         // reduction chunk: rows (spectrum) or bins (synthesis) per LDS stage

// tw_s[i][c] = w_N^{(p0 + i) (q0 + c)}: 32 x 32 twiddles by 256 threads, 4 consecutive c each.  The
// exact table index (p q mod N) is kept incrementally: one 64-bit modulo per thread at the start, then
// additions with a conditional subtract as q advances by 1 inside a stage and by 32 between stages.
struct TwStage {
  int i, c0, pm, cur, step32, N;
  __device__ __forceinline__ void init(int tid, long long p0, int n) {
    N = n;
    i = (tid * 4) / TD_K; c0 = (tid * 4) % TD_K;
    pm = (int)((p0 + i) % N);
    cur = (int)(((long long)pm * c0) % N);
    step32 = (int)(((long long)pm * TD_K) % N);
  }
  // stage the chunk that starts at q0 = 32 * (number of earlier calls); pcnt / qcnt = valid extents
  __device__ __forceinline__ void stage(cf (*tw_s)[TD_K + 1], const cf* __restrict__ tw, int pcnt, int qcnt) {
    int idx = cur;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tw_s[i][c0 + e] = (i < pcnt && c0 + e < qcnt) ? tw[idx] : mk(0.f, 0.f);
      idx += pm; if (idx >= N) idx -= N;
    }
    cur += step32; if (cur >= N) cur -= N;
  }
  // the same in two halves, so that the table loads of the NEXT chunk can be in flight during the products
  __device__ __forceinline__ void fetch(cf (&v)[4], const cf* __restrict__ tw, int pcnt, int qcnt) {
    int idx = cur;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (i < pcnt && c0 + e < qcnt) ? tw[idx] : mk(0.f, 0.f);
      idx += pm; if (idx >= N) idx -= N;
    }
    cur += step32; if (cur >= N) cur -= N;
  }
  __device__ __forceinline__ void put(cf (*tw_s)[TD_K + 1], const cf (&v)[4]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) tw_s[i][c0 + e] = v[e];
  }
};

// X[b, f, d] = sum_{n < R} x[b, n, d] w_N^{f n},  f < k
__global__ __launch_bounds__(256) void k_tiled_spectrum(const float* __restrict__ x, cf* __restrict__ xk,
                                                        DirectArgs a) {
  __shared__ float xs[TD_K][TD_CH];
  __shared__ cf tw_s[TD_BINS][TD_K + 1];
  const int tid = threadIdx.x, fg = tid >> 5, dg = tid & 31;         // 8 bin groups x 32 channel groups
  const int d0 = blockIdx.x * TD_CH, f0 = blockIdx.y * TD_BINS, b = blockIdx.z;
  const int R = a.rows_present();
  const float* xb = x + (size_t)b * R * a.D;
  cf acc[4][4], part[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc[i][c] = mk(0.f, 0.f); part[i][c] = mk(0.f, 0.f); }
  const int fcnt = min(TD_BINS, a.k - f0);
  TwStage ts;
  ts.init(tid, f0, a.N);
  for (int n0 = 0; n0 < R; n0 += TD_K) {
    const int ncnt = min(TD_K, R - n0);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {                 // 16 coalesced loads per thread
      const int idx = e * 256 + tid, r = idx / TD_CH, c = idx % TD_CH;
      xs[r][c] = (r < ncnt && d0 + c < a.D) ? xb[(size_t)(n0 + r) * a.D + d0 + c] : 0.f;
    }
    ts.stage(tw_s, a.tw, fcnt, ncnt);
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < TD_K; ++r) {
      float xv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = xs[r][dg * 4 + c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const cf w = tw_s[fg * 4 + i][r];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          part[i][c].x = fmaf(xv[c], w.x, part[i][c].x);
          part[i][c].y = fmaf(xv[c], w.y, part[i][c].y);
        }
      }
    }
    if ((n0 / TD_K) % 4 == 3 || n0 + TD_K >= R) {                    // fold every 128 rows
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[i][c] = cadd(acc[i][c], part[i][c]); part[i][c] = mk(0.f, 0.f); }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + fg * 4 + i;
    if (f >= a.k) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = d0 + dg * 4 + c;
      if (d < a.D) xk[((size_t)b * a.k + f) * a.D + d] = real_bin(acc[i][c], f, a.N);
    }
  }
}

// y[b, n, d] = bias[d] + sum_{f < k} Re( S[b, f, d] conj(w_N^{f n}) ),  n < R
__global__ __launch_bounds__(256) void k_tiled_synth(const cf* __restrict__ sk, const float* __restrict__ bias,
                                                     float* __restrict__ y, DirectArgs a) {
  __shared__ cf ss[TD_K][TD_CH];
  __shared__ cf tw_s[TD_BINS][TD_K + 1];                             // [row][bin]
  const int tid = threadIdx.x, ng = tid >> 5, dg = tid & 31;
  const int d0 = blockIdx.x * TD_CH, n0 = blockIdx.y * TD_BINS, b = blockIdx.z;
  const int R = a.rows_present();
  float acc[4][4], part[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc[i][c] = 0.f; part[i][c] = 0.f; }
  const int ncnt = min(TD_BINS, R - n0);
  const cf* sb = sk + (size_t)b * a.k * a.D;
  TwStage ts;                                                        // rows fixed, bins advance
  ts.init(tid, n0, a.N);
  for (int f0 = 0; f0 < a.k; f0 += TD_K) {
    const int fcnt = min(TD_K, a.k - f0);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {
      const int idx = e * 256 + tid, r = idx / TD_CH, c = idx % TD_CH;
      ss[r][c] = (r < fcnt && d0 + c < a.D) ? sb[(size_t)(f0 + r) * a.D + d0 + c] : mk(0.f, 0.f);
    }
    ts.stage(tw_s, a.tw, ncnt, fcnt);      // tw_s[i][c] = w_N^{(n0 + i)(f0 + c)}
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < TD_K; ++r) {
      cf sv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) sv[c] = ss[r][dg * 4 + c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const cf w = tw_s[ng * 4 + i][r];
#pragma unroll
        for (int c = 0; c < 4; ++c) part[i][c] = fmaf(sv[c].x, w.x, fmaf(sv[c].y, w.y, part[i][c]));
      }
    }
    if ((f0 / TD_K) % 4 == 3 || f0 + TD_K >= a.k) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[i][c] += part[i][c]; part[i][c] = 0.f; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ng * 4 + i;
    if (n >= R) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = d0 + dg * 4 + c;
      if (d < a.D) y[((size_t)b * R + n) * a.D + d] = acc[i][c] + (bias ? bias[d] : 0.f);
    }
  }
}

// ---- the same two products on the matrix cores -----------------------------------------------------
// v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain -- no precision is traded.  fp32 MFMA
// peaks at the fp32 vector rate on gfx950 (157 TFLOP/s), but a wave issues ONE instruction per 64 cycles for
// 4096 flops where the VALU tile above spends 16 FMAs + operand reads per 32 flops: the measured gain is
// the issue overhead, not a higher peak (guide: 122 vs 52 TFLOP/s on a 4096^3 GEMM).
// Workgroup tile as above (32 bins or rows x 128 channels); wave w owns channels [32 w, 32 w + 32).
// Lane l supplies A[i = l & 31][kk = l >> 5] and B[kk = l >> 5][j = l & 31]; it receives
// C[row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)][col = l & 31].
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma_spectrum(const float* __restrict__ x, cf* __restrict__ xk,
                                                       DirectArgs a) {
  __shared__ float xs[TD_K][TD_CH];
  __shared__ cf tw_s[TD_BINS][TD_K + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d0 = blockIdx.x * TD_CH, f0 = blockIdx.y * TD_BINS, b = blockIdx.z;
  const int R = a.rows_present();
  const float* xb = x + (size_t)b * R * a.D;
  f32x16 acc_re = {0}, acc_im = {0}, part_re = {0}, part_im = {0};
  const int fcnt = min(TD_BINS, a.k - f0);
  TwStage ts;
  ts.init(tid, f0, a.N);
  const int ai = lane & 31, kk = lane >> 5;
  // chunk c + 1 travels from global memory to registers while chunk c is multiplied out of LDS
  float xr[(TD_K * TD_CH) / 256];
  cf twr[4];
  auto fetch = [&](int n0) {
    const int ncnt = min(TD_K, R - n0);
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {
      const int idx = e * 256 + tid, r = idx / TD_CH, c = idx % TD_CH;
      xr[e] = (r < ncnt && d0 + c < a.D) ? xb[(size_t)(n0 + r) * a.D + d0 + c] : 0.f;
    }
    ts.fetch(twr, a.tw, fcnt, ncnt);
  };
  fetch(0);
  for (int n0 = 0; n0 < R; n0 += TD_K) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {
      const int idx = e * 256 + tid;
      xs[idx / TD_CH][idx % TD_CH] = xr[e];
    }
    ts.put(tw_s, twr);
    __syncthreads();
    if (n0 + TD_K < R) fetch(n0 + TD_K);
#pragma unroll
    for (int s2 = 0; s2 < TD_K / 2; ++s2) {
      const cf w = tw_s[ai][2 * s2 + kk];
      const float bv = xs[2 * s2 + kk][wv * 32 + ai];
      part_re = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, bv, part_re, 0, 0, 0);
      part_im = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, bv, part_im, 0, 0, 0);
    }
    if ((n0 / TD_K) % 4 == 3 || n0 + TD_K >= R) {                    // fold every 128 rows
      acc_re += part_re; acc_im += part_im;
      part_re = f32x16{0}; part_im = f32x16{0};
    }
  }
  const int d = d0 + wv * 32 + ai;
  if (d < a.D) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = f0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (f < a.k) xk[((size_t)b * a.k + f) * a.D + d] = real_bin(mk(acc_re[r], acc_im[r]), f, a.N);
    }
  }
}

__global__ __launch_bounds__(256) void k_mfma_synth(const cf* __restrict__ sk, const float* __restrict__ bias,
                                                    float* __restrict__ y, DirectArgs a) {
  __shared__ cf ss[TD_K][TD_CH];
  __shared__ cf tw_s[TD_BINS][TD_K + 1];                             // [row][bin]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d0 = blockIdx.x * TD_CH, n0 = blockIdx.y * TD_BINS, b = blockIdx.z;
  const int R = a.rows_present();
  f32x16 acc = {0}, part = {0};
  const int ncnt = min(TD_BINS, R - n0);
  const cf* sb = sk + (size_t)b * a.k * a.D;
  TwStage ts;
  ts.init(tid, n0, a.N);
  const int ai = lane & 31, kk = lane >> 5;
  cf sr[(TD_K * TD_CH) / 256];
  cf twr[4];
  auto fetch = [&](int f0) {
    const int fcnt = min(TD_K, a.k - f0);
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {
      const int idx = e * 256 + tid, r = idx / TD_CH, c = idx % TD_CH;
      sr[e] = (r < fcnt && d0 + c < a.D) ? sb[(size_t)(f0 + r) * a.D + d0 + c] : mk(0.f, 0.f);
    }
    ts.fetch(twr, a.tw, ncnt, fcnt);
  };
  if (a.k > 0) fetch(0);
  for (int f0 = 0; f0 < a.k; f0 += TD_K) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (TD_K * TD_CH) / 256; ++e) {
      const int idx = e * 256 + tid;
      ss[idx / TD_CH][idx % TD_CH] = sr[e];
    }
    ts.put(tw_s, twr);
    __syncthreads();
    if (f0 + TD_K < a.k) fetch(f0 + TD_K);
#pragma unroll
    for (int s2 = 0; s2 < TD_K / 2; ++s2) {
      const cf w = tw_s[ai][2 * s2 + kk];
      const cf sv = ss[2 * s2 + kk][wv * 32 + ai];
      part = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, sv.x, part, 0, 0, 0);   // Re(s conj(w)) = s.x w.x + s.y w.y
      part = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, sv.y, part, 0, 0, 0);
    }
    if ((f0 / TD_K) % 4 == 3 || f0 + TD_K >= a.k) { acc += part; part = f32x16{0}; }
  }
  const int d = d0 + wv * 32 + ai;
  if (d < a.D) {
    const float bv = bias ? bias[d] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (n < R) y[((size_t)b * R + n) * a.D + d] = acc[r] + bv;
    }
  }
}

// ---- the same two products with bf16 x 3 operands (round 3) --------------------------------------------
// Each fp32 operand is split into three bf16 terms, v = v1 + v2 + v3 (v1 = bf16(v), v2 = bf16(v - v1),
// v3 = bf16(v - v1 - v2): 24 mantissa bits).  A product x w is the six terms x1 w1 + (x1 w2 + x2 w1) +
// (x1 w3 + x3 w1 + x2 w2); the three dropped ones are below 2^-32 relative.  Every bf16 x bf16 product is EXACT
// in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, so what remains is fp32 summation error -- the same as the
// f32 MFMA kernels above (measured 2e-7 on the suite's shapes) -- while six bf16 MFMAs of 32 cycles do the work
// of eight f32 ones of 64 (2.7x per product).  The leading term and the five small ones go to separate
// accumulators (the small ones would otherwise be rounded against the large running sum).
// LDS holds the operands already split and in the MFMA's own order: a lane reads its 8 consecutive k of one
// row / column with one ds_read_b128 (row pitch 80 B: 16-byte aligned, conflict-free for the 16-lane groups).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int BX_PITCH = TD_K + 8;     // bf16 elements per LDS row

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  float r = v - (float)a;
  b = (__bf16)r;
  r -= (float)b;
  c = (__bf16)r;
}
// two consecutive k of one row: three packed 32-bit words (one per split level)
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned (&w)[3]) {
  __bf16 a0, b0, c0, a1, b1, c1;
  split3(v0, a0, b0, c0);
  split3(v1, a1, b1, c1);
  bf16x2 p;
  p = bf16x2{a0, a1}; w[0] = __builtin_bit_cast(unsigned, p);
  p = bf16x2{b0, b1}; w[1] = __builtin_bit_cast(unsigned, p);
  p = bf16x2{c0, c1}; w[2] = __builtin_bit_cast(unsigned, p);
}
// hi += a1 b1 ; lo += a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& hi, f32x16& lo) {
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
}

// Workgroup tile: BX_T x 32 bins (spectrum) or rows (synthesis) x 128 channels; a wave owns 32 channels and ALL the
// tile's bin / row blocks, so one B fragment (x or S) feeds BX_T A fragments and x / S are re-read k / (32 BX_T)
// times instead of k / 32.
constexpr int BX_T = 1;      // (2 measured: no gain at (64,4000,256), 30 % slower at (16,4000,255) -- the operand split, not the MFMAs, is what costs)

// X[b, f, d] = sum_{n < R} x[b, n, d] w_N^{f n}: A = twiddles [bin][n] (re and im), B = x [n][channel]
__global__ __launch_bounds__(256) void k_bx3_spectrum(const float* __restrict__ x, cf* __restrict__ xk,
                                                      DirectArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][2][BX_T * 32][BX_PITCH];   // [level][re|im][bin][n]
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][TD_CH][BX_PITCH];          // [level][channel][n]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d0 = blockIdx.x * TD_CH, f0 = blockIdx.y * (BX_T * 32), b = blockIdx.z;
  const int R = a.rows_present();
  const float* xb = x + (size_t)b * R * a.D;
  f32x16 hi_re[BX_T], lo_re[BX_T], hi_im[BX_T], lo_im[BX_T];
#pragma unroll
  for (int q = 0; q < BX_T; ++q) { hi_re[q] = f32x16{0}; lo_re[q] = f32x16{0}; hi_im[q] = f32x16{0}; lo_im[q] = f32x16{0}; }
  TwStage ts[BX_T];                             // thread: bin 32 q + ts.i, rows ts.c0 .. ts.c0 + 3 of the chunk
  int fcnt[BX_T];
#pragma unroll
  for (int q = 0; q < BX_T; ++q) {
    ts[q].init(tid, f0 + 32 * q, a.N);
    fcnt[q] = max(0, min(32, a.k - f0 - 32 * q));
  }
  const int r31 = lane & 31, h = lane >> 5;
  // x tile: thread (c = tid & 127, half = tid >> 7) takes rows 4 e + 2 half, + 1 of channel c, e = 0 .. 7
  const int xc = tid & 127, xh = tid >> 7;
  float xr[16];
  cf twr[BX_T][4];
  auto fetch = [&](int n0) {
    const int ncnt = min(TD_K, R - n0);
    const bool cok = d0 + xc < a.D;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r0 = 4 * e + 2 * xh;
      xr[2 * e] = (cok && r0 < ncnt) ? xb[(size_t)(n0 + r0) * a.D + d0 + xc] : 0.f;
      xr[2 * e + 1] = (cok && r0 + 1 < ncnt) ? xb[(size_t)(n0 + r0 + 1) * a.D + d0 + xc] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < BX_T; ++q) ts[q].fetch(twr[q], a.tw, fcnt[q], ncnt);
  };
  fetch(0);
  for (int n0 = 0; n0 < R; n0 += TD_K) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned w[3];
      split3_pair(xr[2 * e], xr[2 * e + 1], w);
      const int r0 = 4 * e + 2 * xh;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(&Bs[p][xc][r0]) = w[p];
    }
#pragma unroll
    for (int q = 0; q < BX_T; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned wre[3], wim[3];
        split3_pair(twr[q][2 * e].x, twr[q][2 * e + 1].x, wre);
        split3_pair(twr[q][2 * e].y, twr[q][2 * e + 1].y, wim);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          *reinterpret_cast<unsigned*>(&As[p][0][32 * q + ts[q].i][ts[q].c0 + 2 * e]) = wre[p];
          *reinterpret_cast<unsigned*>(&As[p][1][32 * q + ts[q].i][ts[q].c0 + 2 * e]) = wim[p];
        }
      }
    __syncthreads();
    if (n0 + TD_K < R) fetch(n0 + TD_K);
#pragma unroll
    for (int s2 = 0; s2 < TD_K / 16; ++s2) {
      bf16x8 bx[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bx[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wv * 32 + r31][16 * s2 + 8 * h]);
#pragma unroll
      for (int q = 0; q < BX_T; ++q) {
        bf16x8 are[3], aim[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          are[p] = *reinterpret_cast<const bf16x8*>(&As[p][0][32 * q + r31][16 * s2 + 8 * h]);
          aim[p] = *reinterpret_cast<const bf16x8*>(&As[p][1][32 * q + r31][16 * s2 + 8 * h]);
        }
        mfma6(are, bx, hi_re[q], lo_re[q]);
        mfma6(aim, bx, hi_im[q], lo_im[q]);
      }
    }
  }
  const int d = d0 + wv * 32 + r31;
  if (d < a.D) {
#pragma unroll
    for (int q = 0; q < BX_T; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = f0 + 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (f < a.k)
          xk[((size_t)b * a.k + f) * a.D + d] =
              real_bin(mk(hi_re[q][r] + lo_re[q][r], hi_im[q][r] + lo_im[q][r]), f, a.N);
      }
  }
}

// y[b, n, d] = bias[d] + sum_{f < k} (S.x w.x + S.y w.y),  w = w_N^{f n}: A = twiddles [row n][bin] (re and im),
// B = S [bin][channel] (re and im)
__global__ __launch_bounds__(256) void k_bx3_synth(const cf* __restrict__ sk, const float* __restrict__ bias,
                                                   float* __restrict__ y, DirectArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][2][BX_T * 32][BX_PITCH];   // [level][re|im][row][bin]
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][2][TD_CH][BX_PITCH];       // [level][re|im][channel][bin]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d0 = blockIdx.x * TD_CH, n0 = blockIdx.y * (BX_T * 32), b = blockIdx.z;
  const int R = a.rows_present();
  f32x16 hi[BX_T], lo[BX_T];
#pragma unroll
  for (int q = 0; q < BX_T; ++q) { hi[q] = f32x16{0}; lo[q] = f32x16{0}; }
  const cf* sb = sk + (size_t)b * a.k * a.D;
  TwStage ts[BX_T];                             // rows fixed, bins advance
  int ncnt[BX_T];
#pragma unroll
  for (int q = 0; q < BX_T; ++q) {
    ts[q].init(tid, n0 + 32 * q, a.N);
    ncnt[q] = max(0, min(32, R - n0 - 32 * q));
  }
  const int r31 = lane & 31, h = lane >> 5;
  const int xc = tid & 127, xh = tid >> 7;
  cf sr[16];
  cf twr[BX_T][4];
  auto fetch = [&](int f0) {
    const int fcnt = min(TD_K, a.k - f0);
    const bool cok = d0 + xc < a.D;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r0 = 4 * e + 2 * xh;
      sr[2 * e] = (cok && r0 < fcnt) ? sb[(size_t)(f0 + r0) * a.D + d0 + xc] : mk(0.f, 0.f);
      sr[2 * e + 1] = (cok && r0 + 1 < fcnt) ? sb[(size_t)(f0 + r0 + 1) * a.D + d0 + xc] : mk(0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < BX_T; ++q) ts[q].fetch(twr[q], a.tw, ncnt[q], fcnt);
  };
  if (a.k > 0) fetch(0);
  for (int f0 = 0; f0 < a.k; f0 += TD_K) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned wre[3], wim[3];
      split3_pair(sr[2 * e].x, sr[2 * e + 1].x, wre);
      split3_pair(sr[2 * e].y, sr[2 * e + 1].y, wim);
      const int r0 = 4 * e + 2 * xh;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        *reinterpret_cast<unsigned*>(&Bs[p][0][xc][r0]) = wre[p];
        *reinterpret_cast<unsigned*>(&Bs[p][1][xc][r0]) = wim[p];
      }
    }
#pragma unroll
    for (int q = 0; q < BX_T; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned wre[3], wim[3];
        split3_pair(twr[q][2 * e].x, twr[q][2 * e + 1].x, wre);
        split3_pair(twr[q][2 * e].y, twr[q][2 * e + 1].y, wim);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          *reinterpret_cast<unsigned*>(&As[p][0][32 * q + ts[q].i][ts[q].c0 + 2 * e]) = wre[p];
          *reinterpret_cast<unsigned*>(&As[p][1][32 * q + ts[q].i][ts[q].c0 + 2 * e]) = wim[p];
        }
      }
    __syncthreads();
    if (f0 + TD_K < a.k) fetch(f0 + TD_K);
#pragma unroll
    for (int s2 = 0; s2 < TD_K / 16; ++s2) {
      bf16x8 bre[3], bim[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        bre[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][0][wv * 32 + r31][16 * s2 + 8 * h]);
        bim[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][1][wv * 32 + r31][16 * s2 + 8 * h]);
      }
#pragma unroll
      for (int q = 0; q < BX_T; ++q) {
        bf16x8 are[3], aim[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          are[p] = *reinterpret_cast<const bf16x8*>(&As[p][0][32 * q + r31][16 * s2 + 8 * h]);
          aim[p] = *reinterpret_cast<const bf16x8*>(&As[p][1][32 * q + r31][16 * s2 + 8 * h]);
        }
        mfma6(are, bre, hi[q], lo[q]);          // Re(s conj(w)) = s.x w.x + s.y w.y
        mfma6(aim, bim, hi[q], lo[q]);
      }
    }
  }
  const int d = d0 + wv * 32 + r31;
  if (d < a.D) {
    const float bv = bias ? bias[d] : 0.f;
#pragma unroll
    for (int q = 0; q < BX_T; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < R) y[((size_t)b * R + n) * a.D + d] = hi[q][r] + lo[q][r] + bv;
      }
  }
}

// large problems only: below ~2^22 multiply-adds per batch row the literal kernels finish in a few
// microseconds anyway and keep their fp64 accumulation
static int g_tiled_dft = 3;        // 0 literal kernels only, 1 VALU tiles, 2 f32 matrix-core tiles, 3 bf16 x 3 ones (default)
void set_tiled_dft(int on) { g_tiled_dft = on; }
bool use_tiled(const DirectArgs& a) {
  return g_tiled_dft && a.f0 == 0 && a.fstep == 1 && a.rows == 0 && !a.accumulate && a.k >= 8 &&
         (double)a.rows_present() * a.k * a.D >= (double)(1 << 22);
}
hipError_t launch_tiled_spectrum(const float* x, cf* xk, const DirectArgs& a, hipStream_t s) {
  dim3 grid((a.D + TD_CH - 1) / TD_CH, (a.k + TD_BINS - 1) / TD_BINS, a.B);
  if (g_tiled_dft == 3) {
    grid.y = (a.k + BX_T * 32 - 1) / (BX_T * 32);
    hipLaunchKernelGGL(k_bx3_spectrum, grid, dim3(256), 0, s, x, xk, a);
  }
  else if (g_tiled_dft == 2) hipLaunchKernelGGL(k_mfma_spectrum, grid, dim3(256), 0, s, x, xk, a);
  else hipLaunchKernelGGL(k_tiled_spectrum, grid, dim3(256), 0, s, x, xk, a);
  return hipGetLastError();
}
hipError_t launch_tiled_synth(const cf* sk, const float* bias, float* y, const DirectArgs& a, hipStream_t s) {
  dim3 grid((a.D + TD_CH - 1) / TD_CH, (a.rows_present() + TD_BINS - 1) / TD_BINS, a.B);
  if (g_tiled_dft == 3) {
    grid.y = (a.rows_present() + BX_T * 32 - 1) / (BX_T * 32);
    hipLaunchKernelGGL(k_bx3_synth, grid, dim3(256), 0, s, sk, bias, y, a);
  }
  else if (g_tiled_dft == 2) hipLaunchKernelGGL(k_mfma_synth, grid, dim3(256), 0, s, sk, bias, y, a);
  else hipLaunchKernelGGL(k_tiled_synth, grid, dim3(256), 0, s, sk, bias, y, a);
  return hipGetLastError();
}

}  // namespace smx
