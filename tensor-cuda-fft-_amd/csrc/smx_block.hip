// smx_block.hip -- row kernels of the fused SpectralMLPBlock first half:
//     y = x + SpectralMixingLayer(LayerNorm(x))          reference fft_tensor/spectral_layers.py:185
//
// The LayerNorm itself is folded into the load of the decimated forward kernel and the residual add
// into its store (smx_decim.hip, k_fused_blk); what remains here is
//   * k_ln_stats    per-row (mean, rstd): the one extra read of x the fusion cannot avoid -- a row's
//                   statistics need all D channels, a workgroup of the transform owns 32;
//   * k_ln_bwd      LayerNorm backward + residual: grad_x = g + LN'(grad_h), in place over grad_h, and
//                   the per-block partial sums of grad_gamma / grad_beta;
//   * k_ln_colsum   fixed-order reduction of those partials (bitwise reproducible);
//   * k_ln_apply, k_add_rows   unfused fallback used with the split / direct transform paths.
// One wavefront per row, lane l holds elements (l + 64 c) VEC + [0, VEC), c < CH, in registers:
// every global access is a full-wave contiguous segment and a row is read exactly once.
#include <type_traits>

#include "smx_kernels.h"

namespace smx {

namespace {

constexpr int LN_WAVES = 4;                 // wavefronts (rows in flight) per block

// Sum over the 64 lanes, returned in every lane.  DPP row shifts + row broadcasts (six dependent
// VALU adds and one v_readlane) instead of six ds_bpermute round trips through the LDS crossbar.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                                ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0x111, 0xf, true>(v);      // row_shr:1
  v += dpp_f<0x112, 0xf, true>(v);      // row_shr:2
  v += dpp_f<0x114, 0xf, true>(v);      // row_shr:4
  v += dpp_f<0x118, 0xf, true>(v);      // row_shr:8  -> lane 15 of each row holds the row sum
  v += dpp_f<0x142, 0xa, false>(v);     // row_bcast:15 into rows 1 and 3
  v += dpp_f<0x143, 0xc, false>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int VEC> struct Vec;
template <> struct Vec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
  }
  __device__ __forceinline__ void load_cached(const float* p) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(p);
    v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    f32x4 w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct Vec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = __builtin_nontemporal_load(p); }
  __device__ __forceinline__ void load_cached(const float* p) { v[0] = *p; }
  __device__ __forceinline__ void store(float* p) const { __builtin_nontemporal_store(v[0], p); }
};

// stats[row] = (mean, 1/sqrt(var + eps)), biased variance, two passes over the registers
// (torch.nn.LayerNorm, used by the reference at spectral_layers.py:162,185).
template <int VEC, int CH>
__global__ __launch_bounds__(64 * LN_WAVES) void k_ln_stats(const float* __restrict__ x,
                                                             cf* __restrict__ stats, long long rows,
                                                             int D, float eps) {
  constexpr int R = CH <= 2 ? 2 : 1;          // rows in flight per wavefront (read-only kernel)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv_d = 1.f / (float)D;
  const long long step = (long long)gridDim.x * LN_WAVES;
  for (long long row0 = (long long)blockIdx.x * LN_WAVES + wv; row0 < rows; row0 += R * step) {
    Vec<VEC> xv[R][CH];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const long long row = row0 + q * step;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int e = (lane + 64 * c) * VEC;
        if (e < D && row < rows) xv[q][c].load(x + (size_t)row * D + e);
        else
#pragma unroll
          for (int i = 0; i < VEC; ++i) xv[q][c].v[i] = 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const long long row = row0 + q * step;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += xv[q][c].v[i];
      const float mean = wave_sum(s) * inv_d;
      float v2 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int e = (lane + 64 * c) * VEC;
        if (e < D)
#pragma unroll
          for (int i = 0; i < VEC; ++i) { const float dl = xv[q][c].v[i] - mean; v2 = fmaf(dl, dl, v2); }
      }
      const float var = wave_sum(v2) * inv_d;
      if (lane == 0 && row < rows) stats[row] = mk(mean, 1.f / sqrtf(var + eps));
    }
  }
}

// h = (x - mean) rstd gamma + beta          (unfused fallback)
template <int VEC, int CH>
__global__ __launch_bounds__(64 * LN_WAVES) void k_ln_apply(const float* __restrict__ x,
                                                             const cf* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ h, long long rows,
                                                             int D) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  Vec<VEC> gm[CH], bt[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (lane + 64 * c) * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { gm[c].v[i] = 1.f; bt[c].v[i] = 0.f; }
    if (e < D) {
      if (gamma) gm[c].load_cached(gamma + e);
      if (beta) bt[c].load_cached(beta + e);
    }
  }
  for (long long row = (long long)blockIdx.x * LN_WAVES + wv; row < rows;
       row += (long long)gridDim.x * LN_WAVES) {
    const cf st = stats[row];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (lane + 64 * c) * VEC;
      if (e < D) {
        Vec<VEC> xv;
        xv.load(x + (size_t)row * D + e);
#pragma unroll
        for (int i = 0; i < VEC; ++i) xv.v[i] = fmaf((xv.v[i] - st.x) * st.y, gm[c].v[i], bt[c].v[i]);
        xv.store(h + (size_t)row * D + e);
      }
    }
  }
}

// y += x      (unfused fallback: the residual)
__global__ void k_add_rows(float* __restrict__ y, const float* __restrict__ x, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += x[i];
}

// LayerNorm backward + the residual branch, in place over grad_h:
//   xh = (x - mean) rstd ; u = gamma grad_h ; grad_x = g + rstd (u - mean_d(u) - xh mean_d(u xh))
// and per block:  part[blk][0][d] = sum_rows grad_h xh   (grad_gamma),  part[blk][1][d] = sum_rows grad_h.
template <int VEC, int CH>
__global__ __launch_bounds__(64 * LN_WAVES) void k_ln_bwd(float* __restrict__ gh_dx,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ g,
                                                           const cf* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           float* __restrict__ part, long long rows,
                                                           int D) {
  __shared__ float red[LN_WAVES][2][64 * VEC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv_d = 1.f / (float)D;
  Vec<VEC> gm[CH], ag[CH], ab[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (lane + 64 * c) * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { gm[c].v[i] = 1.f; ag[c].v[i] = 0.f; ab[c].v[i] = 0.f; }
    if (e < D && gamma) gm[c].load_cached(gamma + e);
  }
  for (long long row = (long long)blockIdx.x * LN_WAVES + wv; row < rows;
       row += (long long)gridDim.x * LN_WAVES) {
    const size_t o = (size_t)row * D;
    Vec<VEC> hv[CH], xv[CH], gv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (lane + 64 * c) * VEC;
      if (e < D) {
        hv[c].load_cached(gh_dx + o + e);      // just written by the transform: may still be in cache
        xv[c].load(x + o + e);
        gv[c].load(g + o + e);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { hv[c].v[i] = 0.f; xv[c].v[i] = 0.f; gv[c].v[i] = 0.f; }
      }
    }
    const cf st = stats[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (lane + 64 * c) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xh = e < D ? (xv[c].v[i] - st.x) * st.y : 0.f;
        const float u = gm[c].v[i] * hv[c].v[i];
        xv[c].v[i] = xh;
        s1 += u;
        s2 = fmaf(u, xh, s2);
        ag[c].v[i] = fmaf(hv[c].v[i], xh, ag[c].v[i]);
        ab[c].v[i] += hv[c].v[i];
        hv[c].v[i] = u;
      }
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (lane + 64 * c) * VEC;
      if (e < D) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          gv[c].v[i] = fmaf(st.y, hv[c].v[i] - m1 - xv[c].v[i] * m2, gv[c].v[i]);
        gv[c].store(gh_dx + o + e);
      }
    }
  }
  // block partials: waves are added in index order
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      red[wv][0][lane * VEC + i] = ag[c].v[i];
      red[wv][1][lane * VEC + i] = ab[c].v[i];
    }
    __syncthreads();
    if (wv == 0) {
      const int e = (lane + 64 * c) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (e + i < D) {
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < LN_WAVES; ++w2) {
            a0 += red[w2][0][lane * VEC + i];
            a1 += red[w2][1][lane * VEC + i];
          }
          part[((size_t)blockIdx.x * 2 + 0) * D + e + i] = a0;
          part[((size_t)blockIdx.x * 2 + 1) * D + e + i] = a1;
        }
      }
    }
    __syncthreads();
  }
}

// out[which][d] = sum_blk part[blk][which][d]; 16 channels x 64 block groups per workgroup, groups
// added in index order (bitwise reproducible).  2 D/16 workgroups: 32 at D = 256.
__global__ __launch_bounds__(1024) void k_ln_colsum(const float* __restrict__ part, int nblk, int D,
                                                    float* __restrict__ g_gamma,
                                                    float* __restrict__ g_beta) {
  __shared__ float red[64][17];
  const int tx = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int d = blockIdx.x * 16 + tx, which = blockIdx.y;
  float acc = 0.f;
  if (d < D) {
    int b = grp;
    for (; b + 64 * 32 <= nblk; b += 64 * 32) {      // all of a thread's loads in one batch
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = part[((size_t)(b + 64 * u) * 2 + which) * D + d];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += v[u];
    }
    for (; b + 64 * 8 <= nblk; b += 64 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(b + 64 * u) * 2 + which) * D + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblk; b += 64) acc += part[((size_t)b * 2 + which) * D + d];
  }
  red[grp][tx] = acc;
  __syncthreads();
  if (grp == 0 && d < D) {
    float s = 0.f;
#pragma unroll 8
    for (int g2 = 0; g2 < 64; ++g2) s += red[g2][tx];
    float* out = which == 0 ? g_gamma : g_beta;
    if (out) out[d] = s;
  }
}

// out = mask * scale * in over (B, row_elems): the same mask function as the fused epilogues
__global__ void k_dropout_rows(const float* __restrict__ in, float* __restrict__ out, int B,
                               long long row_elems, unsigned thr, float scale,
                               const unsigned long long* __restrict__ rng) {
  const unsigned long long s0 = rng[0], s1 = rng[1];
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const unsigned key = drop_row_key(s0, s1, b);
    const size_t base = (size_t)b * (size_t)row_elems;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < row_elems;
         e += (long long)gridDim.x * blockDim.x) {
      const unsigned h = drop_hash((unsigned)((unsigned long long)e >> 1), key);
      const unsigned u = (e & 1) ? (h >> 16) : (h & 0xffffu);
      out[base + e] = u >= thr ? in[base + e] * scale : 0.f;
    }
  }
}

__global__ void k_rng_next(unsigned long long* state, unsigned long long* saved) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long a = state[0], c = state[1];
    saved[0] = a; saved[1] = c;
    state[1] = c + 1;
  }
}

int ln_blocks(long long rows) { return ln_num_blocks(rows); }

// dispatch on (VEC, CH): smallest register tile that covers D
template <typename Fn4, typename Fn1>
bool ln_dispatch(int D, Fn4 f4, Fn1 f1) {
  if (D % 4 == 0) {
    const int ch = (D / 4 + 63) / 64;
    if (ch <= 1) f4(std::integral_constant<int, 1>());
    else if (ch <= 2) f4(std::integral_constant<int, 2>());
    else if (ch <= 4) f4(std::integral_constant<int, 4>());
    else if (ch <= 8) f4(std::integral_constant<int, 8>());
    else if (ch <= 16) f4(std::integral_constant<int, 16>());
    else return false;
    return true;
  }
  const int ch = (D + 63) / 64;
  if (ch <= 1) f1(std::integral_constant<int, 1>());
  else if (ch <= 4) f1(std::integral_constant<int, 4>());
  else if (ch <= 16) f1(std::integral_constant<int, 16>());
  else return false;
  return true;
}

}  // namespace

int ln_num_blocks(long long rows) {
  long long b = (rows + LN_WAVES - 1) / LN_WAVES;
  return (int)(b < 1 ? 1 : b > LN_MAX_BLOCKS ? LN_MAX_BLOCKS : b);
}

bool ln_supported(int D) { return D >= 1 && (D % 4 == 0 ? D <= LN_MAX_D : D <= LN_MAX_D_ODD); }

hipError_t launch_dropout_rows(const float* in, float* out, int B, long long row_elems, unsigned thr,
                               float scale, const unsigned long long* rng, hipStream_t s) {
  if (B <= 0 || row_elems <= 0) return hipSuccess;
  const long long nb = (row_elems + 255) / 256;
  dim3 grid((unsigned)(nb < 4096 ? nb : 4096), (unsigned)(B < 64 ? B : 64));
  hipLaunchKernelGGL(k_dropout_rows, grid, dim3(256), 0, s, in, out, B, row_elems, thr, scale, rng);
  return hipGetLastError();
}

hipError_t launch_rng_next(unsigned long long* state, unsigned long long* saved, hipStream_t s) {
  hipLaunchKernelGGL(k_rng_next, dim3(1), dim3(64), 0, s, state, saved);
  return hipGetLastError();
}

hipError_t launch_ln_stats(const float* x, cf* stats, long long rows, int D, float eps,
                           hipStream_t s) {
  const dim3 grid(ln_blocks(rows)), block(64 * LN_WAVES);
  ln_dispatch(
      D,
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_stats<4, decltype(ch)::value>), grid, block, 0, s, x, stats, rows,
                           D, eps);
      },
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_stats<1, decltype(ch)::value>), grid, block, 0, s, x, stats, rows,
                           D, eps);
      });
  return hipGetLastError();
}

hipError_t launch_ln_apply(const float* x, const cf* stats, const float* gamma, const float* beta,
                           float* h, long long rows, int D, hipStream_t s) {
  const dim3 grid(ln_blocks(rows)), block(64 * LN_WAVES);
  ln_dispatch(
      D,
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_apply<4, decltype(ch)::value>), grid, block, 0, s, x, stats, gamma,
                           beta, h, rows, D);
      },
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_apply<1, decltype(ch)::value>), grid, block, 0, s, x, stats, gamma,
                           beta, h, rows, D);
      });
  return hipGetLastError();
}

hipError_t launch_add_rows(float* y, const float* x, size_t total, hipStream_t s) {
  if (total == 0) return hipSuccess;
  const size_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(k_add_rows, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, s, y, x,
                     total);
  return hipGetLastError();
}

hipError_t launch_ln_bwd(float* gh_dx, const float* x, const float* g, const cf* stats,
                         const float* gamma, float* part, float* g_gamma, float* g_beta,
                         long long rows, int D, hipStream_t s) {
  const int nblk = ln_blocks(rows);
  const dim3 grid(nblk), block(64 * LN_WAVES);
  ln_dispatch(
      D,
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_bwd<4, decltype(ch)::value>), grid, block, 0, s, gh_dx, x, g, stats,
                           gamma, part, rows, D);
      },
      [&](auto ch) {
        hipLaunchKernelGGL((k_ln_bwd<1, decltype(ch)::value>), grid, block, 0, s, gh_dx, x, g, stats,
                           gamma, part, rows, D);
      });
  if (hipError_t e = hipGetLastError()) return e;
  if (g_gamma || g_beta) {
    hipLaunchKernelGGL(k_ln_colsum, dim3((D + 15) / 16, 2), dim3(1024), 0, s, part, nblk, D, g_gamma,
                       g_beta);
  }
  return hipGetLastError();
}

}  // namespace smx
