// smx_fourstep.hip -- four-step path for full spectra of long transforms (N = 256 L, 5 <= L <= 16 or 32):
// (A) tile spectra -> workspace, (F) per-thread column pairs, (B) inverse tiles.  Arithmetic and the
// description live at the end of smx_core.h; a translation unit of its own so that the L-templated column
// kernels compile in parallel with smx_decim.hip.
//
// Replaces for those lengths: reference fft_lm/train_fixed_full.py:515-519, :553 (rfft / irfft of the
// zero-padded sequence), fft_tensor/spectral_enhancements.py:147, :164, complex_rope.py:207, :216,
// frequency_ops.py:201.
#include "smx_launch.h"
#include "smx_fs_big.h"

namespace smx {

// ---- four-step path: see the end of smx_core.h ------------------------------------------------------
// (A) tile spectra of a chunk of residues -> workspace.  Same streaming loop as k_split_a.
template <bool PAD>
__global__ __launch_bounds__(TPB, 2) void k_fs_a(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int rbeg = c * a.lc, cnt = min(a.lc, g.L - rbeg);
  if (cnt <= 0) return;
  const int rend = rbeg + cnt;
  const float* xb = a.in + (size_t)b * g.R * g.D + (valid ? d : g.D - 2);
  cf* dst0 = a.ws_f + (size_t)wg * g.L * EX + tid;
  TState<1> st;
  cf nx[16];
  int r = rbeg + w.rot % cnt;
  load_tile<PAD>(xb, g, t, r, nx);
  cf cn = a.tw[(size_t)t * g.L + r];
  for (int i = 0; i < cnt; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) st.v[u] = nx[u];
    const cf cc = cn;
    int rn = r + 1;
    if (rn == rend) rn = rbeg;
    if (i + 1 < cnt) {
      load_part_tile<0, 8, PAD>(xb, g, t, rn, nx);
      cn = a.tw[(size_t)t * g.L + rn];
    }
    cf* E = lds + (i & 1) * EX;
    fwd_phase1<1>(st, cc, E, t, j);
    __syncthreads();
    if (i + 1 < cnt) load_part_tile<8, 8, PAD>(xb, g, t, rn, nx);
    fwd_phase2_out(E, a.bt + (size_t)r * BT_STRIDE, t, j, dst0 + (size_t)r * EX);
    r = rn;
  }
}

// (F) column pairs {fu, 256 - fu}: L-point transforms across the residues, unpack, filter, repack, back.
// 129 column units per (batch row, d-tile): grid.y = 9 blocks of 16 units x 16 channel pairs.
template <int L, int MODE>
__global__ __launch_bounds__(TPB) void k_fs_f(const DecimArgs a) {
  __shared__ cf red[MODE == 1 ? TPB : 1];
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, u = blockIdx.y * 16 + (tid >> 4);
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  cf gs = mk(0.f, 0.f);
  if (u <= 128) fs_columns<L, MODE>(a.ws_f + (size_t)wg * L * EX, g, a.fa, a.tw, b, d, d < g.D, u, j,
                                    MODE == 1 ? &gs : nullptr);
  if constexpr (MODE == 1) {
    if (a.fa.gsc_part != nullptr) {      // row-scale gradient: sum over the block's 16 column units, fixed order
      red[tid] = gs;
      __syncthreads();
      if (tid < 16) {
        cf acc = mk(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = cadd(acc, red[i * 16 + tid]);
        a.fa.gsc_part[((size_t)wg * 9 + blockIdx.y) * 16 + tid] = acc;
      }
    }
  }
}

// (F) of smx_irfft_ex: columns from a given one-sided spectrum (fs_synth_columns), same grid as k_fs_f
template <int L>
__global__ __launch_bounds__(TPB) void k_fs_synth(const DecimArgs a) {
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, u = blockIdx.y * 16 + (tid >> 4);
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  if (u <= 128) fs_synth_columns<L>(a.ws_f + (size_t)wg * L * EX, g, a.fa, a.tw, b, d, d < g.D, u, j);
}

// Backward with the slab summed over batch groups (option "fs_bgroups", off by default -- measured slower):
// blockIdx.x = d-tile + ndt * batch group; one thread walks the group's batch rows and keeps the sums of its
// slab rows in registers (4 L floats).  A kernel of its own so that k_fs_f<L, 1> stays lean.
template <int L>
__global__ __launch_bounds__(TPB) void k_fs_f_grouped(const DecimArgs a) {
  __shared__ cf red[TPB];
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, u = blockIdx.y * 16 + (tid >> 4);
  const int ndt = (g.D + DT - 1) / DT;
  const bool want_gs = a.fa.gsc_part != nullptr;
  const int dt = blockIdx.x % ndt, grp = blockIdx.x / ndt, d = dt * DT + 2 * j;
  const int per = (g.B + a.fs_bgroups - 1) / a.fs_bgroups;
  const int b0 = grp * per, b1 = min(g.B, b0 + per);
  cf pacc[L][2];
#pragma unroll
  for (int i = 0; i < L; ++i) { pacc[i][0] = mk(0.f, 0.f); pacc[i][1] = mk(0.f, 0.f); }
  cf gbacc = mk(0.f, 0.f);
  for (int b = b0; b < b1; ++b) {
    const int wg = b * ndt + dt;
    cf gs = mk(0.f, 0.f);
    if (u <= 128) fs_columns<L, 1>(a.ws_f + (size_t)wg * L * EX, g, a.fa, a.tw, b, d, d < g.D, u, j, &gs, pacc,
                                   &gbacc);
    if (want_gs) {
      __syncthreads();
      red[tid] = gs;
      __syncthreads();
      if (tid < 16) {
        cf acc = mk(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = cadd(acc, red[i * 16 + tid]);
        a.fa.gsc_part[((size_t)wg * 9 + blockIdx.y) * 16 + tid] = acc;
      }
    }
  }
  if (u <= 128 && b1 > b0) fs_store_slab<L>(pacc, gbacc, g, a.fa, grp, d, d < g.D, u);
}

// gsc[b, d] = sum over the 9 column-unit blocks of the four-step filter launch
__global__ void k_fs_gsc(const cf* __restrict__ part, float* __restrict__ gsc, int B, int D, int ny) {
  const int ndt = (D + DT - 1) / DT;
  const long long total = (long long)B * ndt * 16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int jj = (int)(i % 16);
    const long long wg = i / 16;
    const int b = (int)(wg / ndt), d = (int)(wg % ndt) * DT + 2 * jj;
    if (d >= D) continue;
    cf acc = mk(0.f, 0.f);
    for (int ub = 0; ub < ny; ++ub) acc = cadd(acc, part[((size_t)wg * ny + ub) * 16 + jj]);
    gsc[(size_t)b * D + d] = acc.x;
    gsc[(size_t)b * D + d + 1] = acc.y;
  }
}

// (B) inverse tiles of a chunk of residues from the filtered workspace.
template <bool PAD>
__global__ __launch_bounds__(TPB, 2) void k_fs_b(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int rbeg = c * a.lc, cnt = min(a.lc, g.L - rbeg);
  if (cnt <= 0) return;
  const cf* src0 = a.ws_f + (size_t)wg * g.L * EX + tid;
  float* yb = a.out + (size_t)b * g.R * g.D + d;
  float osa = 1.f, osb = 1.f;
  if (a.out_scale && valid) { osa = a.out_scale[(size_t)b * g.D + d]; osb = a.out_scale[(size_t)b * g.D + d + 1]; }
  TState<1> st;
  cf nx[16];
  int r = rbeg + w.rot % cnt;
#pragma unroll
  for (int s = 0; s < 16; ++s) nx[s] = src0[(size_t)r * EX + s * TPB];
  // (the tile's twiddle is requested with the tile, one iteration ahead: a vector load issued at the top of an
  //  iteration sits behind the previous tile's 16 stores in the in-order vmcnt queue -- DESIGN 4.1)
  cf cn = a.tw[(size_t)t * g.L + r];
  for (int i = 0; i < cnt; ++i) {
    cf v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = nx[s];
    const cf cc = cn;
    int rn = r + 1;
    if (rn == rbeg + cnt) rn = rbeg;
    if (i + 1 < cnt) {
#pragma unroll
      for (int s = 0; s < 16; ++s) nx[s] = src0[(size_t)rn * EX + s * TPB];
      cn = a.tw[(size_t)t * g.L + rn];
    }
    cf* E = lds + (i & 1) * EX;
    inv_phase1_in(v, a.bt + (size_t)r * BT_STRIDE, E, t, j);
    __syncthreads();
    inv_phase2<1>(st, cc, E, t, j);
    if (a.out_scale) {
#pragma unroll
      for (int u = 0; u < 16; ++u) st.v[u] = mk(st.v[u].x * osa, st.v[u].y * osb);
    }
    store_tile<PAD>(yb, g, t, r, valid, st.v);
    r = rn;
  }
}

// ---- rank-one filter (fs_conv_columns in smx_core.h) ---------------------------------------------------
template <int L, int DIR>
__global__ __launch_bounds__(TPB) void k_fs_conv(const DecimArgs a) {
  __shared__ cf red[TPB];
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, ul = tid >> 4, u = blockIdx.y * 16 + ul;
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  cf pp[DIR ? L : 1], pm[DIR ? L : 1];
  cf rr = mk(0.f, 0.f);
  if (u <= 128)
    fs_conv_columns<L, DIR>(a.conv_src + (size_t)wg * L * EX, a.ws_f + (size_t)wg * L * EX,
                            DIR ? a.ca.xs + (size_t)wg * L * EX : nullptr, g, a.ca, a.tw, b, d, d < g.D, u, j, pp,
                            pm, &rr);
  if constexpr (DIR == 1) {
    // P: sum over the 16 channel pairs of the unit (xor butterfly inside each group of 16 lanes)
    if (u <= 128) {
      const int fum = (256 - u) & 255;
      const bool one_col = (u == 0 || u == 128);
#pragma unroll
      for (int f2 = 0; f2 < L; ++f2) {
        cf vp = pp[f2], vm = pm[f2];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
          vp.x += __shfl_xor(vp.x, m, 16); vp.y += __shfl_xor(vp.y, m, 16);
          vm.x += __shfl_xor(vm.x, m, 16); vm.y += __shfl_xor(vm.y, m, 16);
        }
        if (j == 0) {
          a.ca.p_part[(size_t)wg * g.N + u + 256 * f2] = vp;
          if (!one_col) a.ca.p_part[(size_t)wg * g.N + fum + 256 * f2] = vm;
        }
      }
    }
    // (R1, R2): sum over the block's 16 column units
    red[tid] = rr;
    __syncthreads();
    if (tid < 16) {
      cf acc = mk(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = cadd(acc, red[i * 16 + tid]);
      a.ca.r_part[((size_t)wg * 9 + blockIdx.y) * 16 + tid] = acc;
    }
  }
}

// the same on the two-level columns (L = 16 L2; L = 32 with L2 = 2): grid.y = conv_column_blocks(L)
template <int L2, int DIR>
__global__ __launch_bounds__(TPB) void k_fs_conv_big(const DecimArgs a) {
  __shared__ cf X[EX];
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, sub = tid >> 4, t2 = sub % L2, ul = sub / L2;
  const int u = blockIdx.y * (16 / L2) + ul;
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  const bool valid = d < g.D, act = u <= 128;
  const size_t wo = (size_t)wg * (16 * L2) * EX;
  BigState sg, sx;
  if constexpr (DIR == 1) {
    big_forward<L2>(sx, a.ca.xs + wo, a.tw, X, act, u, ul, t2, j);
    __syncthreads();
  }
  big_forward<L2>(sg, a.conv_src + wo, a.tw, X, act, u, ul, t2, j);
  cf rr = mk(0.f, 0.f);
  if constexpr (DIR == 1) {
    if (act) {
      // P: sum over the 16 channel pairs of the unit (xor butterfly inside each group of 16 lanes)
      fsb_conv_sums<L2>(sg, sx, g, a.ca, b, d, valid, u, t2, rr, [&](int fp, int fm, cf vp, cf vm, bool one_col) {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
          vp.x += __shfl_xor(vp.x, m, 16); vp.y += __shfl_xor(vp.y, m, 16);
          vm.x += __shfl_xor(vm.x, m, 16); vm.y += __shfl_xor(vm.y, m, 16);
        }
        if (j == 0) {
          a.ca.p_part[(size_t)wg * g.N + fp] = vp;
          if (!one_col) a.ca.p_part[(size_t)wg * g.N + fm] = vm;
        }
      });
    }
  }
  if (act) fsb_conv_scale<L2, DIR>(sg, g, a.ca, valid, u, t2);
  big_inverse<L2>(sg, a.ws_f + wo, a.tw, X, act, u, ul, t2, j);
  if constexpr (DIR == 1) {
    __syncthreads();
    X[tid] = rr;
    __syncthreads();
    if (tid < 16) {
      cf acc = mk(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = cadd(acc, X[i * 16 + tid]);
      a.ca.r_part[((size_t)wg * gridDim.y + blockIdx.y) * 16 + tid] = acc;
    }
  }
}

template <int L>
static void launch_fs_conv_t(const DecimArgs& a, int dir, dim3 grid, hipStream_t s) {
  if (dir == 0) hipLaunchKernelGGL((k_fs_conv<L, 0>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fs_conv<L, 1>), grid, dim3(TPB), 0, s, a);
}
template <int L2>
static void launch_fs_conv_big_t(const DecimArgs& a, int dir, hipStream_t s) {
  const dim3 grid(n_wg(a), (129 + 16 / L2 - 1) / (16 / L2));
  if (dir == 0) hipLaunchKernelGGL((k_fs_conv_big<L2, 0>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fs_conv_big<L2, 1>), grid, dim3(TPB), 0, s, a);
}
int conv_column_blocks(int L) { return L >= 32 ? (129 + 16 / (L / 16) - 1) / (16 / (L / 16)) : 9; }

// Both reductions behind a backward column launch in ONE launch (round 3: three launches before -- two stages of the P
// sum and the (R1, R2) sum -- each about 5 us of latency):
//   blocks [0, nbh): dL/dH[f] = c_f Q[f] / N, Q[f] = (P[f] + conj P[N - f]) / 2, P = sum over the nwg workgroup rows of
//     the partials: eight bins per block, 32 row groups per bin (thread g sums rows g, g + 32, ... in order), then the 32
//     group sums in order -- a fixed summation order, whatever the launch geometry;
//   blocks [nbh, ...): grad_s[b, d], grad_s[b, d+1] = rscale (R1 +/- R2) from the ny partial rows of each workgroup.
__global__ __launch_bounds__(256) void k_conv_grads(const cf* __restrict__ ppart, int nwg, int N,
                                                   float* __restrict__ gh_re, float* __restrict__ gh_im, int nbh,
                                                   const cf* __restrict__ rpart, float* __restrict__ gs, int B, int D,
                                                   float rscale, int ny, int nj) {
  __shared__ cf ra[32][8], rb[32][8];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nbh) {
    const int fl = tid & 7, g = tid >> 3, f = blockIdx.x * 8 + fl;
    const bool live = f <= N / 2;
    const int fc = live ? f : 0, fn = (N - fc) % N;
    cf a = mk(0.f, 0.f), b = mk(0.f, 0.f);
    int w = g;
    for (; w + 96 < nwg; w += 128) {                  // four rows of each in flight (same order of additions)
      cf va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { va[u] = ppart[(size_t)(w + 32 * u) * N + fc]; vb[u] = ppart[(size_t)(w + 32 * u) * N + fn]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a = cadd(a, va[u]); b = cadd(b, vb[u]); }
    }
    for (; w < nwg; w += 32) { a = cadd(a, ppart[(size_t)w * N + fc]); b = cadd(b, ppart[(size_t)w * N + fn]); }
    ra[g][fl] = a; rb[g][fl] = b;
    __syncthreads();
    if (g == 0 && live) {
      cf sa = ra[0][fl], sb = rb[0][fl];
#pragma unroll
      for (int i = 1; i < 32; ++i) { sa = cadd(sa, ra[i][fl]); sb = cadd(sb, rb[i][fl]); }
      const bool edge = f == 0 || 2 * f == N;                     // imaginary parts of DC / Nyquist never reach y
      const float sc = (edge ? 0.5f : 1.0f) / (float)N;           // c_f / 2 / N
      gh_re[f] = (sa.x + sb.x) * sc;
      gh_im[f] = edge ? 0.f : (sa.y - sb.y) * sc;
    }
    return;
  }
  if (gs == nullptr) return;
  const int dtj = 2 * nj, ndt = (D + dtj - 1) / dtj;
  const long long total = (long long)B * ndt * nj;
  const long long i = (long long)(blockIdx.x - nbh) * 256 + tid;
  if (i >= total) return;
  const int jj = (int)(i % nj);
  const long long wg = i / nj;
  const int bb = (int)(wg / ndt), d = (int)(wg % ndt) * dtj + 2 * jj;
  if (d >= D) return;
  cf acc = mk(0.f, 0.f);
  for (int ub = 0; ub < ny; ++ub) acc = cadd(acc, rpart[((size_t)wg * ny + ub) * nj + jj]);
  gs[(size_t)bb * D + d] = (acc.x + acc.y) * rscale;
  gs[(size_t)bb * D + d + 1] = (acc.x - acc.y) * rscale;
}

// the sums behind a backward column launch: P partials -> dL/dH (gh_re, gh_im: N/2 + 1 each, or null),
// (R1, R2) partials ([workgroup][ny][nj]) -> grad_scale (B, D) = rscale (R1 +/- R2) (or null)
hipError_t launch_conv_reduce(const DecimArgs& a, float* gh_re, float* gh_im, float* grad_scale, int ny,
                              float rscale, hipStream_t s, int nwg, int nj) {
  if (nwg <= 0) nwg = n_wg(a);
  const bool want_h = gh_re && gh_im;
  const int nbh = want_h ? (a.g.N / 2 + 1 + 7) / 8 : 0;
  const long long total = grad_scale ? (long long)nwg * nj : 0;
  const int nbr = (int)((total + 255) / 256);
  if (nbh + nbr == 0) return hipSuccess;
  hipLaunchKernelGGL(k_conv_grads, dim3(nbh + nbr), dim3(256), 0, s, a.ca.p_part, nwg, a.g.N, gh_re, gh_im, nbh,
                     a.ca.r_part, grad_scale, a.g.B, a.g.D, rscale, ny, nj);
  return hipGetLastError();
}

hipError_t launch_fs_conv(const DecimArgs& a, int dir, float* gh_re, float* gh_im, float* grad_scale,
                          hipStream_t s) {
  const dim3 grid(n_wg(a), 9);
  switch (a.g.L) {
    case 32: launch_fs_conv_big_t<2>(a, dir, s); break;
    case 64: launch_fs_conv_big_t<4>(a, dir, s); break;
    case 128: launch_fs_conv_big_t<8>(a, dir, s); break;
    case 256: launch_fs_conv_big_t<16>(a, dir, s); break;
#define SMX_FS_CASE(LL) case LL: launch_fs_conv_t<LL>(a, dir, grid, s); break;
    SMX_FS_CASE(2) SMX_FS_CASE(4) SMX_FS_CASE(8) SMX_FS_CASE(16)     // (L = 32 was measured: 512 registers +
#undef SMX_FS_CASE                                                     //  116 spills in backward, no faster than k_fs_f)
    default: return hipErrorInvalidValue;
  }
  if (dir == 1) return launch_conv_reduce(a, gh_re, gh_im, grad_scale, conv_column_blocks(a.g.L), 0.5f * a.g.inv_n, s);
  return hipGetLastError();
}

hipError_t launch_fs_a(const DecimArgs& a, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    if (r.g.R < r.g.N) hipLaunchKernelGGL((k_fs_a<true>), grid, dim3(TPB), 0, s, r);
    else hipLaunchKernelGGL((k_fs_a<false>), grid, dim3(TPB), 0, s, r);
  });
}
hipError_t launch_fs_b(const DecimArgs& a, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    if (r.g.R < r.g.N) hipLaunchKernelGGL((k_fs_b<true>), grid, dim3(TPB), 0, s, r);
    else hipLaunchKernelGGL((k_fs_b<false>), grid, dim3(TPB), 0, s, r);
  });
}
template <int L>
static void launch_fs_f_t(const DecimArgs& a, int mode, dim3 grid, hipStream_t s) {
  if (mode == 0) hipLaunchKernelGGL((k_fs_f<L, 0>), grid, dim3(TPB), 0, s, a);
  else if (mode == 1 && a.fs_bgroups > 0) {
    if constexpr (L >= 5 && L <= 16) hipLaunchKernelGGL((k_fs_f_grouped<L>), grid, dim3(TPB), 0, s, a);
  } else if (mode == 1) hipLaunchKernelGGL((k_fs_f<L, 1>), grid, dim3(TPB), 0, s, a);
  else if (mode == 2) hipLaunchKernelGGL((k_fs_f<L, 2>), grid, dim3(TPB), 0, s, a);
  else if (mode == 4) hipLaunchKernelGGL((k_fs_synth<L>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fs_f<L, 3>), grid, dim3(TPB), 0, s, a);
}
// L = L1 L2 on the two-level columns: L2 threads per column pair (a divisor of 16), first-level length L1 <= 16.
// 64 / 128 / 256 = 16 x 4 / 8 / 16 (round 2); round 3: L1 = 9 ... 15 with the smallest L2 in {4, 8, 16} that fits.
bool fs_two_level(int L, int* L1, int* L2) {
  for (int l2 = 4; l2 <= 16; l2 *= 2)
    if (L % l2 == 0 && L / l2 >= 9 && L / l2 <= 16) { *L1 = L / l2; *L2 = l2; return true; }
  return false;
}
int fs_column_blocks(int L) {
  int l1, l2;
  if (L >= 33 && fs_two_level(L, &l1, &l2)) return (129 + 16 / l2 - 1) / (16 / l2);
  return 9;
}

hipError_t launch_fs_f(const DecimArgs& a, int mode, hipStream_t s) {
  const int ndt = (a.g.D + DT - 1) / DT;
  int l1 = 0, l2 = 0;
  if (a.g.L >= 33 && fs_two_level(a.g.L, &l1, &l2)) {
    if (l1 == 16) {
      if (l2 == 4) launch_fs_big_t<4>(a, mode, s);
      else if (l2 == 8) launch_fs_big_t<8>(a, mode, s);
      else launch_fs_big_t<16>(a, mode, s);
    } else {
      if (hipError_t e = launch_fs_big_general(a, mode, l1, l2, s)) return e;
    }
    if (mode == 1 && a.fa.gsc_part && a.fa.gsc) {
      const long long total = (long long)n_wg(a) * 16;
      hipLaunchKernelGGL(k_fs_gsc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.fa.gsc_part, a.fa.gsc,
                         a.g.B, a.g.D, fs_column_blocks(a.g.L));
    }
    return hipGetLastError();
  }
  const bool grouped = mode == 1 && a.fs_bgroups > 0 && a.g.L >= 5 && a.g.L <= 16;
  const dim3 grid(grouped ? ndt * a.fs_bgroups : n_wg(a), 9);
  switch (a.g.L) {
#define SMX_FS_CASE(LL) case LL: launch_fs_f_t<LL>(a, mode, grid, s); break;
    SMX_FS_CASE(2) SMX_FS_CASE(4)          // complex sequence FFT only (filter plans start at L = 5)
    SMX_FS_CASE(5) SMX_FS_CASE(6) SMX_FS_CASE(7) SMX_FS_CASE(8) SMX_FS_CASE(9) SMX_FS_CASE(10) SMX_FS_CASE(11)
    SMX_FS_CASE(12) SMX_FS_CASE(13) SMX_FS_CASE(14) SMX_FS_CASE(15) SMX_FS_CASE(16) SMX_FS_CASE(32)
    SMX_FS_CASE(18) SMX_FS_CASE(20) SMX_FS_CASE(22) SMX_FS_CASE(24) SMX_FS_CASE(26) SMX_FS_CASE(28) SMX_FS_CASE(30)
    // odd tile counts 17 ... 31 (round 3): the L x L product in one thread's registers, as for the odd L <= 15
    SMX_FS_CASE(17) SMX_FS_CASE(19) SMX_FS_CASE(21) SMX_FS_CASE(23) SMX_FS_CASE(25) SMX_FS_CASE(27) SMX_FS_CASE(29)
    SMX_FS_CASE(31)
#undef SMX_FS_CASE
    default: return hipErrorInvalidValue;
  }
  if (mode == 1 && a.fa.gsc_part && a.fa.gsc) {
    const long long total = (long long)n_wg(a) * 16;
    hipLaunchKernelGGL(k_fs_gsc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.fa.gsc_part, a.fa.gsc,
                       a.g.B, a.g.D, 9);
  }
  return hipGetLastError();
}

}  // namespace smx
