// smx_fs_big.h -- the two-level column kernel of the four-step path (L = L1 L2 residues, L2 threads per column
// pair), shared by smx_fourstep.hip (L1 = 16: L = 64, 128, 256, and the rank-one filter's columns) and
// smx_fourstep2.hip (round 3: any first-level length L1 = 9 ... 15, L = 36 ... 240).  Arithmetic: smx_core.h.
#pragma once
#include "smx_launch.h"

namespace smx {

// ---- two-level columns (smx_core.h): the exchanges of one block, barriers included --------------------------
// forward: residues of both columns -> registers hold the bins (zp) and their mirror partners (zm)
template <int L2, int L1 = 16>
__device__ __forceinline__ void big_forward(BigState& st, const cf* __restrict__ src, const cf* __restrict__ tw,
                                            cf* X, bool act, int u, int ul, int t2, int j) {
  if (act) {
    fsb_load<L2, L1>(st, src, u, t2, j);
    fsb_pub<L2, L1>(st.zp, tw, X, ul, t2, j);
  }
  __syncthreads();
  if (act) fsb_gather<L2, false, L1>(st.zp, X, tw, u, ul, t2, j);
  __syncthreads();
  if (act) fsb_pub<L2, L1>(st.zm, tw, X, ul, t2, j);
  __syncthreads();
  if (act) fsb_gather<L2, true, L1>(st.zm, X, tw, u, ul, t2, j);
}
// inverse: bins -> residues, stored to dst (a one-column unit has nothing to store for the mirror column)
template <int L2, int L1 = 16>
__device__ __forceinline__ void big_inverse(BigState& st, cf* __restrict__ dst, const cf* __restrict__ tw, cf* X,
                                            bool act, int u, int ul, int t2, int j) {
  const bool two = act && u != 0 && u != 128;
  __syncthreads();
  if (act) fsb_unpub<L2, false, L1>(st.zp, X, tw, ul, t2, j);
  __syncthreads();
  if (act) fsb_ungather<L2, L1>(st.zp, dst, X, tw, u, ul, t2, j);
  __syncthreads();
  if (two) fsb_unpub<L2, true, L1>(st.zm, X, tw, ul, t2, j);
  __syncthreads();
  if (two) fsb_ungather<L2, L1>(st.zm, dst, X, tw, (256 - u) & 255, ul, t2, j);
}

// (F) for L = L1 L2 residues (L1 = 16: N = 16384 / 32768 / 65536): L2 threads per column pair.
// grid.y = ceil(129 / (16 / L2)) blocks of 16 / L2 column units.
// MODE 0 / 1 / 2 as k_fs_f, 3 = packed bins out (complex sequence FFT), 4 = synthesis from a given spectrum.
template <int L2, int MODE, int L1 = 16>
__global__ __launch_bounds__(TPB) void k_fs_big(const DecimArgs a) {
  __shared__ cf X[EX];                                       // 32 KiB
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, sub = tid >> 4, t2 = sub % L2, ul = sub / L2;
  const int u = blockIdx.y * (16 / L2) + ul;
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  const bool valid = d < g.D, act = u <= 128;
  cf* wsb = a.ws_f + (size_t)wg * (L1 * L2) * EX;
  BigState st;
  cf gs = mk(0.f, 0.f);
  if constexpr (MODE == 4) {
    if (act) fsb_synth<L2, L1>(st, g, a.fa, b, d, valid, u, t2);
  } else {
    big_forward<L2, L1>(st, wsb, a.tw, X, act, u, ul, t2, j);
    if (act) fsb_pairs<L2, MODE, L1>(st, g, a.fa, b, d, valid, u, t2, MODE == 1 ? &gs : nullptr);
  }
  if constexpr (MODE == 0 || MODE == 1 || MODE == 4) big_inverse<L2, L1>(st, wsb, a.tw, X, act, u, ul, t2, j);
  if constexpr (MODE == 1) {
    if (a.fa.gsc_part != nullptr) {      // row-scale gradient: sum over the block's 16 (unit, t2) threads per j
      __syncthreads();
      X[tid] = act ? gs : mk(0.f, 0.f);
      __syncthreads();
      if (tid < 16) {
        cf acc = mk(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = cadd(acc, X[i * 16 + tid]);
        a.fa.gsc_part[((size_t)wg * gridDim.y + blockIdx.y) * 16 + tid] = acc;
      }
    }
  }
}

template <int L2, int L1 = 16>
static void launch_fs_big_t(const DecimArgs& a, int mode, hipStream_t s) {
  const dim3 grid(n_wg(a), (129 + 16 / L2 - 1) / (16 / L2));
  if (mode == 0) hipLaunchKernelGGL((k_fs_big<L2, 0, L1>), grid, dim3(TPB), 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL((k_fs_big<L2, 1, L1>), grid, dim3(TPB), 0, s, a);
  else if (mode == 2) hipLaunchKernelGGL((k_fs_big<L2, 2, L1>), grid, dim3(TPB), 0, s, a);
  else if (mode == 4) hipLaunchKernelGGL((k_fs_big<L2, 4, L1>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fs_big<L2, 3, L1>), grid, dim3(TPB), 0, s, a);
}

}  // namespace smx
