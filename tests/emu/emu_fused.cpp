// CPU emulation of one launch of the fused decimated kernel (tensor-cuda-fft-_amd/csrc/smx_decim.hip).
// TEST INFRASTRUCTURE: runs the same __host__ __device__ phase functions as the GPU kernel, one
// "thread" at a time, with barriers replaced by loop boundaries.  Built by tests/emu/build.sh with g++.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "smx_core.h"
#include "smx_tables.h"

using namespace smx;

template <int NB, int MODE>
static void run(const float* xin, const FilterArgs& fa, float* yout, const Geom& g, int stagger) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<NB>> st(TPB);
  std::vector<cf> lds(2 * EX);
  for (int bid = 0; bid < g.B * ndt; ++bid) {
    const int b = bid / ndt, d0 = (bid % ndt) * DT;
    const float* xb = xin + (size_t)b * g.N * g.D;
    float* yb = yout + (size_t)b * g.N * g.D;
    const int r0 = stagger ? (bid * 7) % g.L : 0;
    for (int tid = 0; tid < TPB; ++tid) {
      for (int s = 0; s < 16 * NB; ++s) st[tid].acc[s] = mk(0.f, 0.f);
      const int d = d0 + 2 * (tid & 15);
      prefetch_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
    }
    for (int i = 0; i < g.L; ++i) {
      const int r = (r0 + i) % g.L;
      cf* E = lds.data() + (i & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_tile(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
        fwd_phase1<NB>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
      }
      for (int tid = 0; tid < TPB; ++tid)
        fwd_phase2<NB>(st[tid], E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15);
    }
    // unpack + filter in LDS rounds, as unpack_filter() in smx_decim.hip (barriers = loop boundaries)
    std::vector<cf> zsave(TPB);
    for (int tid = 0; tid < TPB; ++tid) zsave[tid] = st[tid].acc[NB == 4 ? 16 : 0];
    for (int tid = 0; tid < TPB; ++tid) unpack_phase1<NB, 0>(st[tid], lds.data(), tid >> 4, tid & 15);
    for (int tid = 0; tid < TPB; ++tid) {
      const int j = tid & 15, q = tid >> 4, d = d0 + 2 * j;
      unpack_phase2<NB, MODE, 0>(st[tid], lds.data(), g, fa, b, d, d < g.D, q, j, zsave[tid]);
    }
    if constexpr (NB == 4) {
      for (int tid = 0; tid < TPB; ++tid) unpack_phase1<NB, 1>(st[tid], lds.data(), tid >> 4, tid & 15);
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, q = tid >> 4, d = d0 + 2 * j;
        unpack_phase2<NB, MODE, 1>(st[tid], lds.data(), g, fa, b, d, d < g.D, q, j, zsave[tid]);
      }
    }
    if (!yout) {
      for (int tid = 0; tid < TPB; ++tid) {
        const int d = d0 + 2 * (tid & 15);
        store_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
      }
      continue;
    }
    for (int i = 0; i < g.L; ++i) {
      const int r = (r0 + i) % g.L;
      cf* E = lds.data() + (i & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid)
        inv_phase1<NB>(st[tid], bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<NB>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        store_tile(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
    for (int tid = 0; tid < TPB; ++tid) {
      const int d = d0 + 2 * (tid & 15);
      store_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
    }
  }
}

extern "C" int emu_fused(int mode, const float* xin, const float* w_re, const float* w_im,
                         const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                         int B, int N, int D, int F, int conj_w, int stagger) {
  if (N % M || D % 2) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = F; g.k = F < N / 2 ? F : N / 2; g.L = N / M; g.R = N;
  g.inv_n = (float)(1.0 / (double)N);
  if (g.k > 512) return -2;
  FilterArgs fa{};
  fa.w_re = w_re; fa.w_im = w_im; fa.bias = bias; fa.conj_w = conj_w;
  fa.xk_out = mode == 0 ? xk : nullptr;
  fa.xk_in = mode == 1 ? xk : nullptr;
  fa.pslab = pslab; fa.gb_part = gb_part;
  const int nb = g.k > 256 ? 4 : g.k > 128 ? 2 : 1;
  if (mode == 0) {
    if (nb == 1) run<1, 0>(xin, fa, yout, g, stagger);
    else if (nb == 2) run<2, 0>(xin, fa, yout, g, stagger);
    else run<4, 0>(xin, fa, yout, g, stagger);
  } else {
    if (nb == 1) run<1, 1>(xin, fa, yout, g, stagger);
    else if (nb == 2) run<2, 1>(xin, fa, yout, g, stagger);
    else run<4, 1>(xin, fa, yout, g, stagger);
  }
  return 0;
}

// Dropout keep-mask of batch row b for the elements [0, row_elems) of that row, exactly as the kernels
// derive it (smx_core.h: drop_row_key / drop_hash): out[e] = 1 if the element survives.
extern "C" void emu_drop_mask(unsigned long long seed, unsigned long long counter, int b,
                              long long row_elems, unsigned thr, unsigned char* out) {
  const unsigned key = drop_row_key(seed, counter, b);
  for (long long e = 0; e < row_elems; ++e) {
    const unsigned h = drop_hash((unsigned)((unsigned long long)e >> 1), key);
    out[e] = ((e & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
  }
}
