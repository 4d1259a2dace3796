// CPU emulation of one launch of the fused decimated kernel (tensor-cuda-fft-_amd/csrc/smx_decim.hip).
// TEST INFRASTRUCTURE: runs the same __host__ __device__ phase functions as the GPU kernel, one
// "thread" at a time, with barriers replaced by loop boundaries.  Built by tests/emu/build.sh with g++.
#include <cstdlib>
#include <cstring>
#include <array>
#include <vector>
#include "smx_core.h"
#include "smx_tables.h"

using namespace smx;

// one LDS round of the unpack for every "thread" (barriers = loop boundaries), rounds 0 .. N-1
template <int NB, int MODE, int ROUND>
static void unpack_rounds(std::vector<TState<NB>>& st, std::vector<cf>& lds, const Geom& g,
                          const FilterArgs& fa, int b, int d0, const std::vector<ZSave<NB>>& zsave,
                          std::vector<cf>* gsv) {
  if constexpr (ROUND < UnpackRounds<NB>::N) {
    for (int tid = 0; tid < TPB; ++tid) unpack_phase1<NB, ROUND>(st[tid], lds.data(), tid >> 4, tid & 15);
    for (int tid = 0; tid < TPB; ++tid) {
      const int j = tid & 15, q = tid >> 4, d = d0 + 2 * j;
      unpack_phase2<NB, MODE, ROUND>(st[tid], lds.data(), g, fa, b, d, d < g.D, q, j, zsave[tid], nullptr,
                                     gsv ? &(*gsv)[tid] : nullptr);
    }
    unpack_rounds<NB, MODE, ROUND + 1>(st, lds, g, fa, b, d0, zsave, gsv);
  }
}

template <int R>
static void store8(TState<8>& st, const cf* E, const cf* bt_r, int t, int j, int r) {
  if constexpr (R < 8) {
    if (r == R) fwd_phase2_store<8, R>(st, E, bt_r, t, j);
    else store8<R + 1>(st, E, bt_r, t, j, r);
  }
}
template <int R>
static void from8(TState<8>& st, const cf* bt_r, cf* E, int q, int j, int r) {
  if constexpr (R < 8) {
    if (r == R) inv_phase1_from<8, R>(st, bt_r, E, q, j);
    else from8<R + 1>(st, bt_r, E, q, j, r);
  }
}

// NB == 8: the full-spectrum kernel for N = 2048 (per-residue spectra + 8-point transform across them)
template <int NB, int MODE>
static void run(const float* xin, const FilterArgs& fa, float* yout, const Geom& g, int stagger) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L), tq = make_tq(g.N);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<NB>> st(TPB);
  std::vector<cf> lds(2 * EX);
  for (int bid = 0; bid < g.B * ndt; ++bid) {
    const int b = bid / ndt, d0 = (bid % ndt) * DT;
    const float* xb = xin + (size_t)b * g.R * g.D;
    float* yb = yout + (size_t)b * g.R * g.D;
    const int r0 = (stagger && NB != 8) ? (bid * 7) % g.L : 0;
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
        load_tile<true>(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
        if constexpr (NB == 8) fwd_phase1<NB>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        else {                       // the streaming loops of smx_decim.hip: c^q from the table row (load_cp)
          load_cp(tq.data() + ((size_t)t * g.L + r) * 16, st[tid].cp);
          fwd_phase1_cp<NB>(st[tid], E, t, j);
        }
      }
      for (int tid = 0; tid < TPB; ++tid) {
        if constexpr (NB == 8) store8<0>(st[tid], E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15, r);
        else fwd_phase2<NB, true>(st[tid], E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15);
      }
    }
    if constexpr (NB == 8)
      for (int tid = 0; tid < TPB; ++tid) residue_fft8<-1>(st[tid]);
    // unpack + filter in LDS rounds, as unpack_filter() in smx_decim.hip (barriers = loop boundaries)
    std::vector<ZSave<NB>> zsave(TPB);
    for (int tid = 0; tid < TPB; ++tid) zsave[tid] = save_z<NB>(st[tid]);
    std::vector<cf> gsv(TPB, mk(0.f, 0.f));
    const bool want_gs = MODE == 1 && fa.gsc != nullptr;
    unpack_rounds<NB, MODE, 0>(st, lds, g, fa, b, d0, zsave, want_gs ? &gsv : nullptr);
    if (want_gs)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        if (d >= g.D) continue;
        float sa = 0.f, sb = 0.f;
        for (int t = 0; t < 16; ++t) { sa += gsv[t * 16 + j].x; sb += gsv[t * 16 + j].y; }
        fa.gsc[(size_t)b * g.D + d] = sa; fa.gsc[(size_t)b * g.D + d + 1] = sb;
      }
    if (!yout) {
      for (int tid = 0; tid < TPB; ++tid) {
        const int d = d0 + 2 * (tid & 15);
        store_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
      }
      continue;
    }
    if constexpr (NB == 8)
      for (int tid = 0; tid < TPB; ++tid) residue_fft8<+1>(st[tid]);
    for (int i = 0; i < g.L; ++i) {
      const int r = (r0 + i) % g.L;
      cf* E = lds.data() + (i & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid) {
        if constexpr (NB == 8) from8<0>(st[tid], bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15, r);
        else inv_phase1<NB, true>(st[tid], bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        if constexpr (NB == 8) inv_phase2<NB>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        else {
          load_cp(tq.data() + ((size_t)t * g.L + r) * 16, st[tid].cp);
          inv_phase2_gather<NB>(st[tid], E, t, j);
          fft16<+1>(st[tid].v);
        }
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
    for (int tid = 0; tid < TPB; ++tid) {
      const int d = d0 + 2 * (tid & 15);
      store_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
    }
  }
}

// General shapes (include/smx.h, smx_*_ex): x / y have R <= N rows, k <= N/2 + 1 kept bins.
extern "C" int emu_fused_ex2(int mode, const float* xin, const float* w_re, const float* w_im,
                             const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                             int B, int R, int D, int F, int N, int k, int conj_w, int stagger,
                             const float* sc, float* gsc);
extern "C" int emu_fused_ex(int mode, const float* xin, const float* w_re, const float* w_im,
                            const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                            int B, int R, int D, int F, int N, int k, int conj_w, int stagger) {
  return emu_fused_ex2(mode, xin, w_re, w_im, bias, yout, xk, pslab, gb_part, B, R, D, F, N, k, conj_w, stagger,
                       nullptr, nullptr);
}
extern "C" int emu_fused_ex2(int mode, const float* xin, const float* w_re, const float* w_im,
                             const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                             int B, int R, int D, int F, int N, int k, int conj_w, int stagger,
                             const float* sc, float* gsc) {
  if (N % M || D % 2 || R > N || k > N / 2 + 1 || k > F) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = F; g.k = k; g.L = N / M; g.R = R;
  g.inv_n = (float)(1.0 / (double)N);
  const int kb = k > N / 2 ? N / 2 : k;
  int nb = kb > 256 ? 4 : kb > 128 ? 2 : 1;
  if (kb > 512) {
    if (g.L != 8) return -2;
    nb = 8;
  }
  FilterArgs fa{};
  fa.w_re = w_re; fa.w_im = w_im; fa.bias = bias; fa.conj_w = conj_w;
  fa.xk_out = mode == 0 ? xk : nullptr;
  fa.xk_in = mode == 1 ? xk : nullptr;
  fa.pslab = pslab; fa.gb_part = gb_part;
  fa.sc = sc; fa.gsc = gsc;
  if (mode == 0) {
    if (nb == 1) run<1, 0>(xin, fa, yout, g, stagger);
    else if (nb == 2) run<2, 0>(xin, fa, yout, g, stagger);
    else if (nb == 4) run<4, 0>(xin, fa, yout, g, stagger);
    else run<8, 0>(xin, fa, yout, g, stagger);
  } else {
    if (nb == 1) run<1, 1>(xin, fa, yout, g, stagger);
    else if (nb == 2) run<2, 1>(xin, fa, yout, g, stagger);
    else if (nb == 4) run<4, 1>(xin, fa, yout, g, stagger);
    else run<8, 1>(xin, fa, yout, g, stagger);
  }
  return 0;
}

extern "C" int emu_fused(int mode, const float* xin, const float* w_re, const float* w_im,
                         const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                         int B, int N, int D, int F, int conj_w, int stagger) {
  const int k = F < N / 2 ? F : N / 2;
  if (k > 512) return -2;
  return emu_fused_ex(mode, xin, w_re, w_im, bias, yout, xk, pslab, gb_part, B, N, D, F, N, k, conj_w,
                      stagger);
}

// ---- sixteen-row decimation (k_fused16 in smx_decim.hip): N = 16 P, any P; rows >= R zero-padded / cropped ----
template <int NB, int MODE>
static void run16(const float* xin, const FilterArgs& fa, float* yout, const Geom& g) {
  std::vector<cf> tw = make_tw(g.N), v16 = make_v16(g.N), b16 = make_b16(g.N);
  const int ndt = (g.D + DT - 1) / DT, T = g.L;
  std::vector<TState<NB>> st(TPB);
  std::vector<cf> lds(2 * EX);
  for (int bid = 0; bid < g.B * ndt; ++bid) {
    const int b = bid / ndt, d0 = (bid % ndt) * DT;
    const float* xb = xin + (size_t)b * g.R * g.D;
    float* yb = yout ? yout + (size_t)b * g.R * g.D : nullptr;
    const int rot = (bid * 5) % T;
    for (int tid = 0; tid < TPB; ++tid) {
      for (int s = 0; s < 16 * NB; ++s) st[tid].acc[s] = mk(0.f, 0.f);
      const int d = d0 + 2 * (tid & 15);
      if (NB == 1) prefetch_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
    }
    for (int i = 0; i < T; ++i) {
      const int tau = (rot + i) % T;
      cf* E = lds.data() + (i & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_tile16<true>(xb + (d < g.D ? d : g.D - 2), g, t, tau, st[tid].v);
        const int r = 16 * tau + t;
        fwd_phase1<NB>(st[tid], tw[r < g.N ? r : g.N - 1], E, t, j);
      }
      for (int tid = 0; tid < TPB; ++tid)
        fwd16_phase2<NB>(st[tid], E, v16.data(), b16.data() + (size_t)tau * 32, tid >> 4, tid & 15);
    }
    std::vector<ZSave<NB>> zsave(TPB);
    for (int tid = 0; tid < TPB; ++tid) zsave[tid] = save_z<NB>(st[tid]);
    if (NB == 2 && MODE == 1)
      for (int tid = 0; tid < TPB; ++tid)
        prefetch_io<NB, MODE>(st[tid], g, fa, b, d0 + 2 * (tid & 15), d0 + 2 * (tid & 15) < g.D, tid >> 4);
    unpack_rounds<NB, MODE, 0>(st, lds, g, fa, b, d0, zsave, nullptr);
    if (NB == 2 && MODE == 1)
      for (int tid = 0; tid < TPB; ++tid)
        store_io<NB, MODE>(st[tid], g, fa, b, d0 + 2 * (tid & 15), d0 + 2 * (tid & 15) < g.D, tid >> 4);
    if (yout) {
      for (int i = 0; i < T; ++i) {
        const int tau = (rot + i) % T;
        cf* E = lds.data() + (i & 1) * EX;
        for (int tid = 0; tid < TPB; ++tid)
          inv16_phase1<NB>(st[tid], v16.data(), b16.data() + (size_t)tau * 32, E, tid >> 4, tid & 15);
        for (int tid = 0; tid < TPB; ++tid) {
          const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j, r = 16 * tau + t;
          inv_phase2<NB>(st[tid], tw[r < g.N ? r : g.N - 1], E, t, j);
          store_tile16<true>(yb + d, g, t, tau, d < g.D, st[tid].v);
        }
      }
    }
    if (NB == 1)
      for (int tid = 0; tid < TPB; ++tid) {
        const int d = d0 + 2 * (tid & 15);
        store_io<NB, MODE>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
      }
  }
}

extern "C" int emu_fused16(int mode, const float* xin, const float* w_re, const float* w_im, const float* bias,
                           float* yout, float* xk, float* pslab, float* gb_part, int B, int R, int D, int F, int N,
                           int k, int conj_w) {
  if (N % 16 || D % 2 || R > N || k > N / 2 || k > F || k > 256 || k < 1) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = F; g.k = k; g.R = R;
  g.P = N / 16; g.L = (g.P + 15) / 16;
  g.inv_n = (float)(1.0 / (double)N);
  FilterArgs fa{};
  fa.w_re = w_re; fa.w_im = w_im; fa.bias = bias; fa.conj_w = conj_w;
  fa.xk_out = mode == 0 ? xk : nullptr;
  fa.xk_in = mode == 1 ? xk : nullptr;
  fa.pslab = pslab; fa.gb_part = gb_part;
  const int nb = k > 128 ? 2 : 1;
  if (mode == 0) { if (nb == 1) run16<1, 0>(xin, fa, yout, g); else run16<2, 0>(xin, fa, yout, g); }
  else { if (nb == 1) run16<1, 1>(xin, fa, yout, g); else run16<2, 1>(xin, fa, yout, g); }
  return 0;
}

// Four-step path (smx_core.h, end): (A) tile spectra -> workspace, (F) per-thread column pairs, (B) inverse.
template <int L, int MODE>
static void run_fourstep(const float* xin, const FilterArgs& fa, float* yout, const Geom& g) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<1>> st(TPB);
  std::vector<cf> lds(2 * EX), ws((size_t)L * EX);
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DT;
    const float* xb = xin + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_tile<true>(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
        fwd_phase1<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
      }
      for (int tid = 0; tid < TPB; ++tid)
        fwd_phase2_out(E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15, ws.data() + (size_t)r * EX + tid);
    }
    std::vector<cf> gsj(16, mk(0.f, 0.f));
    const bool want_gs = MODE == 1 && fa.gsc != nullptr;
    for (int u = 0; u <= 128; ++u)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        fs_columns<L, MODE>(ws.data(), g, fa, tw.data(), b, d, d < g.D, u, j, want_gs ? &gsj[j] : nullptr);
      }
    if (want_gs)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        if (d < g.D) { fa.gsc[(size_t)b * g.D + d] = gsj[j].x; fa.gsc[(size_t)b * g.D + d + 1] = gsj[j].y; }
      }
    if (!yout || MODE == 2) continue;
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        cf v[16];
        for (int s = 0; s < 16; ++s) v[s] = ws[(size_t)r * EX + s * TPB + tid];
        inv_phase1_in(v, bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}

// the column launch of the two-level path (k_fs_big in smx_fourstep.hip): phases = barrier-to-barrier loops
template <int L2, int L1 = 16, typename Each>
static void big_forward_emu(Each each, std::vector<BigState>& st, const cf* src, const cf* tw, cf* X) {
  each([&](int tid, int u, int ul, int t2, int j, int) {
    fsb_load<L2, L1>(st[tid], src, u, t2, j);
    fsb_pub<L2, L1>(st[tid].zp, tw, X, ul, t2, j);
  });
  each([&](int tid, int u, int ul, int t2, int j, int) { fsb_gather<L2, false, L1>(st[tid].zp, X, tw, u, ul, t2, j); });
  each([&](int tid, int, int ul, int t2, int j, int) { fsb_pub<L2, L1>(st[tid].zm, tw, X, ul, t2, j); });
  each([&](int tid, int u, int ul, int t2, int j, int) { fsb_gather<L2, true, L1>(st[tid].zm, X, tw, u, ul, t2, j); });
}
template <int L2, int L1 = 16, typename Each>
static void big_inverse_emu(Each each, std::vector<BigState>& st, cf* dst, const cf* tw, cf* X) {
  each([&](int tid, int, int ul, int t2, int j, int) { fsb_unpub<L2, false, L1>(st[tid].zp, X, tw, ul, t2, j); });
  each([&](int tid, int u, int ul, int t2, int j, int) { fsb_ungather<L2, L1>(st[tid].zp, dst, X, tw, u, ul, t2, j); });
  each([&](int tid, int u, int ul, int t2, int j, int) {
    if (u != 0 && u != 128) fsb_unpub<L2, true, L1>(st[tid].zm, X, tw, ul, t2, j);
  });
  each([&](int tid, int u, int ul, int t2, int j, int) {
    if (u != 0 && u != 128) fsb_ungather<L2, L1>(st[tid].zm, dst, X, tw, (256 - u) & 255, ul, t2, j);
  });
}
template <int L2, int MODE, int L1 = 16>
static void big_columns(cf* ws, const Geom& g, const FilterArgs& fa, const cf* tw, int b, int d0, std::vector<cf>* gsj) {
  constexpr int UPB = 16 / L2;
  std::vector<BigState> st(TPB);
  std::vector<cf> X(EX);
  for (int by = 0; by < (129 + UPB - 1) / UPB; ++by) {
    auto each = [&](auto f) {
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, sub = tid >> 4, t2 = sub % L2, ul = sub / L2, u = by * UPB + ul;
        if (u <= 128) f(tid, u, ul, t2, j, d0 + 2 * j);
      }
    };
    if constexpr (MODE == 4) {
      each([&](int tid, int u, int, int t2, int, int d) { fsb_synth<L2, L1>(st[tid], g, fa, b, d, d < g.D, u, t2); });
    } else {
      big_forward_emu<L2, L1>(each, st, ws, tw, X.data());
      each([&](int tid, int u, int, int t2, int j, int d) {
        fsb_pairs<L2, MODE, L1>(st[tid], g, fa, b, d, d < g.D, u, t2, (MODE == 1 && gsj) ? &(*gsj)[j] : nullptr);
      });
    }
    if constexpr (MODE == 0 || MODE == 1 || MODE == 4) big_inverse_emu<L2, L1>(each, st, ws, tw, X.data());
  }
}
template <int L2, int MODE, int L1 = 16>
static void run_fourstep_big(const float* xin, const FilterArgs& fa, float* yout, const Geom& g) {
  constexpr int L = L1 * L2;
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<1>> st(TPB);
  std::vector<cf> lds(2 * EX), ws((size_t)L * EX);
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DT;
    if constexpr (MODE != 4) {
      const float* xb = xin + (size_t)b * g.R * g.D;
      for (int r = 0; r < L; ++r) {
        cf* E = lds.data();
        for (int tid = 0; tid < TPB; ++tid) {
          const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
          load_tile<true>(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
          fwd_phase1<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        }
        for (int tid = 0; tid < TPB; ++tid)
          fwd_phase2_out(E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15, ws.data() + (size_t)r * EX + tid);
      }
    }
    std::vector<cf> gsj(16, mk(0.f, 0.f));
    const bool want_gs = MODE == 1 && fa.gsc != nullptr;
    big_columns<L2, MODE, L1>(ws.data(), g, fa, tw.data(), b, d0, want_gs ? &gsj : nullptr);
    if (want_gs)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        if (d < g.D) { fa.gsc[(size_t)b * g.D + d] = gsj[j].x; fa.gsc[(size_t)b * g.D + d + 1] = gsj[j].y; }
      }
    if (!yout || MODE == 2 || MODE == 3) continue;
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        cf v[16];
        for (int s = 0; s < 16; ++s) v[s] = ws[(size_t)r * EX + s * TPB + tid];
        inv_phase1_in(v, bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}

extern "C" int emu_fourstep_ex(int mode, const float* xin, const float* w_re, const float* w_im,
                               const float* bias, float* yout, float* xk, float* pslab, float* gb_part,
                               int B, int R, int D, int F, int N, int k, int conj_w, const float* sc,
                               float* gsc) {
  if (N % M || D % 2 || R > N || k > N / 2 + 1 || k > F) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = F; g.k = k; g.L = N / M; g.R = R;
  g.inv_n = (float)(1.0 / (double)N);
  FilterArgs fa{};
  fa.w_re = w_re; fa.w_im = w_im; fa.bias = bias; fa.conj_w = conj_w;
  fa.xk_out = mode == 0 ? xk : nullptr;
  fa.xk_in = mode == 1 ? xk : nullptr;
  fa.pslab = pslab; fa.gb_part = gb_part;
  fa.sc = sc; fa.gsc = gsc;
  if (g.L == 8) { if (mode == 0) run_fourstep<8, 0>(xin, fa, yout, g); else run_fourstep<8, 1>(xin, fa, yout, g); }
  else if (g.L == 16) { if (mode == 0) run_fourstep<16, 0>(xin, fa, yout, g); else run_fourstep<16, 1>(xin, fa, yout, g); }
  else if (g.L == 32) { if (mode == 0) run_fourstep<32, 0>(xin, fa, yout, g); else run_fourstep<32, 1>(xin, fa, yout, g); }
  else if (g.L == 5) { if (mode == 0) run_fourstep<5, 0>(xin, fa, yout, g); else run_fourstep<5, 1>(xin, fa, yout, g); }
  else if (g.L == 12) { if (mode == 0) run_fourstep<12, 0>(xin, fa, yout, g); else run_fourstep<12, 1>(xin, fa, yout, g); }
  else if (g.L == 24) { if (mode == 0) run_fourstep<24, 0>(xin, fa, yout, g); else run_fourstep<24, 1>(xin, fa, yout, g); }
  else if (g.L == 26) { if (mode == 0) run_fourstep<26, 0>(xin, fa, yout, g); else run_fourstep<26, 1>(xin, fa, yout, g); }
  else if (g.L == 17) { if (mode == 0) run_fourstep<17, 0>(xin, fa, yout, g); else run_fourstep<17, 1>(xin, fa, yout, g); }
  else if (g.L == 25) { if (mode == 0) run_fourstep<25, 0>(xin, fa, yout, g); else run_fourstep<25, 1>(xin, fa, yout, g); }
  else if (g.L == 64) { if (mode == 0) run_fourstep_big<4, 0>(xin, fa, yout, g); else run_fourstep_big<4, 1>(xin, fa, yout, g); }
  else if (g.L == 128) { if (mode == 0) run_fourstep_big<8, 0>(xin, fa, yout, g); else run_fourstep_big<8, 1>(xin, fa, yout, g); }
  else if (g.L == 256) { if (mode == 0) run_fourstep_big<16, 0>(xin, fa, yout, g); else run_fourstep_big<16, 1>(xin, fa, yout, g); }
  // round 3: first-level lengths 9 ... 15 (a multiple of L2, one with a padded thread, one per L2)
  else if (g.L == 48) { if (mode == 0) run_fourstep_big<4, 0, 12>(xin, fa, yout, g); else run_fourstep_big<4, 1, 12>(xin, fa, yout, g); }
  else if (g.L == 36) { if (mode == 0) run_fourstep_big<4, 0, 9>(xin, fa, yout, g); else run_fourstep_big<4, 1, 9>(xin, fa, yout, g); }
  else if (g.L == 52) { if (mode == 0) run_fourstep_big<4, 0, 13>(xin, fa, yout, g); else run_fourstep_big<4, 1, 13>(xin, fa, yout, g); }
  else if (g.L == 80) { if (mode == 0) run_fourstep_big<8, 0, 10>(xin, fa, yout, g); else run_fourstep_big<8, 1, 10>(xin, fa, yout, g); }
  else if (g.L == 144) { if (mode == 0) run_fourstep_big<16, 0, 9>(xin, fa, yout, g); else run_fourstep_big<16, 1, 9>(xin, fa, yout, g); }
  else return -2;
  return 0;
}

// Synthesis from a given one-sided spectrum (smx_irfft_ex): synth_fill + inverse loop (k <= 512 and the
// self-paired Nyquist slot), or fs_synth_columns + inverse tiles (fourstep != 0).
template <int NB>
static void run_synth(const FilterArgs& fa, float* yout, const Geom& g) {
  std::vector<cf> bt = make_bt(g.N, g.L), tq = make_tq(g.N);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<NB>> st(TPB);
  std::vector<cf> lds(2 * EX);
  for (int bid = 0; bid < g.B * ndt; ++bid) {
    const int b = bid / ndt, d0 = (bid % ndt) * DT;
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int tid = 0; tid < TPB; ++tid) {
      const int d = d0 + 2 * (tid & 15);
      synth_fill<NB>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
    }
    for (int r = 0; r < g.L; ++r) {
      cf* E = lds.data() + (r & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid)
        inv_phase1<NB, true>(st[tid], bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_cp(tq.data() + ((size_t)t * g.L + r) * 16, st[tid].cp);       // as inverse_loop in smx_decim.hip
        inv_phase2_gather<NB>(st[tid], E, t, j);
        fft16<+1>(st[tid].v);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}
template <int L>
static void run_fs_synth(const FilterArgs& fa, float* yout, const Geom& g) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<1>> st(TPB);
  std::vector<cf> lds(2 * EX), ws((size_t)L * EX);
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DT;
    for (int u = 0; u <= 128; ++u)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        fs_synth_columns<L>(ws.data(), g, fa, tw.data(), b, d, d < g.D, u, j);
      }
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        cf v[16];
        for (int s = 0; s < 16; ++s) v[s] = ws[(size_t)r * EX + s * TPB + tid];
        inv_phase1_in(v, bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}
static void run_synth8(const FilterArgs& fa, float* yout, const Geom& g) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<8>> st(TPB);
  std::vector<cf> lds(2 * EX);
  for (int bid = 0; bid < g.B * ndt; ++bid) {
    const int b = bid / ndt, d0 = (bid % ndt) * DT;
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int tid = 0; tid < TPB; ++tid) {
      const int d = d0 + 2 * (tid & 15);
      synth_fill<8>(st[tid], g, fa, b, d, d < g.D, tid >> 4);
      residue_fft8<+1>(st[tid]);
    }
    for (int r = 0; r < 8; ++r) {
      cf* E = lds.data() + (r & 1) * EX;
      for (int tid = 0; tid < TPB; ++tid) from8<0>(st[tid], bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15, r);
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<8>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}
extern "C" int emu_synth(const float* spec, float* yout, int B, int R, int D, int N, int k, float scale,
                         int hermitian, int fourstep) {
  if (N % M || D % 2 || R > N || k > N / 2 + 1) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = k; g.k = k; g.L = N / M; g.R = R;
  g.inv_n = (float)(1.0 / (double)N);
  FilterArgs fa{};
  fa.xk_in = spec; fa.sp_scale = scale; fa.sp_herm = hermitian;
  if (fourstep == 2) {                     // eight bands in registers (N = 2048)
    if (g.L != 8) return -2;
    run_synth8(fa, yout, g);
    return 0;
  }
  if (fourstep) {
    switch (g.L) {
      case 5: run_fs_synth<5>(fa, yout, g); break;
      case 8: run_fs_synth<8>(fa, yout, g); break;
      case 12: run_fs_synth<12>(fa, yout, g); break;
      case 16: run_fs_synth<16>(fa, yout, g); break;
      case 32: run_fs_synth<32>(fa, yout, g); break;
      case 24: run_fs_synth<24>(fa, yout, g); break;
      case 17: run_fs_synth<17>(fa, yout, g); break;
      case 64: run_fourstep_big<4, 4>(nullptr, fa, yout, g); break;
      case 128: run_fourstep_big<8, 4>(nullptr, fa, yout, g); break;
      case 256: run_fourstep_big<16, 4>(nullptr, fa, yout, g); break;
      case 36: run_fourstep_big<4, 4, 9>(nullptr, fa, yout, g); break;
      case 48: run_fourstep_big<4, 4, 12>(nullptr, fa, yout, g); break;
      case 80: run_fourstep_big<8, 4, 10>(nullptr, fa, yout, g); break;
      default: return -2;
    }
    return 0;
  }
  const int kb = k > N / 2 ? N / 2 : k;
  if (kb > 512) return -2;
  if (kb > 256) run_synth<4>(fa, yout, g);
  else if (kb > 128) run_synth<2>(fa, yout, g);
  else run_synth<1>(fa, yout, g);
  return 0;
}

// complex sequence FFT through the two-level path (MODE 3): z viewed as real (B, N, 2 Dc), out (B, N, Dc) complex
extern "C" int emu_cfft_big(const float* zin, float* out, int B, int N, int D2) {
  if (N % M || D2 % 2) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D2; g.F = N / 2 + 1; g.k = N / 2 + 1; g.L = N / M; g.R = N;
  g.inv_n = (float)(1.0 / (double)N);
  FilterArgs fa{};
  fa.xk_out = out;
  if (g.L == 64) run_fourstep_big<4, 3>(zin, fa, nullptr, g);
  else if (g.L == 128) run_fourstep_big<8, 3>(zin, fa, nullptr, g);
  else if (g.L == 256) run_fourstep_big<16, 3>(zin, fa, nullptr, g);
  else if (g.L == 36) run_fourstep_big<4, 3, 9>(zin, fa, nullptr, g);
  else if (g.L == 80) run_fourstep_big<8, 3, 10>(zin, fa, nullptr, g);
  else return -2;
  return 0;
}

// Rank-one filter on the four-step path (fs_conv_columns): dir 0 forward (xs receives the tile spectra of x),
// dir 1 backward (xs = the forward's, p_out (N complex) and gs (B, D) receive the sums).
template <int L, int DIR>
static void run_conv(const float* xin, float* yout, const Geom& g, const ConvArgs& ca0, cf* xs, cf* p_out,
                     float* gs) {
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<1>> st(TPB);
  std::vector<cf> lds(2 * EX), ws((size_t)L * EX);
  if (DIR == 1) for (int f = 0; f < g.N; ++f) p_out[f] = mk(0.f, 0.f);
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DT;
    const float* xb = xin + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_tile<true>(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
        fwd_phase1<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
      }
      for (int tid = 0; tid < TPB; ++tid)
        fwd_phase2_out(E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15, ws.data() + (size_t)r * EX + tid);
    }
    cf* xsb = xs + (size_t)wg * L * EX;
    if (DIR == 0) std::memcpy(xsb, ws.data(), sizeof(cf) * (size_t)L * EX);
    ConvArgs ca = ca0;
    std::vector<cf> racc(16, mk(0.f, 0.f));
    for (int u = 0; u <= 128; ++u)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        cf pp[L], pm[L], rr = mk(0.f, 0.f);
        fs_conv_columns<L, DIR>(ws.data(), ws.data(), xsb, g, ca, tw.data(), b, d, d < g.D, u, j, pp, pm, &rr);
        if (DIR == 1) {
          const int fum = (256 - u) & 255;
          for (int f2 = 0; f2 < L; ++f2) {
            p_out[u + 256 * f2] = cadd(p_out[u + 256 * f2], pp[f2]);
            if (u != 0 && u != 128) p_out[fum + 256 * f2] = cadd(p_out[fum + 256 * f2], pm[f2]);
          }
          racc[j] = cadd(racc[j], rr);
        }
      }
    if (DIR == 1 && gs)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        if (d >= g.D) continue;
        gs[(size_t)b * g.D + d] = (racc[j].x + racc[j].y) * 0.5f * g.inv_n;
        gs[(size_t)b * g.D + d + 1] = (racc[j].x - racc[j].y) * 0.5f * g.inv_n;
      }
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        cf v[16];
        for (int s = 0; s < 16; ++s) v[s] = ws[(size_t)r * EX + s * TPB + tid];
        inv_phase1_in(v, bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        if (ca.sc && d < g.D)
          for (int uu = 0; uu < 16; ++uu)
            st[tid].v[uu] = mk(st[tid].v[uu].x * ca.sc[(size_t)b * g.D + d], st[tid].v[uu].y * ca.sc[(size_t)b * g.D + d + 1]);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}

// rank-one filter on the two-level columns (k_fs_conv_big): barrier-to-barrier loops over the block's threads
template <int L2, int DIR>
static void run_conv_big(const float* xin, float* yout, const Geom& g, const ConvArgs& ca, cf* xs, cf* p_out,
                         float* gs) {
  constexpr int L = 16 * L2, UPB = 16 / L2;
  std::vector<cf> tw = make_tw(g.N), bt = make_bt(g.N, g.L);
  const int ndt = (g.D + DT - 1) / DT;
  std::vector<TState<1>> st(TPB);
  std::vector<cf> lds(2 * EX), ws((size_t)L * EX), X(EX);
  std::vector<BigState> sg(TPB), sx(TPB);
  if (DIR == 1) for (int f = 0; f < g.N; ++f) p_out[f] = mk(0.f, 0.f);
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DT;
    const float* xb = xin + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        load_tile<true>(xb + (d < g.D ? d : g.D - 2), g, t, r, st[tid].v);
        fwd_phase1<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
      }
      for (int tid = 0; tid < TPB; ++tid)
        fwd_phase2_out(E, bt.data() + (size_t)r * BT_STRIDE, tid >> 4, tid & 15, ws.data() + (size_t)r * EX + tid);
    }
    cf* xsb = xs + (size_t)wg * L * EX;
    if (DIR == 0) std::memcpy(xsb, ws.data(), sizeof(cf) * (size_t)L * EX);
    std::vector<cf> racc(16, mk(0.f, 0.f));
    for (int by = 0; by < (129 + UPB - 1) / UPB; ++by) {
      auto each = [&](auto f) {
        for (int tid = 0; tid < TPB; ++tid) {
          const int j = tid & 15, sub = tid >> 4, t2 = sub % L2, ul = sub / L2, u = by * UPB + ul;
          if (u <= 128) f(tid, u, ul, t2, j, d0 + 2 * j);
        }
      };
      if (DIR == 1) big_forward_emu<L2>(each, sx, xsb, tw.data(), X.data());
      big_forward_emu<L2>(each, sg, ws.data(), tw.data(), X.data());
      each([&](int tid, int u, int, int t2, int j, int d) {
        if (DIR == 1) {
          cf rr;
          fsb_conv_sums<L2>(sg[tid], sx[tid], g, ca, b, d, d < g.D, u, t2, rr,
                            [&](int fp, int fm, cf vp, cf vm, bool one_col) {
                              p_out[fp] = cadd(p_out[fp], vp);
                              if (!one_col) p_out[fm] = cadd(p_out[fm], vm);
                            });
          racc[j] = cadd(racc[j], rr);
        }
        fsb_conv_scale<L2, DIR>(sg[tid], g, ca, d < g.D, u, t2);
      });
      big_inverse_emu<L2>(each, sg, ws.data(), tw.data(), X.data());
    }
    if (DIR == 1 && gs)
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + 2 * j;
        if (d >= g.D) continue;
        gs[(size_t)b * g.D + d] = (racc[j].x + racc[j].y) * 0.5f * g.inv_n;
        gs[(size_t)b * g.D + d + 1] = (racc[j].x - racc[j].y) * 0.5f * g.inv_n;
      }
    float* yb = yout + (size_t)b * g.R * g.D;
    for (int r = 0; r < L; ++r) {
      cf* E = lds.data();
      for (int tid = 0; tid < TPB; ++tid) {
        cf v[16];
        for (int s = 0; s < 16; ++s) v[s] = ws[(size_t)r * EX + s * TPB + tid];
        inv_phase1_in(v, bt.data() + (size_t)r * BT_STRIDE, E, tid >> 4, tid & 15);
      }
      for (int tid = 0; tid < TPB; ++tid) {
        const int j = tid & 15, t = tid >> 4, d = d0 + 2 * j;
        inv_phase2<1>(st[tid], tw[(size_t)t * g.L + r], E, t, j);
        if (ca.sc && d < g.D)
          for (int uu = 0; uu < 16; ++uu)
            st[tid].v[uu] = mk(st[tid].v[uu].x * ca.sc[(size_t)b * g.D + d], st[tid].v[uu].y * ca.sc[(size_t)b * g.D + d + 1]);
        store_tile<true>(yb + d, g, t, r, d < g.D, st[tid].v);
      }
    }
  }
}

extern "C" int emu_conv(int dir, const float* xin, const float* h_re, const float* h_im, const float* sc,
                        float* yout, float* xs, float* p_out, float* gs, int B, int R, int D, int N) {
  if (N % M || D % 2 || R > N) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = N / 2 + 1; g.k = N / 2 + 1; g.L = N / M; g.R = R;
  g.inv_n = (float)(1.0 / (double)N);
  ConvArgs ca{};
  ca.h_re = h_re; ca.h_im = h_im; ca.sc = sc;
  if (g.L == 8) { if (dir == 0) run_conv<8, 0>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv<8, 1>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else if (g.L == 16) { if (dir == 0) run_conv<16, 0>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv<16, 1>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else if (g.L == 32) { if (dir == 0) run_conv_big<2, 0>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv_big<2, 1>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else if (g.L == 64) { if (dir == 0) run_conv_big<4, 0>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv_big<4, 1>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else if (g.L == 256) { if (dir == 0) run_conv_big<16, 0>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv_big<16, 1>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else return -2;
  return 0;
}

// Rank-one filter in one launch (k_conv1, smx_conv1.hip): the 512 threads of a workgroup, barrier to barrier.
// xs uses that kernel's layout ([workgroup][16 LP][512]); p_out / gs as emu_conv.
template <int LP, int R, int NJ>
static void conv1_fwd_phase2(std::vector<std::array<cf, 64>>& acc, const cf* lds) {
  constexpr int TS = 16 * NJ, EXJ = 256 * NJ;
  for (int tid = 0; tid < c1_tpb<NJ>(); ++tid) {
    const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ;
    c1_fwd_phase2<LP, R, NJ>(reinterpret_cast<cf(&)[16 * LP]>(*acc[tid].data()), lds + (2 * p + (R & 1)) * EXJ, t, j);
  }
}
template <int LP, int R, int NJ>
static void conv1_inv_phase1(std::vector<std::array<cf, 64>>& acc, std::vector<std::array<cf, 16>>& v, cf* lds) {
  constexpr int TS = 16 * NJ, EXJ = 256 * NJ;
  for (int tid = 0; tid < c1_tpb<NJ>(); ++tid) {
    const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ;
    c1_inv_phase1<LP, R, NJ>(reinterpret_cast<const cf(&)[16 * LP]>(*acc[tid].data()),
                             reinterpret_cast<cf(&)[16]>(*v[tid].data()), lds + 2 * p * EXJ, t, j);
  }
}
template <int LP, int DIR, int NJ>
static void run_conv1(const float* xin, float* yout, const Geom& g, const ConvArgs& ca, cf* xs, cf* p_out, float* gs) {
  std::vector<cf> tw = make_tw(g.N);
  constexpr int TPBJ = c1_tpb<NJ>(), TS = 16 * NJ, EXJ = 256 * NJ, DTJ = 2 * NJ;
  const int ndt = (g.D + DTJ - 1) / DTJ, N = g.N;
  Geom h = g;
  h.N = g.N / 2; h.L = LP;
  Geom hh = h;
  hh.R = g.R - h.N;
  const bool fold = 2 * g.R > g.N;
  std::vector<std::array<cf, 64>> acc(TPBJ);
  std::vector<std::array<cf, 16>> v(TPBJ);
  std::vector<cf> lds(4 * EXJ + 512 * LP);
  if (DIR == 1) for (int f = 0; f < N; ++f) p_out[f] = mk(0.f, 0.f);
  auto A = [&](int tid) -> cf(&)[16 * LP] { return reinterpret_cast<cf(&)[16 * LP]>(*acc[tid].data()); };
  auto V = [&](int tid) -> cf(&)[16] { return reinterpret_cast<cf(&)[16]>(*v[tid].data()); };
  for (int wg = 0; wg < g.B * ndt; ++wg) {
    const int b = wg / ndt, d0 = (wg % ndt) * DTJ;
    const float* xb = xin + (size_t)b * g.R * g.D;
    cf* Hs = lds.data() + 4 * EXJ;
    for (int tid = 0; tid < TPBJ; ++tid) c1_stage_h<NJ>(ca, N, g.inv_n, Hs, tid);
    for (int R = 0; R < LP; ++R) {
      for (int tid = 0; tid < TPBJ; ++tid) {
        const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ, d = d0 + 2 * j;
        load_tile<true>(xb + (d < g.D ? d : g.D - 2), h, t, R, V(tid));
        if (fold) {                                   // rows > N / 2: x[n] +/- x[n + N/2]
          cf hi[16];
          load_tile<true>(xb + (size_t)h.N * g.D + (d < g.D ? d : g.D - 2), hh, t, R, hi);
          c1_fold_in(V(tid), hi, p);
        }
        c1_fwd_phase1<LP, NJ>(V(tid), tw.data(), lds.data() + (2 * p + (R & 1)) * EXJ, p, t, j, R);
      }
      if (R == 0) conv1_fwd_phase2<LP, 0, NJ>(acc, lds.data());
      if constexpr (LP >= 2) if (R == 1) conv1_fwd_phase2<LP, 1, NJ>(acc, lds.data());
      if constexpr (LP >= 4) { if (R == 2) conv1_fwd_phase2<LP, 2, NJ>(acc, lds.data()); if (R == 3) conv1_fwd_phase2<LP, 3, NJ>(acc, lds.data()); }
    }
    cf* xsb = xs + (size_t)wg * (16 * LP) * TPBJ;
    std::vector<cf> racc(NJ, mk(0.f, 0.f));
    for (int tid = 0; tid < TPBJ; ++tid) {
      const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ, d = d0 + 2 * j;
      c1_residues<LP, -1>(A(tid));
      if (DIR == 0) c1_mid_fwd<LP, NJ>(A(tid), Hs, xsb, p, t, tid);
    }
    if (DIR == 1) {
      for (int tid = 0; tid < TPBJ; ++tid) {
        const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ, d = d0 + 2 * j;
        const int dl = d < g.D ? d : g.D - 2;
        float sa = 1.f, sb = 1.f;
        if (ca.sc) { sa = ca.sc[(size_t)b * g.D + dl]; sb = ca.sc[(size_t)b * g.D + dl + 1]; }
        cf rr;
        const bool valid = d < g.D;
        c1_mid_bwd<LP, NJ>(A(tid), Hs, xsb, valid ? 0.5f * (sa + sb) : 0.f, valid ? 0.5f * (sa - sb) : 0.f, p, t, j, tid, rr,
                       [&](int grp, const float (&px)[16], const float (&py)[16]) {
                         for (int i = 0; i < 16; ++i) {
                           cf& o = p_out[c1_bin(p, t, c1_group_slot<LP>(grp, i))];
                           o = cadd(o, mk(px[i], py[i]));
                         }
                       });
        if (!valid) rr = mk(0.f, 0.f);
        racc[j] = cadd(racc[j], rr);
      }
      if (gs)
        for (int j = 0; j < NJ; ++j) {
          const int d = d0 + 2 * j;
          if (d >= g.D) continue;
          gs[(size_t)b * g.D + d] = (racc[j].x + racc[j].y) * 0.5f;
          gs[(size_t)b * g.D + d + 1] = (racc[j].x - racc[j].y) * 0.5f;
        }
    }
    for (int tid = 0; tid < TPBJ; ++tid) c1_residues<LP, +1>(A(tid));
    float* yb = yout + (size_t)b * g.R * g.D;
    cf* C = lds.data() + EXJ;
    for (int R = 0; R < LP; ++R) {
      if (R == 0) conv1_inv_phase1<LP, 0, NJ>(acc, v, lds.data());
      if constexpr (LP >= 2) if (R == 1) conv1_inv_phase1<LP, 1, NJ>(acc, v, lds.data());
      if constexpr (LP >= 4) { if (R == 2) conv1_inv_phase1<LP, 2, NJ>(acc, v, lds.data()); if (R == 3) conv1_inv_phase1<LP, 3, NJ>(acc, v, lds.data()); }
      for (int tid = 0; tid < TPBJ; ++tid) {
        const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ;
        c1_inv_phase2<LP, NJ>(V(tid), tw.data(), lds.data() + 2 * p * EXJ, p, t, j, R);
      }
      for (int tid = 0; tid < TPBJ; ++tid) c1_comb_write<NJ>(V(tid), C, tid / TS, tid % TS);
      for (int tid = 0; tid < TPBJ; ++tid) {
        const int p = tid / TS, lt = tid % TS, j = lt % NJ, t = lt / NJ, d = d0 + 2 * j;
        const int dl = d < g.D ? d : g.D - 2;
        float sa = 1.f, sb = 1.f;
        if (ca.sc) { sa = ca.sc[(size_t)b * g.D + dl]; sb = ca.sc[(size_t)b * g.D + dl + 1]; }
        if (fold) c1_comb_store<true, NJ, true>(V(tid), C, yb + d, h, p, t, lt, R, d < g.D, sa, sb);
        else c1_comb_store<true, NJ>(V(tid), C, yb + d, h, p, t, lt, R, d < g.D, sa, sb);
      }
    }
  }
}
template <int NJ>
static int emu_conv1_nj(int dir, const float* xin, float* yout, const Geom& g, const ConvArgs& ca, float* xs, float* p_out,
                        float* gs) {
  const int N = g.N;
  if (N == 512) { if (dir == 0) run_conv1<1, 0, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv1<1, 1, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else if (N == 1024) { if (dir == 0) run_conv1<2, 0, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv1<2, 1, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  else { if (dir == 0) run_conv1<4, 0, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); else run_conv1<4, 1, NJ>(xin, yout, g, ca, (cf*)xs, (cf*)p_out, gs); }
  return 0;
}
// nj = channel pairs per workgroup (16 or 8)
extern "C" int emu_conv1(int dir, const float* xin, const float* h_re, const float* h_im, const float* sc,
                         float* yout, float* xs, float* p_out, float* gs, int B, int R, int D, int N, int nj) {
  if (!(N == 512 || N == 1024 || N == 2048) || D % 2 || R > N || (nj != 16 && nj != 8)) return -2;
  Geom g;
  g.B = B; g.N = N; g.D = D; g.F = N / 2 + 1; g.k = N / 2 + 1; g.L = N / M; g.R = R;
  g.inv_n = (float)(1.0 / (double)N);
  ConvArgs ca{};
  ca.h_re = h_re; ca.h_im = h_im; ca.sc = sc;
  return nj == 8 ? emu_conv1_nj<8>(dir, xin, yout, g, ca, xs, p_out, gs) : emu_conv1_nj<16>(dir, xin, yout, g, ca, xs, p_out, gs);
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
