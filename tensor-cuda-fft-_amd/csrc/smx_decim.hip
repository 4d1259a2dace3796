// smx_decim.hip -- decimated, output-pruned Stockham kernels for gfx950 (N = 256 L, k <= 512).
//
// Workgroup = 256 threads = 16 row-groups t x 16 packed channel pairs j, owning (batch row b,
// 32 channels).  A global row is 128 contiguous bytes per workgroup (16 lanes x float2), so every
// wave instruction moves 4 full 128-B segments.  x is read once, y is written once; the only
// other HBM traffic is the (B,k,D) spectrum (N/k times smaller than x).
//
// Replaces: reference fft_tensor/spectral_layers.py:88 (fft), :94-109 (filter), :112-116 (ifft, bias)
// and the autograd backward of the same lines.
#include "smx_launch.h"

namespace smx {

template <int NB>
__device__ __forceinline__ void zero_acc(TState<NB>& st) {
#pragma unroll
  for (int s = 0; s < 16 * NB; ++s) st.acc[s] = mk(0.f, 0.f);
}

// unpack + filter between the two loops, in LDS rounds of 32 slots (UnpackRounds in smx_core.h).
// Enters and leaves with the LDS free (barriers included).
template <int NB, int MODE, bool BATCHED, int ROUND>
__device__ __forceinline__ void unpack_round(TState<NB>& st, cf* lds, const Geom& g,
                                             const FilterArgs& fa, int b, int d, bool valid, int t,
                                             int j, const ZSave<NB>& zs, const WPre* wp, cf* gs) {
  if constexpr (ROUND < UnpackRounds<NB>::N) {
    __syncthreads();
    unpack_phase1<NB, ROUND>(st, lds, t, j);
    const cf* wl = nullptr;
    if constexpr (NB == 1 && MODE != 2) {
      if (wp) { stage_w(*wp, lds + EX, t * 16 + j, fa.conj_w); wl = lds + EX; }
    }
    __syncthreads();
    if constexpr (BATCHED) unpack_phase2_batched<NB, MODE, ROUND>(st, lds, g, fa, b, d, valid, t, j, zs, gs, wl);
    else unpack_phase2<NB, MODE, ROUND>(st, lds, g, fa, b, d, valid, t, j, zs, wl, gs);
    unpack_round<NB, MODE, BATCHED, ROUND + 1>(st, lds, g, fa, b, d, valid, t, j, zs, wp, gs);
  }
}

template <int NB, int MODE, bool BATCHED>
__device__ __forceinline__ void unpack_filter(TState<NB>& st, cf* lds, const Geom& g,
                                              const FilterArgs& fa, int b, int d, bool valid, int t,
                                              int j, const WPre* wp = nullptr) {
  const ZSave<NB> zs = save_z<NB>(st);
  // two bands, backward: the 16 rows of the saved spectrum this thread needs are requested here in one
  // burst (the tile registers of the loops are dead by now) and the slab rows leave right after the
  // unpack -- not as dependent load -> store pairs inside the slot loop
  if constexpr (NB == 2 && MODE == 1 && !BATCHED) prefetch_io<NB, MODE>(st, g, fa, b, d, valid, t);
  cf gs = mk(0.f, 0.f);                 // gradient of the per-row filter factor (summed unconditionally: a
  const bool want_gs = MODE == 1 && fa.gsc != nullptr;         // run-time pointer would push it to scratch)
  unpack_round<NB, MODE, BATCHED, 0>(st, lds, g, fa, b, d, valid, t, j, zs, wp, MODE == 1 ? &gs : nullptr);
  if constexpr (NB == 2 && MODE == 1 && !BATCHED) store_io<NB, MODE>(st, g, fa, b, d, valid, t);
  if constexpr (MODE == 1) {
    if (want_gs) {               // the 16 threads that share a channel pair: fixed-order sum through LDS
      __syncthreads();
      lds[t * 16 + j] = gs;
      __syncthreads();
      if (t == 0 && valid) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { sa += lds[u * 16 + j].x; sb += lds[u * 16 + j].y; }
        fa.gsc[(size_t)b * g.D + d] = sa;
        fa.gsc[(size_t)b * g.D + d + 1] = sb;
      }
    }
  }
}

// LayerNorm folded into the load (fused block only): per-row (mean, rstd) and this thread's two
// channels of gamma / beta.  st[u] belongs to row (16 u + t) L + r, like the tile itself.
struct LnLoad {
  const cf* sb;          // stats of batch row b, (N)
  float g0, g1, b0, b1;
};
template <int U0, int CNT>
__device__ __forceinline__ void load_stats(const cf* __restrict__ sb, const Geom& g, int t, int r,
                                           cf (&sv)[16]) {
  const cf* p = sb + (size_t)t * g.L + r;
#pragma unroll
  for (int u = U0; u < U0 + CNT; ++u) sv[u] = p[(size_t)u * 16 * g.L];
}

// One tile of the forward half.  LAST = no tile follows (nothing is prefetched): the loops below peel their last
// iteration instead of guarding the prefetches with `if (i + 1 < cnt)` -- with the guards the register allocator
// put the copies of the conditionally loaded registers right behind the loads (s_waitcnt vmcnt(7) eight
// instructions after the burst: the whole memory latency exposed once per tile).
// DROP (backward launches): the tile is g, multiplied by the dropout mask of the forward pass as it
// is moved into the working registers.  pj = this thread's pair index inside a row, (d >> 1).
template <int NB, bool LN, bool DROP, bool PAD, bool LAST>
__device__ __forceinline__ void forward_tile(TState<NB>& st, cf* E, const RowBuf& xb,
                                             const DecimArgs& a, int t, int j, int r, int rn, cf (&nx)[16],
                                             cf (&ns)[LN ? 16 : 1], const LnLoad* ln, Drop dr, unsigned pj) {
  const Geom& g = a.g;
  if constexpr (LN) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
      st.v[u] = mk(fmaf((nx[u].x - ns[u].x) * ns[u].y, ln->g0, ln->b0),
                   fmaf((nx[u].y - ns[u].x) * ns[u].y, ln->g1, ln->b1));
  } else if constexpr (DROP) {
    const unsigned hd = (unsigned)(g.D >> 1), pstride = 16u * (unsigned)g.L * hd;
    const unsigned p0 = ((unsigned)t * (unsigned)g.L + (unsigned)r) * hd + pj;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      st.v[u] = drop_apply(nx[u], drop_hash(p0 + (unsigned)u * pstride, dr.key), dr.thr, dr.scale);
  } else {
#pragma unroll
    for (int u = 0; u < 16; ++u) st.v[u] = nx[u];
  }
  // The next tile's 16 loads go out in two bursts, before and after the exchange barrier:
  // smoother request issue measured ~3 us faster per launch than one 16-load burst (and than four).
  if constexpr (!LAST) {
    load_rows<0, 8, PAD>(xb, rn, nx);
    if constexpr (LN) load_stats<0, 8>(ln->sb, g, t, rn, ns);
    __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks this burst below the transform)
  }
  fwd_phase1_cp<NB>(st, E, t, j);
  // the next tile's inter-pass twiddles, into the registers this tile's products have just freed
  if constexpr (!LAST) load_cp(a.tq + ((size_t)t * g.L + rn) * 16, st.cp);
  __syncthreads();
  if constexpr (!LAST) {
    load_rows<8, 8, PAD>(xb, rn, nx);
    if constexpr (LN) load_stats<8, 8>(ln->sb, g, t, rn, ns);
  }
  fwd_phase2<NB, true>(st, E, a.bt + (size_t)r * BT_STRIDE, t, j);
}

// forward half: accumulate residues [rbeg, rbeg+cnt) (visited in rotated order) into st.acc.
template <int NB, bool LN = false, bool DROP = false, bool PAD = false>
__device__ __forceinline__ void forward_loop(TState<NB>& st, cf* lds, const RowBuf& xb,
                                             const DecimArgs& a, int t, int j, int rbeg, int cnt,
                                             int rot, const LnLoad* ln = nullptr, Drop dr = Drop{},
                                             unsigned pj = 0) {
  const Geom& g = a.g;
  const int rend = rbeg + cnt;
  // One tile is prefetched in registers while the previous one is transformed.  (Two tiles ahead
  // was measured slower twice: the memory system is already saturated, deeper queues only add latency.)
  cf nx[16];
  cf ns[LN ? 16 : 1];
  int r = rbeg + rot;
  load_rows<0, 16, PAD>(xb, r, nx);
  if constexpr (LN) load_stats<0, 16>(ln->sb, g, t, r, ns);
  // inter-pass twiddles c^q = w_N^{q (L t + r)} from row (t, r) of the table tq (round 4; rounds 1-3 raised
  // c to its powers per tile: 56 of the loop's 530 vector instructions), one tile ahead like the tile itself
  load_cp(a.tq + ((size_t)t * g.L + r) * 16, st.cp);
  int i = 0;
  for (; i + 1 < cnt; ++i) {
    int rn = r + 1;
    if (rn == rend) rn = rbeg;
    forward_tile<NB, LN, DROP, PAD, false>(st, lds + (i & 1) * EX, xb, a, t, j, r, rn, nx, ns, ln, dr, pj);
    r = rn;
  }
  forward_tile<NB, LN, DROP, PAD, true>(st, lds + (i & 1) * EX, xb, a, t, j, r, r, nx, ns, ln, dr, pj);
}

// One tile of the inverse half.  No vector-memory LOAD may sit between a tile's stores and the next use of loaded
// data: vmcnt counts loads and stores in one in-order queue, so waiting for such a load means waiting for every
// store issued before it (rounds 1-3 read the per-residue twiddles with per-lane loads at the top of each
// iteration and so drained the previous tile's 16 stores before computing anything).  The per-residue twiddles now
// come through the scalar cache, and the next tile's c^q row is requested BEFORE this tile's stores (st.cp is dead
// once the gather has multiplied by it).
// RES: add the rows of `res` (the block input x, same addressing as the output) before the store;
// the rows of the next tile are fetched while the current one is transformed.  (Walking the residues
// in reverse with cacheable loads, so that the second read of x would hit the 256 MiB Infinity
// Cache, was measured: no gain inside a fwd+bwd sequence -- tools/probe_mall.hip shows re-reads at
// HBM rate whatever the footprint.)
// DROP (forward launches): dropout of the tile before the residual add and the store.
template <int NB, bool RES, bool DROP, bool PAD, bool LAST>
__device__ __forceinline__ void inverse_tile(TState<NB>& st, cf* E, const RowBuf& yb, const DecimArgs& a,
                                             int t, int j, int r, int rn,
                                             const RowBuf& res, cf (&rx)[RES ? 16 : 1], Drop dr,
                                             unsigned pj) {
  const Geom& g = a.g;
  inv_phase1<NB, true>(st, a.bt + (size_t)r * BT_STRIDE, E, t, j);
  __syncthreads();
  inv_phase2_gather<NB>(st, E, t, j);
  if constexpr (!LAST) {
    // pinned between the last use of st.cp and the stores: hoisted above the gather the loads need registers of
    // their own and copies at the loop's end -- behind the stores, i.e. a wait for them
    __builtin_amdgcn_sched_barrier(0);
    load_cp(a.tq + ((size_t)t * g.L + rn) * 16, st.cp);
    __builtin_amdgcn_sched_barrier(0);
  }
  fft16<+1>(st.v);
  if constexpr (DROP) {
    const unsigned hd = (unsigned)(g.D >> 1), pstride = 16u * (unsigned)g.L * hd;
    const unsigned p0 = ((unsigned)t * (unsigned)g.L + (unsigned)r) * hd + pj;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      st.v[u] = drop_apply(st.v[u], drop_hash(p0 + (unsigned)u * pstride, dr.key), dr.thr, dr.scale);
  }
  if constexpr (RES) {
#pragma unroll
    for (int u = 0; u < 16; ++u) st.v[u] = cadd(st.v[u], rx[u]);
    if constexpr (!LAST) load_rows<0, 16, PAD>(res, rn, rx);
  }
  store_rows<PAD>(yb, r, st.v, a.st_plain);
}

template <int NB, bool RES = false, bool DROP = false, bool PAD = false>
__device__ __forceinline__ void inverse_loop(TState<NB>& st, cf* lds, const RowBuf& yb,
                                             const DecimArgs& a, int t, int j, int rbeg,
                                             int cnt, int rot, const RowBuf& res, Drop dr = Drop{},
                                             unsigned pj = 0) {
  const Geom& g = a.g;
  int r = rbeg + rot;
  cf rx[RES ? 16 : 1];
  if constexpr (RES) load_rows<0, 16, PAD>(res, r, rx);
  load_cp(a.tq + ((size_t)t * g.L + r) * 16, st.cp);
  // The first row is waited for HERE, so that the loop is entered with nothing in flight: the compiler's wait counts
  // inside the loop are the stricter of "entered from above" and "came round the back edge", and with these eight
  // loads pending on entry every iteration would wait vmcnt(0) for its twiddles -- i.e. for the previous tile's stores
  // -- instead of vmcnt(16).  (An L2 hit, once per launch.)
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), gfx9 encoding
  int i = 0;
  for (; i + 1 < cnt; ++i) {
    int rn = r + 1;
    if (rn == rbeg + cnt) rn = rbeg;
    inverse_tile<NB, RES, DROP, PAD, false>(st, lds + (i & 1) * EX, yb, a, t, j, r, rn, res, rx, dr, pj);
    r = rn;
  }
  inverse_tile<NB, RES, DROP, PAD, true>(st, lds + (i & 1) * EX, yb, a, t, j, r, r, res, rx, dr, pj);
}

__device__ __forceinline__ Drop make_drop(const DecimArgs& a, int b) {
  Drop dr;
  dr.thr = a.drop_thr; dr.scale = a.drop_scale;
  dr.key = drop_row_key(a.rng[0], a.rng[1], b);
  return dr;
}

// ---- parameter gradients inside the backward launch ---------------------------------------------
// Workgroups appended behind the transform workgroups of the launch (DecimArgs::n_cons).  Block cb: d-tile
// cb % ndt, bins [GWT_BINS * (cb / ndt), + GWT_BINS) -- the last block row is grad_bias.  Same arithmetic, in the
// same order, as k_gradw (smx_direct.hip): every thread sums a contiguous run of batch rows, the 8 partial sums
// are added in group order through LDS -> bitwise equal to the separate launch, bitwise reproducible.
// Synchronisation: one FLAG word per transform workgroup, sync[dt * B + b] (no read-modify-write: 64 increments
// of one counter from eight XCDs at the end of the launch cost 30 us), set after that workgroup's slab rows have
// landed.  Coherence without cache-wide operations: the producers write slab rows and bias partials through to
// the level every XCD sees (st4_agent / st1_agent), wait for those stores (s_waitcnt vmcnt(0)) and only then
// raise their flag; the readers use agent-scope loads, which do not trust this XCD's L2 either.
// The area is zero between launches: the last reduction workgroup to finish clears it (see sync_words()).
__device__ __forceinline__ cf ld_agent(const cf* p) {
  union { unsigned long long u; float f[2]; } v;
  v.u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return mk(v.f[0], v.f[1]);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void gradw_tail(const DecimArgs& a, cf* lds, int cb) {
  const Geom& g = a.g;
  const int tid = threadIdx.x, tx = tid & 31, grp = tid >> 5;
  const int ndt = (g.D + DT - 1) / DT;
  const int dt = cb % ndt, fb = cb / ndt, nfb = (g.F + GWT_BINS - 1) / GWT_BINS;
  const int d = dt * DT + tx;
  if (tid < 64) {                      // first wave: all B flags of this d-tile (relaxed polls: an acquire here
    unsigned it = 0;                   // would invalidate this XCD's L2 once per poll, under the workgroups still streaming)
    const unsigned* fl = a.sync + (size_t)dt * g.B;
    for (;;) {
      int ok = 1;
      for (int i = tid; i < g.B; i += 64)
        ok &= __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(4);
      if (++it > (1u << 26)) __builtin_trap();      // seconds: the producers are gone -- fail loudly, never hang
    }
  }
  __syncthreads();
  const int per = (g.B + 7) / 8;
  const int b0 = grp * per, b1 = min(g.B, b0 + per);
  float* pre = reinterpret_cast<float*>(lds);              // [GWT_BINS][8][32] re, then the same for im
  float* pim = pre + GWT_BINS * 8 * 32;
  const bool bias_row = fb == nfb;
  const int f0 = fb * GWT_BINS;
  float re[GWT_BINS], im[GWT_BINS];
#pragma unroll
  for (int i = 0; i < GWT_BINS; ++i) { re[i] = 0.f; im[i] = 0.f; }
  if (d < g.D) {
    if (bias_row) {
      const float* gp = a.fa.gb_part + d;
      int b = b0;
      for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld_agent(gp + (size_t)(b + u) * g.D);
#pragma unroll
        for (int u = 0; u < 8; ++u) re[0] += v[u];
      }
      for (; b < b1; ++b) re[0] += ld_agent(gp + (size_t)b * g.D);
    } else {
      const cf* sp = reinterpret_cast<const cf*>(a.fa.pslab) + (size_t)f0 * g.D + d;
      const size_t bs = (size_t)g.k * g.D;
      const int nb = min(GWT_BINS, g.k - f0);             // bins of this block that exist (<= 0: all zero)
      int b = b0;
      constexpr int RB = 32 / GWT_BINS;                    // rows x bins = 32 loads in flight
      for (; b + RB <= b1; b += RB) {
        cf v[RB][GWT_BINS];
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
          for (int i = 0; i < GWT_BINS; ++i)
            v[u][i] = i < nb ? ld_agent(sp + (size_t)(b + u) * bs + (size_t)i * g.D) : mk(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
          for (int i = 0; i < GWT_BINS; ++i) { re[i] += v[u][i].x; im[i] += v[u][i].y; }
      }
      for (; b < b1; ++b)
#pragma unroll
        for (int i = 0; i < GWT_BINS; ++i)
          if (i < nb) { const cf v = ld_agent(sp + (size_t)b * bs + (size_t)i * g.D); re[i] += v.x; im[i] += v.y; }
    }
  }
#pragma unroll
  for (int i = 0; i < GWT_BINS; ++i) { pre[(i * 8 + grp) * 32 + tx] = re[i]; pim[(i * 8 + grp) * 32 + tx] = im[i]; }
  __syncthreads();
  if (d < g.D && grp < GWT_BINS) {                         // thread (tx, grp) finishes bin f0 + grp
    const int i = grp, f = f0 + i;
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) { sr += pre[(i * 8 + g2) * 32 + tx]; si += pim[(i * 8 + g2) * 32 + tx]; }
    if (bias_row) {
      if (i == 0) a.gbias[d] = sr;
    } else if (f < g.F) {
      a.gw_re[(size_t)d * g.F + f] = sr;
      a.gw_im[(size_t)d * g.F + f] = -si;
    }
  }
  // the last reduction workgroup to get here (all of them are past their wait) leaves the area zero again
  __shared__ unsigned s_last;
  if (tid == 0)
    s_last = __hip_atomic_fetch_add(a.sync + (size_t)g.B * ndt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_last == (unsigned)a.n_cons - 1u)
    for (int i = tid; i <= g.B * ndt; i += TPB)
      __hip_atomic_store(a.sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fused: one launch per direction ----------------------------------------------------------
// ACC (band groups after the first, k > 512): the launch adds its bins' contribution to what the
// earlier groups stored, read-modify-write per tile by the workgroup that owns it.
// PAD: x / y hold R < N rows (zero-padded transform, cropped output).
template <int NB, int MODE, bool DROP = false, bool ACC = false, bool PAD = false>
__global__ __launch_bounds__(TPB, 2) void k_fused(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  if constexpr (MODE == 1 && !ACC && NB <= 2) {     // (four bands: no register room, smx_api keeps k_gradw)
    if (a.n_cons > 0 && a.bid0 + (int)blockIdx.x >= g.B * ndt) {      // appended reduction workgroup
      gradw_tail(a, lds, a.bid0 + (int)blockIdx.x - g.B * ndt);
      return;
    }
  }
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, g.L, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j, rot = w.rot;
  const bool valid = d < g.D;
  const RowBuf xb = row_buf(a.in + (size_t)b * g.R * g.D, g, t, valid ? d : g.D - 2);

  if constexpr (MODE == 0 && !ACC) {       // forward: leave the sync area of this workspace zero for the backward
    if (a.sync != nullptr && a.bid0 + (int)blockIdx.x == 0)       // call that trusts it (SMX_PHASE_SYNC_CLEAN)
      for (int i = tid; i <= g.B * ndt; i += TPB) a.sync[i] = 0u;
  }
  TState<NB> st;
  zero_acc<NB>(st);
  if constexpr (NB == 1) prefetch_io<NB, MODE>(st, g, a.fa, b, d, valid, t);   // saved spectrum, see smx_core.h
  WPre wp;
  constexpr bool STAGE_W = NB == 1 && MODE != 2;           // filter slice via LDS, see prefetch_w
  if constexpr (STAGE_W) prefetch_w(wp, g, a.fa.w_re, a.fa.w_im, w.dt * DT, tid);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  const unsigned pj = (unsigned)((valid ? d : g.D - 2) >> 1);
  // SMX_AB_*: timing ablations for tools/ab.sh (wrong results by design; never defined in the shipped build)
#ifndef SMX_AB_NO_FWD
  forward_loop<NB, false, DROP && MODE == 1, PAD>(st, lds, xb, a, t, j, 0, g.L, rot, nullptr, dr, pj);
#endif
#ifndef SMX_AB_NO_UNPACK
  unpack_filter<NB, MODE, false>(st, lds, g, a.fa, b, d, valid, t, j, STAGE_W ? &wp : nullptr);
#endif
#ifdef SMX_AB_NO_INV
  if (st.acc[0].x == 12345.f) a.out[0] = st.acc[1].y;     // keeps the first half alive
  return;
#endif
  if (a.out == nullptr) {
    if constexpr (NB == 1) store_io<NB, MODE>(st, g, a.fa, b, d, valid, t);
    // phase-split backward: park the filtered spectrum for k_split_b (same layout as k_split_f)
    if (a.ws_s != nullptr) {
      cf* s = a.ws_s + (size_t)(b * ndt + w.dt) * (16 * NB * TPB);
#pragma unroll
      for (int sl = 0; sl < 16 * NB; ++sl) s[sl * TPB + tid] = st.acc[sl];
    }
    return;
  }
  __syncthreads();
  const RowBuf yb = row_buf(a.out + (size_t)b * g.R * g.D, g, t, d, valid);   // (ACC: also read back)
  inverse_loop<NB, ACC, DROP && MODE == 0, PAD>(st, lds, yb, a, t, j, 0, g.L, rot, yb, dr, pj);
  if constexpr (NB == 1) store_io<NB, MODE>(st, g, a.fa, b, d, valid, t);     // saved spectrum / grad slab
  if constexpr (MODE == 1 && !ACC && NB <= 2) {     // (four bands: no register room, smx_api keeps k_gradw)
    if (a.n_cons > 0) {        // tell the appended reduction workgroups that this (b, d-tile)'s slab rows are out
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have landed
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(a.sync + (size_t)w.dt * g.B + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- sixteen-row decimation: N = 16 P, any P (smx_core.h) ------------------------------------------------
// The loops walk tiles of 16 residues [tbeg, tbeg + cnt), starting at tbeg + rot.  DROP: the fused dropout of the
// 256-point kernels -- the mask of element pair (row n, channel pair) is a hash of (n D/2 + d/2, key of the batch
// row); forward launches apply it to the stored tile, backward launches to the loaded g.
template <int NB, bool PAD, bool DROP>
__device__ __forceinline__ void forward16_loop(TState<NB>& st, cf* lds, const float* __restrict__ xb,
                                               const DecimArgs& a, int t, int j, int tbeg, int cnt, int rot,
                                               Drop dr, unsigned pj) {
  const Geom& g = a.g;
  const unsigned hd = (unsigned)(g.D >> 1);
  cf nx[16];
  int tau = tbeg + rot;
  load_tile16<PAD>(xb, g, t, tau, nx);
  cf cn = a.tw[min(16 * tau + t, g.N - 1)];
  for (int i = 0; i < cnt; ++i) {
    if constexpr (DROP) {
      const unsigned p0 = (unsigned)(16 * tau + t) * hd + pj, ps = (unsigned)g.P * hd;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        st.v[u] = drop_apply(nx[u], drop_hash(p0 + (unsigned)u * ps, dr.key), dr.thr, dr.scale);
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) st.v[u] = nx[u];
    }
    const cf c = cn;
    int tn = tau + 1;
    if (tn == tbeg + cnt) tn = tbeg;
    if (i + 1 < cnt) {
      load_tile16<PAD>(xb, g, t, tn, nx);
      cn = a.tw[min(16 * tn + t, g.N - 1)];
    }
    cf* E = lds + (i & 1) * EX;
    fwd_phase1<NB>(st, c, E, t, j);                       // fft16 over the 16 rows, times w_N^{q r}, scatter
    __syncthreads();
    fwd16_phase2<NB>(st, E, a.v16, a.b16 + (size_t)tau * 32, t, j);
    tau = tn;
  }
}
template <int NB, bool PAD, bool DROP>
__device__ __forceinline__ void inverse16_loop(TState<NB>& st, cf* lds, float* __restrict__ yb, const DecimArgs& a,
                                               int t, int j, bool valid, int tbeg, int cnt, int rot, Drop dr,
                                               unsigned pj) {
  const Geom& g = a.g;
  const unsigned hd = (unsigned)(g.D >> 1);
  int tau = tbeg + rot;
  for (int i = 0; i < cnt; ++i) {
    const cf c = a.tw[min(16 * tau + t, g.N - 1)];
    cf* E = lds + (i & 1) * EX;
    inv16_phase1<NB>(st, a.v16, a.b16 + (size_t)tau * 32, E, t, j);
    __syncthreads();
    inv_phase2<NB>(st, c, E, t, j);                       // gather, times w_N^{-q r}, inverse fft16 -> the 16 rows
    if constexpr (DROP) {
      const unsigned p0 = (unsigned)(16 * tau + t) * hd + pj, ps = (unsigned)g.P * hd;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        st.v[u] = drop_apply(st.v[u], drop_hash(p0 + (unsigned)u * ps, dr.key), dr.thr, dr.scale);
    }
    store_tile16<PAD>(yb, g, t, tau, valid, st.v);
    ++tau;
    if (tau == tbeg + cnt) tau = tbeg;
  }
}

// One launch per direction like k_fused<NB, MODE>.  NB = 1: k <= 128, the filter slice staged through LDS;
// NB = 2: k <= 256, the filter from its packed copy (fa.wt) or gathered.
template <int NB, int MODE, bool PAD = false, bool DROP = false>
__global__ __launch_bounds__(TPB, 2) void k_fused16(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const int T = g.L;                                      // tiles
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, T, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j, rot = w.rot;
  const bool valid = d < g.D;
  const float* xb = a.in + (size_t)b * g.R * g.D + (valid ? d : g.D - 2);
  TState<NB> st;
  zero_acc<NB>(st);
  if constexpr (NB == 1) prefetch_io<NB, MODE>(st, g, a.fa, b, d, valid, t);
  WPre wp;
  constexpr bool STAGE_W = NB == 1 && MODE != 2;
  if constexpr (STAGE_W) prefetch_w(wp, g, a.fa.w_re, a.fa.w_im, w.dt * DT, tid);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  const unsigned pj = (unsigned)((valid ? d : g.D - 2) >> 1);
  forward16_loop<NB, PAD, DROP && MODE == 1>(st, lds, xb, a, t, j, 0, T, rot, dr, pj);
  unpack_filter<NB, MODE, false>(st, lds, g, a.fa, b, d, valid, t, j, STAGE_W ? &wp : nullptr);
  if (a.out == nullptr) {
    if constexpr (NB == 1) store_io<NB, MODE>(st, g, a.fa, b, d, valid, t);
    if (a.ws_s != nullptr) {          // phase-split backward: park the filtered spectrum for k_split16_b
      cf* sp = a.ws_s + (size_t)(b * ndt + w.dt) * (16 * NB * TPB);
#pragma unroll
      for (int sl = 0; sl < 16 * NB; ++sl) sp[sl * TPB + tid] = st.acc[sl];
    }
    return;
  }
  __syncthreads();
  inverse16_loop<NB, PAD, DROP && MODE == 0>(st, lds, a.out + (size_t)b * g.R * g.D + d, a, t, j, valid, 0, T, rot, dr,
                                             pj);
  if constexpr (NB == 1) store_io<NB, MODE>(st, g, a.fa, b, d, valid, t);
}

// Few (batch row, d-tile) pairs: the tiles are cut into nsplit chunks as on the 256-point split plan -- (A) partial
// spectra per chunk, then the SAME k_split_sum / k_split_f (the accumulator layout is the same), then (B) the inverse
// per chunk.  (B) with nsplit = 1 is also the inverse half of a phase-split backward.
template <int NB, bool PAD = false, bool DROP = false>
__global__ __launch_bounds__(TPB, 2) void k_split16_a(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int tbeg = c * a.lc, cnt = min(a.lc, g.L - tbeg);
  const float* xb = a.in + (size_t)b * g.R * g.D + (valid ? d : g.D - 2);
  TState<NB> st;
  zero_acc<NB>(st);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  if (cnt > 0)
    forward16_loop<NB, PAD, DROP>(st, lds, xb, a, t, j, tbeg, cnt, w.rot % cnt, dr,
                                  (unsigned)((valid ? d : g.D - 2) >> 1));
  cf* z = a.ws_z + ((size_t)wg * a.nsplit + c) * (16 * NB * TPB);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl) z[sl * TPB + tid] = st.acc[sl];
}
template <int NB, bool PAD = false, bool DROP = false>
__global__ __launch_bounds__(TPB, 2) void k_split16_b(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int tbeg = c * a.lc, cnt = min(a.lc, g.L - tbeg);
  if (cnt <= 0) return;
  TState<NB> st;
  const cf* sp = a.ws_s + (size_t)wg * (16 * NB * TPB);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl) st.acc[sl] = sp[sl * TPB + tid];
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  inverse16_loop<NB, PAD, DROP>(st, lds, a.out + (size_t)b * g.R * g.D + d, a, t, j, valid, tbeg, cnt, w.rot % cnt, dr,
                                (unsigned)((valid ? d : g.D - 2) >> 1));
}

// ---- synthesis from a given one-sided spectrum (smx_irfft_ex): the inverse half alone ---------------
// The accumulators are filled from the rows of the (B,k,D) spectrum (synth_fill: no exchange, no filter) and
// the inverse loop runs as in k_fused; with out == NULL they are parked for k_split_b (residue-split plans).
template <int NB, bool PAD>
__global__ __launch_bounds__(TPB, 2) void k_synth(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, g.L, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j, rot = w.rot;
  const bool valid = d < g.D;
  TState<NB> st;
  synth_fill<NB>(st, g, a.fa, b, d, valid, t);
  if (a.out == nullptr) {
    cf* s = a.ws_s + (size_t)(b * ndt + w.dt) * (16 * NB * TPB);
#pragma unroll
    for (int sl = 0; sl < 16 * NB; ++sl) s[sl * TPB + tid] = st.acc[sl];
    return;
  }
  const RowBuf yb = row_buf(a.out + (size_t)b * g.R * g.D, g, t, d, valid);
  inverse_loop<NB, false, false, PAD>(st, lds, yb, a, t, j, 0, g.L, rot, yb);
}

// ---- full spectrum at N = 2048: eight bands, one launch per direction ------------------------------
// The whole packed spectrum of the workgroup's 16 channel pairs lives in registers (128 complex per
// thread, partly in AGPRs: one workgroup per CU).  Eight tiles with compile-time residues: each tile's
// 256-point spectrum is kept apart (fwd_phase2_store), an 8-point transform across the residues turns
// them into the eight bands, the unpack runs in four LDS rounds, and the inverse mirrors it.  Replaces
// two band-group launches per direction + three edge-bin passes (each group re-reading x and
// re-writing y) for the reference's default causal-convolution lengths (seq_len 1024 + kernel 128 ->
// n_fft 2048, fft_lm/train_fixed_full.py:507-519).
template <int R, int MODE, bool PAD>
__device__ __forceinline__ void full8_fwd_tiles(TState<8>& st, cf* lds, const float* __restrict__ xb,
                                                const DecimArgs& a, int t, int j, cf (&nx)[16], cf& cn) {
  if constexpr (R < 8) {
    const Geom& g = a.g;
#pragma unroll
    for (int u = 0; u < 16; ++u) st.v[u] = nx[u];
    const cf c = cn;
    if constexpr (R + 1 < 8) {
      load_part_tile<0, 8, PAD>(xb, g, t, R + 1, nx);
      cn = a.tw[(size_t)t * 8 + R + 1];
    }
    cf* E = lds + (R & 1) * EX;
    fwd_phase1<8>(st, c, E, t, j);
    __syncthreads();
    if constexpr (R + 1 < 8) load_part_tile<8, 8, PAD>(xb, g, t, R + 1, nx);
    fwd_phase2_store<8, R>(st, E, a.bt + (size_t)R * BT_STRIDE, t, j);
    full8_fwd_tiles<R + 1, MODE, PAD>(st, lds, xb, a, t, j, nx, cn);
  }
}
template <int R, bool PAD>
__device__ __forceinline__ void full8_inv_tiles(TState<8>& st, cf* lds, float* __restrict__ yb,
                                                const DecimArgs& a, int t, int j, bool valid) {
  if constexpr (R < 8) {
    const cf c = a.tw[(size_t)t * 8 + R];
    cf* E = lds + (R & 1) * EX;
    inv_phase1_from<8, R>(st, a.bt + (size_t)R * BT_STRIDE, E, t, j);
    __syncthreads();
    inv_phase2<8>(st, c, E, t, j);
    store_tile<PAD>(yb, a.g, t, R, valid, st.v);
    full8_inv_tiles<R + 1, PAD>(st, lds, yb, a, t, j, valid);
  }
}

template <int MODE, bool PAD>
__global__ __launch_bounds__(TPB, 1) void k_full8(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, g.L, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const float* xb = a.in + (size_t)b * g.R * g.D + (valid ? d : g.D - 2);

  TState<8> st;
  cf nx[16];
  load_tile<PAD>(xb, g, t, 0, nx);
  cf cn = a.tw[(size_t)t * 8];
  full8_fwd_tiles<0, MODE, PAD>(st, lds, xb, a, t, j, nx, cn);
  residue_fft8<-1>(st);
#ifndef SMX_FULL8_BATCHED
#define SMX_FULL8_BATCHED false
#endif
  // (the batched unpack of k_split_f spills ~1700 registers here; the slot loop prefetches instead)
  unpack_filter<8, MODE, SMX_FULL8_BATCHED>(st, lds, g, a.fa, b, d, valid, t, j);
  if (a.out == nullptr) return;                      // spectrum only / parameter gradients only
  __syncthreads();
  residue_fft8<+1>(st);
  full8_inv_tiles<0, PAD>(st, lds, a.out + (size_t)b * g.R * g.D + d, a, t, j, valid);
}

// synthesis at N = 2048 from a given one-sided spectrum: the eight bands filled from its rows (synth_fill<8>),
// bands -> residues, inverse tiles -- one launch, the spectrum read once and y written once
template <bool PAD>
__global__ __launch_bounds__(TPB, 1) void k_synth8(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, g.L, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  TState<8> st;
  synth_fill<8>(st, g, a.fa, b, d, valid, t);
  residue_fft8<+1>(st);
  full8_inv_tiles<0, PAD>(st, lds, a.out + (size_t)b * g.R * g.D + d, a, t, j, valid);
}

// ---- fused block: y = x + mix(LayerNorm(x)) in one launch (reference spectral_layers.py:185) ------
// Same structure as k_fused<NB, 0>; x is read a second time at the store for the residual.
// (four bands: 256 VGPRs are not enough for the extra row statistics and residual rows -- 57 spills
// inside the loops cost more than the second workgroup per CU gains: 784 vs 688 us at (32,4096,1024))
template <int NB, bool DROP = false>
__global__ __launch_bounds__(TPB, NB > 2 ? 1 : 2) void k_fused_blk(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, 1, g.L, a.placement);
  const int b = w.b, d = w.dt * DT + 2 * j, rot = w.rot;
  const bool valid = d < g.D;
  const int dc = valid ? d : g.D - 2;
  const RowBuf xb = row_buf(a.in + (size_t)b * g.R * g.D, g, t, dc);
  LnLoad ln;
  ln.sb = a.ln_stats + (size_t)b * g.N;
  ln.g0 = a.ln_w ? a.ln_w[dc] : 1.f; ln.g1 = a.ln_w ? a.ln_w[dc + 1] : 1.f;
  ln.b0 = a.ln_b ? a.ln_b[dc] : 0.f; ln.b1 = a.ln_b ? a.ln_b[dc + 1] : 0.f;

  if (a.sync != nullptr && a.bid0 + (int)blockIdx.x == 0)          // as k_fused<NB, 0>: sync area left zero
    for (int i = tid; i <= g.B * ndt; i += TPB) a.sync[i] = 0u;
  TState<NB> st;
  zero_acc<NB>(st);
  WPre wp;
  if constexpr (NB == 1) prefetch_w(wp, g, a.fa.w_re, a.fa.w_im, w.dt * DT, tid);
  forward_loop<NB, true>(st, lds, xb, a, t, j, 0, g.L, rot, &ln);
  unpack_filter<NB, 0, false>(st, lds, g, a.fa, b, d, valid, t, j, NB == 1 ? &wp : nullptr);
  __syncthreads();
  const RowBuf yb = row_buf(a.out + (size_t)b * g.R * g.D, g, t, d, valid);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  inverse_loop<NB, true, DROP>(st, lds, yb, a, t, j, 0, g.L, rot, xb, dr, (unsigned)(dc >> 1));
  if constexpr (NB == 1) store_io<NB, 0>(st, g, a.fa, b, d, valid, t);
}

// ---- split path: (A) partial forward over a chunk of residues ---------------------------------
template <int NB, bool DROP = false, bool PAD = false>
__global__ __launch_bounds__(TPB, 2) void k_split_a(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int rbeg = c * a.lc, cnt = min(a.lc, g.L - rbeg);
  const RowBuf xb = row_buf(a.in + (size_t)b * g.R * g.D, g, t, valid ? d : g.D - 2);
  const int rot = w.rot % cnt;

  TState<NB> st;
  zero_acc<NB>(st);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  forward_loop<NB, false, DROP, PAD>(st, lds, xb, a, t, j, rbeg, cnt, rot, nullptr, dr,
                                     (unsigned)((valid ? d : g.D - 2) >> 1));
  cf* z = a.ws_z + ((size_t)wg * a.nsplit + c) * (16 * NB * TPB);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl) z[sl * TPB + tid] = st.acc[sl];
}

// (S) sum the nsplit partial spectra of every (b, d-tile): one block per (workgroup item, slot), so
// the whole chip takes part (a single CU streams only ~25 GB/s; with B*ndt blocks this step cost 25 us
// at C3).  Chunks are added in index order -> bitwise reproducible.
template <int NB>
__global__ __launch_bounds__(TPB) void k_split_sum(const DecimArgs a) {
  const int sl = blockIdx.x % (16 * NB), wg = blockIdx.x / (16 * NB), tid = threadIdx.x;
  const cf* z = a.ws_z + (size_t)wg * a.nsplit * (16 * NB * TPB) + sl * TPB + tid;
  cf acc = mk(0.f, 0.f);
  int c = 0;
  for (; c + 8 <= a.nsplit; c += 8) {
    cf v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = z[(size_t)(c + u) * (16 * NB * TPB)];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = cadd(acc, v[u]);
  }
  for (; c < a.nsplit; ++c) acc = cadd(acc, z[(size_t)c * (16 * NB * TPB)]);
  a.ws_zs[(size_t)wg * (16 * NB * TPB) + sl * TPB + tid] = acc;
}

// (F) unpack + filter the summed spectrum; emits S for (B) and the saved spectrum / grad slab
// (only B * ceil(D/32) < 384 workgroups exist on this plan, so one workgroup per CU costs nothing and
// lets the batched unpack of the multi-band variants keep its loads in registers instead of scratch)
template <int NB, int MODE>
__global__ __launch_bounds__(TPB, NB > 1 ? 1 : 2) void k_split_f(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const int wg = blockIdx.x, b = wg / ndt, d = (wg % ndt) * DT + 2 * j;
  const bool valid = d < g.D;
  TState<NB> st;
  // one band: this workgroup's 32-channel slice of (D, F) goes through LDS as in k_fused<1, .> (prefetch_w /
  // stage_w) -- no packed copy of the filter, no k_pack_w launch on the residue-split plan either
  WPre wp;
  constexpr bool STAGE_W = NB == 1 && MODE != 2;
  if constexpr (STAGE_W) prefetch_w(wp, g, a.fa.w_re, a.fa.w_im, (wg % ndt) * DT, tid);
  if (a.sum_in_f) {
    // the chunk partials of this (b, d-tile) summed here, in chunk order like k_split_sum (bitwise the same sums):
    // four slots x up to eight chunks = 32 loads in flight per thread.  One launch and one kernel boundary less per
    // direction; the 64 workgroups of C3 read 16.8 MB instead of 2.1, which costs less than the boundary did.
    const cf* z = a.ws_z + (size_t)wg * a.nsplit * (16 * NB * TPB) + tid;
    const size_t cs = (size_t)16 * NB * TPB;
#pragma unroll
    for (int s0 = 0; s0 < 16 * NB; s0 += 4) {
      cf sum[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sum[i] = mk(0.f, 0.f);
      int c = 0;
      for (; c + 8 <= a.nsplit; c += 8) {
        cf v[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 8; ++u) v[i][u] = z[(size_t)(c + u) * cs + (s0 + i) * TPB];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 8; ++u) sum[i] = cadd(sum[i], v[i][u]);
      }
      for (; c < a.nsplit; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] = cadd(sum[i], z[(size_t)c * cs + (s0 + i) * TPB]);
#pragma unroll
      for (int i = 0; i < 4; ++i) st.acc[s0 + i] = sum[i];
    }
  } else {
    const cf* z = a.ws_zs + (size_t)wg * (16 * NB * TPB);
#pragma unroll
    for (int sl = 0; sl < 16 * NB; ++sl) st.acc[sl] = z[sl * TPB + tid];
  }
  unpack_filter<NB, MODE, true>(st, lds, g, a.fa, b, d, valid, t, j, STAGE_W ? &wp : nullptr);
  if (a.ws_s == nullptr) return;
  cf* s = a.ws_s + (size_t)wg * (16 * NB * TPB);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl) s[sl * TPB + tid] = st.acc[sl];
}

// (B) inverse over a chunk of residues, from the filtered spectrum parked by k_split_f / k_fused.
// (Folding the unpack + filter into this launch was measured: every chunk workgroup repeating the
// latency-bound prologue cost 33 us at C3, against 15 us for the separate B*ndt-block launch.)
template <int NB, bool DROP = false, bool ACC = false, bool PAD = false>
__global__ __launch_bounds__(TPB, 2) void k_split_b(const DecimArgs a) {
  SMX_LDS_DECL;
  const Geom& g = a.g;
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  const int ndt = (g.D + DT - 1) / DT;
  const WgItem w = wg_map(a.bid0 + blockIdx.x, g.B, ndt, a.nsplit, a.lc, a.placement);
  const int c = w.c, b = w.b, wg = b * ndt + w.dt, d = w.dt * DT + 2 * j;
  const bool valid = d < g.D;
  const int rbeg = c * a.lc, cnt = min(a.lc, g.L - rbeg);
  const int rot = w.rot % cnt;
  TState<NB> st;
  const cf* s = a.ws_s + (size_t)wg * (16 * NB * TPB);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl) st.acc[sl] = s[sl * TPB + tid];
  const RowBuf yb = row_buf(a.out + (size_t)b * g.R * g.D, g, t, d, valid);
  Drop dr{};
  if constexpr (DROP) dr = make_drop(a, b);
  inverse_loop<NB, ACC, DROP, PAD>(st, lds, yb, a, t, j, rbeg, cnt, rot, yb, dr,
                                   (unsigned)((valid ? d : g.D - 2) >> 1));
}

// ---- launchers ---------------------------------------------------------------------------------
#ifdef SMX_MINI
// development only (never linked): the BASELINE kernels alone, for a quick look at their ISA --
//   hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -DSMX_MINI smx_decim.hip
template __global__ void k_fused<1, 0>(const DecimArgs);
template __global__ void k_fused<1, 1>(const DecimArgs);
template __global__ void k_fused<2, 0>(const DecimArgs);
template __global__ void k_fused<2, 1>(const DecimArgs);
template __global__ void k_split_a<1>(const DecimArgs);
template __global__ void k_split_b<1>(const DecimArgs);
template __global__ void k_split16_b<1>(const DecimArgs);
template __global__ void k_fused16<1, 0>(const DecimArgs);
#else
// four bands, accumulating store (band groups after the first)
static void launch_fused_acc(const DecimArgs& a, int mode, dim3 grid, hipStream_t s) {
  const bool pad = a.g.R < a.g.N;
  if (mode == 0 && pad) hipLaunchKernelGGL((k_fused<4, 0, false, true, true>), grid, dim3(TPB), 0, s, a);
  else if (mode == 0) hipLaunchKernelGGL((k_fused<4, 0, false, true>), grid, dim3(TPB), 0, s, a);
  else if (pad) hipLaunchKernelGGL((k_fused<4, 1, false, true, true>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fused<4, 1, false, true>), grid, dim3(TPB), 0, s, a);
}

template <int NB>
static void launch_fused_t(const DecimArgs& a, int mode, dim3 grid, hipStream_t s) {
  const bool drop = a.drop_thr != 0;     // mode 0: on the stored tile, mode 1: on the loaded tile
  if (a.g.R < a.g.N) {                   // zero-padded rows (never together with dropout, smx_api)
    if (mode == 0) hipLaunchKernelGGL((k_fused<NB, 0, false, false, true>), grid, dim3(TPB), 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL((k_fused<NB, 1, false, false, true>), grid, dim3(TPB), 0, s, a);
    else hipLaunchKernelGGL((k_fused<NB, 2, false, false, true>), grid, dim3(TPB), 0, s, a);
    return;
  }
  if (mode == 0 && drop) hipLaunchKernelGGL((k_fused<NB, 0, true>), grid, dim3(TPB), 0, s, a);
  else if (mode == 0) hipLaunchKernelGGL((k_fused<NB, 0>), grid, dim3(TPB), 0, s, a);
  else if (mode == 1 && drop) hipLaunchKernelGGL((k_fused<NB, 1, true>), grid, dim3(TPB), 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL((k_fused<NB, 1>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((k_fused<NB, 2>), grid, dim3(TPB), 0, s, a);
}

int gradw_tail_blocks(int D, int F, bool bias) {
  return ((D + DT - 1) / DT) * ((F + GWT_BINS - 1) / GWT_BINS + (bias ? 1 : 0));
}

hipError_t launch_fused(const DecimArgs& a, int nb, int mode, hipStream_t s) {
  const int total = n_wg(a);
  return for_rounds(a, total, [&](const DecimArgs& r, dim3 grid) {
    // the reduction workgroups ride behind the LAST round's transform workgroups
    if (r.n_cons > 0 && r.bid0 + (int)grid.x >= total) grid.x += r.n_cons;
    if (r.accumulate && r.out != nullptr) launch_fused_acc(r, mode, grid, s);
    else if (nb == 4) launch_fused_t<4>(r, mode, grid, s);
    else if (nb == 2) launch_fused_t<2>(r, mode, grid, s);
    else launch_fused_t<1>(r, mode, grid, s);
  }, nb == 4);
}

template <int NB, bool PAD>
static void launch_fused16_t(const DecimArgs& r, int mode, dim3 grid, hipStream_t s) {
  const dim3 block(TPB);
  if (mode == 0) hipLaunchKernelGGL((k_fused16<NB, 0, PAD>), grid, block, 0, s, r);
  else if (mode == 1) hipLaunchKernelGGL((k_fused16<NB, 1, PAD>), grid, block, 0, s, r);
  else hipLaunchKernelGGL((k_fused16<NB, 2, PAD>), grid, block, 0, s, r);
}
template <int NB>
static void launch_fused16_drop(const DecimArgs& r, int mode, dim3 grid, hipStream_t s) {      // never with padded rows
  const dim3 block(TPB);
  if (mode == 0) hipLaunchKernelGGL((k_fused16<NB, 0, false, true>), grid, block, 0, s, r);
  else hipLaunchKernelGGL((k_fused16<NB, 1, false, true>), grid, block, 0, s, r);
}
hipError_t launch_fused16(const DecimArgs& a, int nb, int mode, hipStream_t s) {
  return for_rounds(a, n_wg(a), [&](const DecimArgs& r, dim3 grid) {
    const bool pad = r.g.R < r.g.N;
    if (r.drop_thr != 0 && mode != 2) {
      if (nb == 2) launch_fused16_drop<2>(r, mode, grid, s); else launch_fused16_drop<1>(r, mode, grid, s);
    } else if (nb == 2) { if (pad) launch_fused16_t<2, true>(r, mode, grid, s); else launch_fused16_t<2, false>(r, mode, grid, s); }
    else if (pad) launch_fused16_t<1, true>(r, mode, grid, s);
    else launch_fused16_t<1, false>(r, mode, grid, s);
  });
}
// the split launches: drop = apply the dropout mask (A: to the loaded tile, B: to the stored tile)
hipError_t launch_split16_a(const DecimArgs& a, int nb, bool drop, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    const bool pad = r.g.R < r.g.N, dr = drop && r.drop_thr != 0;
    if (nb == 2) {
      if (dr) hipLaunchKernelGGL((k_split16_a<2, false, true>), grid, block, 0, s, r);
      else if (pad) hipLaunchKernelGGL((k_split16_a<2, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split16_a<2>), grid, block, 0, s, r);
    } else if (dr) hipLaunchKernelGGL((k_split16_a<1, false, true>), grid, block, 0, s, r);
    else if (pad) hipLaunchKernelGGL((k_split16_a<1, true>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_split16_a<1>), grid, block, 0, s, r);
  });
}
hipError_t launch_split16_b(const DecimArgs& a, int nb, bool drop, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    const bool pad = r.g.R < r.g.N, dr = drop && r.drop_thr != 0;
    if (nb == 2) {
      if (dr) hipLaunchKernelGGL((k_split16_b<2, false, true>), grid, block, 0, s, r);
      else if (pad) hipLaunchKernelGGL((k_split16_b<2, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split16_b<2>), grid, block, 0, s, r);
    } else if (dr) hipLaunchKernelGGL((k_split16_b<1, false, true>), grid, block, 0, s, r);
    else if (pad) hipLaunchKernelGGL((k_split16_b<1, true>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_split16_b<1>), grid, block, 0, s, r);
  });
}

hipError_t launch_synth(const DecimArgs& a, int nb, hipStream_t s) {
  return for_rounds(a, n_wg(a), [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    const bool pad = r.g.R < r.g.N;
    if (nb == 4 && pad) hipLaunchKernelGGL((k_synth<4, true>), grid, block, 0, s, r);
    else if (nb == 4) hipLaunchKernelGGL((k_synth<4, false>), grid, block, 0, s, r);
    else if (nb == 2 && pad) hipLaunchKernelGGL((k_synth<2, true>), grid, block, 0, s, r);
    else if (nb == 2) hipLaunchKernelGGL((k_synth<2, false>), grid, block, 0, s, r);
    else if (pad) hipLaunchKernelGGL((k_synth<1, true>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_synth<1, false>), grid, block, 0, s, r);
  }, nb == 4);
}

hipError_t launch_synth8(const DecimArgs& a, hipStream_t s) {
  DecimArgs r = a;
  r.bid0 = 0;
  if (a.g.R < a.g.N) hipLaunchKernelGGL((k_synth8<true>), dim3(n_wg(a)), dim3(TPB), 0, s, r);
  else hipLaunchKernelGGL((k_synth8<false>), dim3(n_wg(a)), dim3(TPB), 0, s, r);
  return hipGetLastError();
}

hipError_t launch_full8(const DecimArgs& a, int mode, hipStream_t s) {
  DecimArgs r = a;
  r.bid0 = 0;
  const dim3 grid(n_wg(a)), block(TPB);
  const bool pad = a.g.R < a.g.N;
  if (mode == 0 && pad) hipLaunchKernelGGL((k_full8<0, true>), grid, block, 0, s, r);
  else if (mode == 0) hipLaunchKernelGGL((k_full8<0, false>), grid, block, 0, s, r);
  else if (mode == 1 && pad) hipLaunchKernelGGL((k_full8<1, true>), grid, block, 0, s, r);
  else if (mode == 1) hipLaunchKernelGGL((k_full8<1, false>), grid, block, 0, s, r);
  else if (pad) hipLaunchKernelGGL((k_full8<2, true>), grid, block, 0, s, r);
  else hipLaunchKernelGGL((k_full8<2, false>), grid, block, 0, s, r);
  return hipGetLastError();
}

hipError_t launch_fused_block(const DecimArgs& a, int nb, hipStream_t s) {
  return for_rounds(a, n_wg(a), [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    if (r.drop_thr != 0) {
      if (nb == 4) hipLaunchKernelGGL((k_fused_blk<4, true>), grid, block, 0, s, r);
      else if (nb == 2) hipLaunchKernelGGL((k_fused_blk<2, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_fused_blk<1, true>), grid, block, 0, s, r);
    } else if (nb == 4) hipLaunchKernelGGL((k_fused_blk<4>), grid, block, 0, s, r);
    else if (nb == 2) hipLaunchKernelGGL((k_fused_blk<2>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_fused_blk<1>), grid, block, 0, s, r);
  }, nb == 4);
}

hipError_t launch_split_a(const DecimArgs& a, int nb, bool drop_in, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    if (r.g.R < r.g.N) {
      if (nb == 4) hipLaunchKernelGGL((k_split_a<4, false, true>), grid, block, 0, s, r);
      else if (nb == 2) hipLaunchKernelGGL((k_split_a<2, false, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split_a<1, false, true>), grid, block, 0, s, r);
    } else if (drop_in && r.drop_thr != 0) {
      if (nb == 4) hipLaunchKernelGGL((k_split_a<4, true>), grid, block, 0, s, r);
      else if (nb == 2) hipLaunchKernelGGL((k_split_a<2, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split_a<1, true>), grid, block, 0, s, r);
    } else if (nb == 4) hipLaunchKernelGGL((k_split_a<4>), grid, block, 0, s, r);
    else if (nb == 2) hipLaunchKernelGGL((k_split_a<2>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_split_a<1>), grid, block, 0, s, r);
  });
}

hipError_t launch_split_f(const DecimArgs& a, int nb, int mode, hipStream_t s) {
  if (!a.sum_in_f) {
    dim3 gs(n_wg(a) * 16 * nb);
    if (nb == 1) hipLaunchKernelGGL((k_split_sum<1>), gs, dim3(TPB), 0, s, a);
    else if (nb == 2) hipLaunchKernelGGL((k_split_sum<2>), gs, dim3(TPB), 0, s, a);
    else hipLaunchKernelGGL((k_split_sum<4>), gs, dim3(TPB), 0, s, a);
  }
  dim3 grid(n_wg(a)), block(TPB);
  if (nb == 4) {
    if (mode == 0) hipLaunchKernelGGL((k_split_f<4, 0>), grid, block, 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL((k_split_f<4, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_split_f<4, 2>), grid, block, 0, s, a);
    return hipGetLastError();
  }
  if (nb == 1 && mode == 0) hipLaunchKernelGGL((k_split_f<1, 0>), grid, block, 0, s, a);
  else if (nb == 1 && mode == 1) hipLaunchKernelGGL((k_split_f<1, 1>), grid, block, 0, s, a);
  else if (nb == 1) hipLaunchKernelGGL((k_split_f<1, 2>), grid, block, 0, s, a);
  else if (mode == 0) hipLaunchKernelGGL((k_split_f<2, 0>), grid, block, 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL((k_split_f<2, 1>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_split_f<2, 2>), grid, block, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_split_b(const DecimArgs& a, int nb, bool drop_out, hipStream_t s) {
  return for_rounds(a, n_wg(a) * a.nsplit, [&](const DecimArgs& r, dim3 grid) {
    const dim3 block(TPB);
    if (r.g.R < r.g.N) {
      if (r.accumulate) hipLaunchKernelGGL((k_split_b<4, false, true, true>), grid, block, 0, s, r);
      else if (nb == 1) hipLaunchKernelGGL((k_split_b<1, false, false, true>), grid, block, 0, s, r);
      else if (nb == 2) hipLaunchKernelGGL((k_split_b<2, false, false, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split_b<4, false, false, true>), grid, block, 0, s, r);
    } else if (r.accumulate) hipLaunchKernelGGL((k_split_b<4, false, true>), grid, block, 0, s, r);
    else if (drop_out && r.drop_thr != 0) {
      if (nb == 1) hipLaunchKernelGGL((k_split_b<1, true>), grid, block, 0, s, r);
      else if (nb == 2) hipLaunchKernelGGL((k_split_b<2, true>), grid, block, 0, s, r);
      else hipLaunchKernelGGL((k_split_b<4, true>), grid, block, 0, s, r);
    } else if (nb == 1) hipLaunchKernelGGL((k_split_b<1>), grid, block, 0, s, r);
    else if (nb == 2) hipLaunchKernelGGL((k_split_b<2>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((k_split_b<4>), grid, block, 0, s, r);
  });
}

#endif  // SMX_MINI

}  // namespace smx
