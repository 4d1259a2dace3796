// smx_core.h -- arithmetic core of the decimated, output-pruned Stockham transform (gfx950).
//
// Replaces the reference's torch.fft.fft / slice-multiply / zero-fill / torch.fft.ifft sequence
// (reference fft_tensor/spectral_layers.py:88-116) with ONE pass over x and ONE pass over y.
//
// Math (N = 256*L, k <= 128*NB kept bins, two real channels packed as one complex sequence z = a + i b):
//   forward   Z[f] = sum_r w_N^{f r} * DFT256( z[L m + r] )[f mod 256]     for signed f in (-128 NB, 128 NB)
//   unpack    A[f] = (Z[f] + conj Z[-f]) / 2 ,  B[f] = (Z[f] - conj Z[-f]) / (2i)
//   filter    Ya = W_a[f] A[f] ... ; S[+f] = (Ya + i Yb)/(2N), S[-f] = (conj Ya + i conj Yb)/(2N), S[0] = Re/N + bias
//   inverse   z_out[L m + r] = IDFT256( sum_bands S[f] w_N^{-f r} )[m]
// Each 256-point transform is two radix-16 passes held in registers by 16 cooperating threads,
// with one LDS exchange between the passes (Stockham auto-sort: no bit reversal anywhere).
//
// Everything here is __host__ __device__ so tests/emu/ can run a whole workgroup on the CPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SMX_HD __host__ __device__ __forceinline__
#else
#define SMX_HD inline
#ifndef __restrict__
#define __restrict__ __restrict
#endif
struct float2 { float x, y; };
#endif

// cache policy of the streamed tensors (x/g in, y/grad_x out): nontemporal = streaming hint
#ifndef SMX_NT_LOAD
#define SMX_NT_LOAD 1
#endif
#ifndef SMX_NT_STORE
#define SMX_NT_STORE 1
#endif

namespace smx {

#if defined(__HIPCC__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

struct cf { float x, y; };

SMX_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
SMX_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
SMX_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
SMX_HD cf cconj(cf a) { return mk(a.x, -a.y); }
SMX_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
// a*b
SMX_HD cf cmul(cf a, cf b) {
  return mk(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
// a*conj(b)
SMX_HD cf cmulc(cf a, cf b) {
  return mk(__builtin_fmaf(a.x, b.x, a.y * b.y), __builtin_fmaf(a.y, b.x, -(a.x * b.y)));
}
// acc + a*b
SMX_HD cf cfma(cf acc, cf a, cf b) {
  float re = __builtin_fmaf(a.x, b.x, acc.x);
  re = __builtin_fmaf(-a.y, b.y, re);
  float im = __builtin_fmaf(a.x, b.y, acc.y);
  im = __builtin_fmaf(a.y, b.x, im);
  return mk(re, im);
}
// acc + a*conj(b)
SMX_HD cf cfmac(cf acc, cf a, cf b) {
  float re = __builtin_fmaf(a.x, b.x, acc.x);
  re = __builtin_fmaf(a.y, b.y, re);
  float im = __builtin_fmaf(a.y, b.x, acc.y);
  im = __builtin_fmaf(-a.x, b.y, im);
  return mk(re, im);
}
SMX_HD cf mul_mi(cf a) { return mk(a.y, -a.x); }   // -i * a
SMX_HD cf mul_pi(cf a) { return mk(-a.y, a.x); }   // +i * a

// 16-byte accesses to the (B,k,D) complex spectra: two adjacent channels = 4 floats, 16-B aligned
// (D even, d even, base pointers 16-B aligned -- checked in smx_api).
SMX_HD void st4(float* p, float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x4 v; v.x = a; v.y = b; v.z = c; v.w = d;
  *reinterpret_cast<f32x4*>(p) = v;
#else
  p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}
// The same 16 bytes written THROUGH to the level every XCD sees (two 8-byte agent-scope stores: sc1 on gfx950), for
// data another workgroup of the SAME launch reads back (the appended reduction workgroups of the backward launch):
// no cache-wide write-back or invalidate is needed on either side, only s_waitcnt on this one.
SMX_HD void st4_agent(float* p, float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  union { float f[2]; unsigned long long u; } lo, hi;
  lo.f[0] = a; lo.f[1] = b; hi.f[0] = c; hi.f[1] = d;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}
SMX_HD void st1_agent(float* p, float a) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = a;
#endif
}
SMX_HD void ld4(const float* p, float& a, float& b, float& c, float& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  a = v.x; b = v.y; c = v.z; d = v.w;
#else
  a = p[0]; b = p[1]; c = p[2]; d = p[3];
#endif
}

// One channel pair of a streamed output row (y / grad_x tiles of the pointer-addressed kernels).  plain = false: the
// streaming hint (`nt`); plain = true: the default write-back policy -- see Geom::st_plain.
// The two floats are made opaque first: when they reach the store as the halves of a 64-bit value the optimiser has
// formed (the butterflies' register renaming is a struct copy), it rewrites the store and DROPS the nontemporal
// metadata -- rounds 1-3 shipped tiles whose 16 stores carried the hint on 4 (the diagonal of that renaming).
SMX_HD void st_stream(float* p, float x, float y, bool plain) {
#if defined(__HIP_DEVICE_COMPILE__) && SMX_NT_STORE
  asm("" : "+v"(x));
  asm("" : "+v"(y));
  if (plain) {
    // (a relaxed wavefront-scope atomic store IS the plain global_store_dwordx2 -- and, being another kind of
    //  instruction, is not merged with the streaming store of the other branch, which would drop the hint again)
    union { float f[2]; unsigned long long u; } b;
    b.f[0] = x; b.f[1] = y;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), b.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  } else {
    f32x2 w; w.x = x; w.y = y;
    __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(p));
  }
#elif defined(__HIP_DEVICE_COMPILE__)
  f32x2 w; w.x = x; w.y = y;
  *reinterpret_cast<f32x2*>(p) = w;
#else
  (void)plain;
  p[0] = x; p[1] = y;
#endif
}

// ---- radix-4 / radix-16 butterflies; SGN = -1 forward (w = e^{-2 pi i/n}), +1 inverse ----------
template <int SGN>
SMX_HD void radix4(cf& x0, cf& x1, cf& x2, cf& x3) {
  cf s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
  cf jd = (SGN < 0) ? mul_mi(d13) : mul_pi(d13);
  x0 = cadd(s02, s13);
  x1 = cadd(d02, jd);
  x2 = csub(s02, s13);
  x3 = csub(d02, jd);
}

// w16^e, e in {1,2,3,6,9}; sign applied to the imaginary part
template <int SGN, int E>
SMX_HD cf w16() {
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
  constexpr float H = 0.70710678118654752440f;
  constexpr float s = (float)SGN;
  if (E == 1) return mk(C1, s * S1);
  if (E == 2) return mk(H, s * H);
  if (E == 3) return mk(S1, s * C1);
  if (E == 6) return mk(-H, s * H);
  /* E == 9 */ return mk(-C1, -s * S1);
}

// In-place natural-order 16-point DFT: a[q] <- sum_u a[u] w16^{u q}.
template <int SGN>
SMX_HD void fft16(cf (&a)[16]) {
  // pass 1: u = 4*aa + b ; radix-4 over aa for each b -> T[b][c] kept at a[4c+b]
#pragma unroll
  for (int b = 0; b < 4; ++b) radix4<SGN>(a[b], a[4 + b], a[8 + b], a[12 + b]);
  // twiddle T[b][c] *= w16^{b c}
  a[4 * 1 + 1] = cmul(a[4 * 1 + 1], w16<SGN, 1>());
  a[4 * 1 + 2] = cmul(a[4 * 1 + 2], w16<SGN, 2>());
  a[4 * 1 + 3] = cmul(a[4 * 1 + 3], w16<SGN, 3>());
  a[4 * 2 + 1] = cmul(a[4 * 2 + 1], w16<SGN, 2>());
  a[4 * 2 + 2] = (SGN < 0) ? mul_mi(a[4 * 2 + 2]) : mul_pi(a[4 * 2 + 2]);   // w16^4
  a[4 * 2 + 3] = cmul(a[4 * 2 + 3], w16<SGN, 6>());
  a[4 * 3 + 1] = cmul(a[4 * 3 + 1], w16<SGN, 3>());
  a[4 * 3 + 2] = cmul(a[4 * 3 + 2], w16<SGN, 6>());
  a[4 * 3 + 3] = cmul(a[4 * 3 + 3], w16<SGN, 9>());
  // pass 2: radix-4 over b for each c -> out[c + 4d] lands at a[4c+d]
#pragma unroll
  for (int c = 0; c < 4; ++c) radix4<SGN>(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
  // natural order (pure register renaming once unrolled)
  cf t[16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) t[c + 4 * d] = a[4 * c + d];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = t[i];
}

// In-place natural-order 8-point DFT: a[k] <- sum_n a[n] w8^{n k}  (even / odd halves, radix-4 each).
template <int SGN>
SMX_HD void fft8(cf (&a)[8]) {
  constexpr float H = 0.70710678118654752440f;
  constexpr float s = (float)SGN;
  radix4<SGN>(a[0], a[2], a[4], a[6]);        // E[k] at a[0], a[2], a[4], a[6]
  radix4<SGN>(a[1], a[3], a[5], a[7]);        // O[k] at a[1], a[3], a[5], a[7]
  const cf o0 = a[1];
  const cf o1 = cmul(a[3], mk(H, s * H));                          // w8^1
  const cf o2 = (SGN < 0) ? mul_mi(a[5]) : mul_pi(a[5]);           // w8^2
  const cf o3 = cmul(a[7], mk(-H, s * H));                         // w8^3
  const cf e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  a[0] = cadd(e0, o0); a[4] = csub(e0, o0);
  a[1] = cadd(e1, o1); a[5] = csub(e1, o1);
  a[2] = cadd(e2, o2); a[6] = csub(e2, o2);
  a[3] = cadd(e3, o3); a[7] = csub(e3, o3);
}

// cp[q] = c^q for q = 1..15 with multiplication depth <= 4 (keeps twiddle error ~2e-7)
SMX_HD void powers16(cf c, cf (&cp)[16]) {
  cp[0] = mk(1.f, 0.f);
  cp[1] = c;
  cp[2] = cmul(c, c);
  cp[3] = cmul(cp[2], c);
  cp[4] = cmul(cp[2], cp[2]);
  cp[5] = cmul(cp[4], c);
  cp[6] = cmul(cp[4], cp[2]);
  cp[7] = cmul(cp[4], cp[3]);
  cp[8] = cmul(cp[4], cp[4]);
  cp[9] = cmul(cp[8], c);
  cp[10] = cmul(cp[8], cp[2]);
  cp[11] = cmul(cp[8], cp[3]);
  cp[12] = cmul(cp[8], cp[4]);
  cp[13] = cmul(cp[8], cp[5]);
  cp[14] = cmul(cp[8], cp[6]);
  cp[15] = cmul(cp[8], cp[7]);
}

// cp[q] = w_N^{q e}, q = 1..15, read from row e of the table tq[e][16] (make_tq: every entry rounded once from
// fp64) instead of being raised from c = w_N^e by powers16: 8 16-byte loads (the row is 128 bytes, 128-byte
// aligned; the 16 lanes that share a row-group read the same row) replace 14 complex products per tile and thread,
// and the twiddles lose the power tree's error.  cp[0] is never used.
SMX_HD void load_cp(const cf* __restrict__ row, cf (&cp)[16]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x4* p = reinterpret_cast<const f32x4*>(row);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = p[i];
    cp[2 * i] = mk(v.x, v.y);
    cp[2 * i + 1] = mk(v.z, v.w);
  }
#else
  for (int i = 0; i < 16; ++i) cp[i] = row[i];
#endif
}

// A wave-uniform table entry (the per-residue twiddles bt[r][.]) read through the CONSTANT address space: the
// compiler then issues s_load_dwordx16 into SGPRs, which the accumulate FMAs take as their scalar operand.  From
// an ordinary pointer the same reads become per-lane global_load_dwordx4 in a kernel that also stores (the
// backend cannot prove the table unclobbered): 1 KiB of return data per wave and instruction for 16 bytes of
// information, and -- vmcnt being one in-order counter for loads and stores -- a wait for every store issued before them.
SMX_HD cf ld_uniform(const cf* __restrict__ p, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const cf __attribute__((address_space(4)))* ccf;
  const cf v = ((ccf)(uintptr_t)p)[i];
  return v;
#else
  return p[i];
#endif
}

// ---- geometry -------------------------------------------------------------------------------
constexpr int M = 256;          // sub-transform length
constexpr int TPB = 256;        // 16 row-groups (t) x 16 packed channel pairs (j)
constexpr int DT = 32;          // real channels per workgroup
constexpr int BT_STRIDE = 64;   // scalar twiddle table: bt[r][s'+32], s' in [-32,32)
constexpr int BT_HALF = 32;
constexpr int EX = 16 * 16 * 16;  // complex elements of one LDS exchange buffer (32 KiB)

// A thread keeps 16 accumulator slots per band; slot = 16*bi + s, unsigned bin fu = q + 16 s.
// NB == 1: one band with wrap-around: fs = fu < 128 ? fu : fu-256                     (|f| < 128)
// NB >= 2: band bi holds fs = fu + 256*beta(bi), beta = 0, -1, +1, -2 for bi = 0..3   (|f| < 128 NB)
SMX_HD int band_beta(int bi) { return (bi & 1) ? -((bi + 1) >> 1) : (bi >> 1); }
SMX_HD int band_index(int beta) { return beta >= 0 ? 2 * beta : -2 * beta - 1; }
template <int NB>
SMX_HD int slot_fs(int q, int slot) {
  int s = slot & 15, fu = q + 16 * s;
  if (NB == 1) return fu < 128 ? fu : fu - 256;
  return fu + 256 * band_beta(slot >> 4);
}
// index into the per-r scalar twiddle row for a slot:  w_N^{16 s' r},  fs = q + 16 s'
template <int NB>
SMX_HD int slot_bt(int slot) {
  int s = slot & 15;
  if (NB == 1) return (s < 8 ? s : s - 16) + BT_HALF;
  return s + 16 * band_beta(slot >> 4) + BT_HALF;
}
// slot (of thread (16-q)&15) that holds bin -fs
template <int NB>
SMX_HD int partner_slot(int q, int slot) {
  int s = slot & 15;
  int sp = q ? 15 - s : ((16 - s) & 15);
  if (NB == 1) return sp;
  const int beta = band_beta(slot >> 4);
  // fu != 0: -f = (256 - fu) + 256 (-beta - 1) ;  fu == 0: -f = 0 + 256 (-beta)
  const int pb = (q == 0 && s == 0) ? -beta : -beta - 1;
  if (pb < -(NB / 2) || pb >= NB / 2) return slot;      // f = -128 NB: never a kept bin (k <= 128 NB)
  return 16 * band_index(pb) + sp;
}
// bins with fs >= 0 own the rows of the (B,k,D) spectrum
template <int NB>
SMX_HD bool slot_pos(int slot) {
  if (NB == 1) return (slot & 15) < 8;
  return band_beta(slot >> 4) >= 0;
}

struct Geom {
  int B, N, D, F, k, L;
  float inv_n;        // 1/N
  int R;              // rows present in x / y (R <= N): rows n >= R read as zero and are not written --
                      // the zero-padded causal convolution of fft_lm (reference train_fixed_full.py:507-519,
                      // :553-555); batch stride of x / y is R * D
  int P = 0;          // sixteen-row decimation (N = 16 P, N % 256 != 0): residues; L counts its tiles of 16 residues.
                      // 0 on every other plan
  int st_plain = 4;   // of the 16 rows a thread stores per tile, the first st_plain (0, 1, 2, 4) go out with the default
                      // write-back policy, the others streaming: about 64 MiB of an output tensor written back through
                      // L2 / Infinity Cache is free (it drains under the next launch's reads), all-streaming and
                      // all-cached stores are both slower at step level (profiles/r04_store_policy.txt); set per
                      // launch from the output size (decim_args)
};
// The bin f = -128 NB is its own mirror image when N = 256 NB (f = N/2, the Nyquist bin): kept when
// k = N/2 + 1 (full one-sided spectrum), it behaves like DC -- real for real input, only Re(W X) counts.
template <int NB>
SMX_HD bool self_nyquist(const Geom& g, int fs) { return fs == -128 * NB && g.L == NB && g.P == 0; }

// ---- dropout (training mode of SpectralMixingLayer.forward, reference spectral_layers.py:118) ----
// Counter-based: the decision for element (b, n, d) is a pure function of (state, b, n D + d), so the
// forward store and the backward load regenerate the same mask whatever the plan or traversal order.
// One 32-bit hash serves the element pair (2i, 2i+1) of a batch row -- the channel pair of a tile row
// when D is even: low / high 16 bits against thr = round(p 65536); survivors are scaled by
// 65536 / (65536 - thr), so the expectation is exact for the quantised p.
struct Drop {
  unsigned thr;        // 0 = no dropout
  float scale;
  unsigned key;        // per launch and batch row
};
SMX_HD unsigned drop_hash(unsigned pair_index, unsigned key) {
  unsigned x = pair_index ^ key;
  x *= 0x9E3779B1u; x ^= x >> 15;
  x *= 0x85EBCA77u; x ^= x >> 13;
  x *= 0xC2B2AE3Du; x ^= x >> 16;
  return x;
}
SMX_HD cf drop_apply(cf v, unsigned h, unsigned thr, float scale) {
  return mk((h & 0xffffu) >= thr ? v.x * scale : 0.f, (h >> 16) >= thr ? v.y * scale : 0.f);
}
// key of batch row b from the 128-bit generator state (seed, call counter): splitmix64 finaliser
SMX_HD unsigned drop_row_key(unsigned long long seed, unsigned long long counter, int b) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + 0xD1B54A32D192ED03ull * (unsigned long long)b;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z ^ (z >> 32));
}

// Per-thread state that lives across barriers (all statically indexed -> registers on the GPU).
template <int NB>
struct TState {
  cf v[16];            // working set of the current tile
  cf acc[16 * NB];     // forward: Z[f] accumulators; inverse: S[f]
  cf cp[16];           // c^q, c = w_N^{L t + r}
  cf io[NB == 2 ? 32 : 16];   // this thread's rows of the (B,k,D) spectra (bins >= 0 x 2 channels), see io_regs:
                              // X prefetched at launch start (backward) / values stored at launch end
};
// Spectrum IO through registers at the two ends of the launch (prefetch_io / store_io) instead of as
// dependent 16-B accesses inside the unpack loop: one band in every mode, two bands in backward only
// (64 registers there; the forward stores are fire-and-forget and stay in the loop).
template <int NB, int MODE> SMX_HD constexpr bool io_regs() { return NB == 1 || (NB == 2 && MODE == 1); }
template <int NB> SMX_HD constexpr int io_bins() { return NB == 1 ? 8 : 16; }   // bins >= 0 per thread

// ---- global <-> register tile moves ---------------------------------------------------------
// row n = (t + 16u) L + r ; thread reads channels (d, d+1) of 16 rows.
// Loads are unconditional: a lane whose channel pair lies past D is pointed at a valid pair by
// the caller (its packed sequence never mixes with the others and is never stored).
// NT = false: ordinary cached loads -- for the one kernel whose two thread teams read the SAME tile (k_conv1): with
// the streaming hint the second team's request went to HBM again (2 x the bytes of x counted, profiles/r03b_f2_*).
template <bool PAD = false, bool NT = true>
SMX_HD void load_tile(const float* __restrict__ xb, const Geom& g, int t, int r, cf (&v)[16]) {
  const size_t stride = (size_t)16 * g.L * g.D;
  const float* p = xb + ((size_t)t * g.L + r) * g.D;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (PAD && (t + 16 * u) * g.L + r >= g.R) { v[u] = mk(0.f, 0.f); continue; }
#if defined(__HIP_DEVICE_COMPILE__) && SMX_NT_LOAD
    f32x2 w = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p + u * stride))
                 : *reinterpret_cast<const f32x2*>(p + u * stride);
#elif defined(__HIP_DEVICE_COMPILE__)
    f32x2 w = *reinterpret_cast<const f32x2*>(p + u * stride);
#else
    float2 w = *reinterpret_cast<const float2*>(p + u * stride);
#endif
    v[u] = mk(w.x, w.y);
  }
}

// part of a tile (rows u = U0 .. U0+CNT-1): lets the caller spread the 16 loads over the iteration
template <int U0, int CNT, bool PAD = false, bool NT = true>
SMX_HD void load_part_tile(const float* __restrict__ xb, const Geom& g, int t, int r, cf (&v)[16]) {
  const size_t stride = (size_t)16 * g.L * g.D;
  const float* p = xb + ((size_t)t * g.L + r) * g.D;
#pragma unroll
  for (int u = U0; u < U0 + CNT; ++u) {
    if (PAD && (t + 16 * u) * g.L + r >= g.R) { v[u] = mk(0.f, 0.f); continue; }
#if defined(__HIP_DEVICE_COMPILE__) && SMX_NT_LOAD
    f32x2 w = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p + u * stride))
                 : *reinterpret_cast<const f32x2*>(p + u * stride);
#elif defined(__HIP_DEVICE_COMPILE__)
    f32x2 w = *reinterpret_cast<const f32x2*>(p + u * stride);
#else
    float2 w = *reinterpret_cast<const float2*>(p + u * stride);
#endif
    v[u] = mk(w.x, w.y);
  }
}

template <bool PAD = false>
SMX_HD void store_tile(float* __restrict__ yb, const Geom& g, int t, int r, bool valid,
                       const cf (&v)[16]) {
  const size_t stride = (size_t)16 * g.L * g.D;
  float* p = yb + ((size_t)t * g.L + r) * g.D;
  if (!valid) return;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (PAD && (t + 16 * u) * g.L + r >= g.R) continue;
    st_stream(p + u * stride, v[u].x, v[u].y, u < 4 && u < g.st_plain);
  }
}

// ---- forward tile: two phases around one barrier -------------------------------------------
// phase 1: radix-16 over u, twiddle by c^q, scatter to E[t][q][j]
template <int NB>
SMX_HD void fwd_phase1(TState<NB>& st, cf c, cf* __restrict__ E, int t, int j) {
  powers16(c, st.cp);
  fft16<-1>(st.v);
#pragma unroll
  for (int q = 1; q < 16; ++q) st.v[q] = cmul(st.v[q], st.cp[q]);
#pragma unroll
  for (int q = 0; q < 16; ++q) E[(t * 16 + q) * 16 + j] = st.v[q];
}
// the same with the twiddles c^q already in st.cp (load_cp: table row instead of the power tree)
template <int NB>
SMX_HD void fwd_phase1_cp(TState<NB>& st, cf* __restrict__ E, int t, int j) {
  fft16<-1>(st.v);
#pragma unroll
  for (int q = 1; q < 16; ++q) st.v[q] = cmul(st.v[q], st.cp[q]);
#pragma unroll
  for (int q = 0; q < 16; ++q) E[(t * 16 + q) * 16 + j] = st.v[q];
}
// phase 2: gather E[t'][q=t][j], radix-16 over t', accumulate with the scalar twiddles of row r
// SC: the twiddle row through the scalar cache (ld_uniform) -- SGPR operands of the FMAs
template <int NB, bool SC = false>
SMX_HD void fwd_phase2(TState<NB>& st, const cf* __restrict__ E, const cf* __restrict__ bt_r,
                       int t, int j) {
  cf e[16];
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) e[t2] = E[(t2 * 16 + t) * 16 + j];
  fft16<-1>(e);
#pragma unroll
  for (int sl = 0; sl < 16 * NB; ++sl)
    st.acc[sl] = cfma(st.acc[sl], SC ? ld_uniform(bt_r, slot_bt<NB>(sl)) : bt_r[slot_bt<NB>(sl)], e[sl & 15]);
}

// ---- full spectrum, N = 256 NB (eight-band kernel): per-residue spectra kept apart, then an NB-point
// transform across the residues -- Z[fu + 256 f2] = sum_r w_NB^{f2 r} (w_N^{fu r} DFT256_r[fu]) -- instead
// of NB accumulations per tile (8 x 128 complex FMAs at NB = 8).  acc[16 R + s] holds residue R during the
// loops and band band_index(beta) between them (the layout of the unpack phase).
template <int NB, int R>
SMX_HD void fwd_phase2_store(TState<NB>& st, const cf* __restrict__ E, const cf* __restrict__ bt_r,
                             int t, int j) {
  cf e[16];
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) e[t2] = E[(t2 * 16 + t) * 16 + j];
  fft16<-1>(e);
#pragma unroll
  for (int s = 0; s < 16; ++s) st.acc[16 * R + s] = cmul(ld_uniform(bt_r, s + BT_HALF), e[s]);
}
template <int NB, int R>
SMX_HD void inv_phase1_from(TState<NB>& st, const cf* __restrict__ bt_r, cf* __restrict__ E, int q, int j) {
#pragma unroll
  for (int s = 0; s < 16; ++s) st.v[s] = cmulc(st.acc[16 * R + s], ld_uniform(bt_r, s + BT_HALF));
  fft16<+1>(st.v);
#pragma unroll
  for (int p = 0; p < 16; ++p) E[(q * 16 + p) * 16 + j] = st.v[p];
}
SMX_HD constexpr int f2_band(int f2) { return f2 < 4 ? 2 * f2 : 2 * (8 - f2) - 1; }   // band_index(f2 or f2 - 8)
// residues -> bands (forward, SGN = -1) / bands -> residues (inverse, SGN = +1), NB == 8
template <int SGN>
SMX_HD void residue_fft8(TState<8>& st) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    cf a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = st.acc[16 * (SGN < 0 ? i : f2_band(i)) + s];
    fft8<SGN>(a);
#pragma unroll
    for (int i = 0; i < 8; ++i) st.acc[16 * (SGN < 0 ? f2_band(i) : i) + s] = a[i];
  }
}

// ---- inverse tile ----------------------------------------------------------------------------
template <int NB, bool SC = false>
SMX_HD void inv_phase1(TState<NB>& st, const cf* __restrict__ bt_r, cf* __restrict__ E, int q,
                       int j) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    cf a = cmulc(st.acc[s], SC ? ld_uniform(bt_r, slot_bt<NB>(s)) : bt_r[slot_bt<NB>(s)]);
#pragma unroll
    for (int bi = 1; bi < NB; ++bi)          // the other bands alias onto the same 256-point bin
      a = cfmac(a, st.acc[16 * bi + s],
                SC ? ld_uniform(bt_r, slot_bt<NB>(16 * bi + s)) : bt_r[slot_bt<NB>(16 * bi + s)]);
    st.v[s] = a;
  }
  fft16<+1>(st.v);
#pragma unroll
  for (int p = 0; p < 16; ++p) E[(q * 16 + p) * 16 + j] = st.v[p];
}
// inv_phase2 in two steps, twiddles from st.cp (load_cp): between them st.cp is dead, so the caller can request the
// next tile's row into the same registers BEFORE this tile's stores enter the (in-order) vector-memory queue
template <int NB>
SMX_HD void inv_phase2_gather(TState<NB>& st, const cf* __restrict__ E, int t, int j) {
#pragma unroll
  for (int q2 = 0; q2 < 16; ++q2) st.v[q2] = E[(q2 * 16 + t) * 16 + j];
#pragma unroll
  for (int q2 = 1; q2 < 16; ++q2) st.v[q2] = cmulc(st.v[q2], st.cp[q2]);
}
template <int NB>
SMX_HD void inv_phase2(TState<NB>& st, cf c, const cf* __restrict__ E, int t, int j) {
  powers16(c, st.cp);
#pragma unroll
  for (int q2 = 0; q2 < 16; ++q2) st.v[q2] = E[(q2 * 16 + t) * 16 + j];
#pragma unroll
  for (int q2 = 1; q2 < 16; ++q2) st.v[q2] = cmulc(st.v[q2], st.cp[q2]);
  fft16<+1>(st.v);
}

// ---- sixteen-row decimation: N = 16 P for ANY P (round 3) ------------------------------------------------
// Lengths that are a multiple of 16 but not of 256 (2000, 4000, 6000, 1200, 128 ...) decimate the other way
// round: n = P m + r with m < 16, so
//     Z[f] = sum_{r < P} w_N^{f r} G_r[f mod 16],   G_r = DFT16 over m of z[P m + r]
// -- one 16-point transform per residue (in one thread's registers) instead of a 256-point one per residue, and
// O(N k / 16) accumulation work instead of the O(N k) of the DFT products.  A tile is 16 residues r = 16 tau + t
// (thread row-group t <-> residue) x 16 rows m; after the usual scatter / gather through LDS, thread q holds the
// 16 values G_r'[q] w_N^{q r'}, r' = 16 tau + t', and owns the same bins f = q + 16 s' as in the 256-point
// kernels, so everything between the loops (unpack, filter, spectrum IO) is shared.  What is left of the residue
// twiddle, w_N^{16 s' r'} = w_P^{s' (16 tau + t')}, is beta[tau][s] V[s][t']: a 16 x 16 matrix V that is the same
// for every tile (and every thread: scalar operands) and 16 scalars per tile.
//     acc[s] += beta[tau][s] sum_t' V[s][t'] e[t']                                    (256 + 16 complex FMAs)
// against one more fft16 + 16 FMAs in the 256-point kernels: about 2.5x their arithmetic, x read once, y written once.
// Rows: n = P u + 16 tau + t, u < 16; residues r >= P of the last tile and rows n >= R (zero-padded input, cropped
// output: Geom::R) read as zero and are not written.
template <bool PAD = false>
SMX_HD void load_tile16(const float* __restrict__ xb, const Geom& g, int t, int tau, cf (&v)[16]) {
  const int r = 16 * tau + t;
  const size_t stride = (size_t)g.P * g.D;
  const float* p = xb + (size_t)(r < g.P ? r : 0) * g.D;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (r >= g.P || (PAD && g.P * u + r >= g.R)) { v[u] = mk(0.f, 0.f); continue; }   // padding residue / zero-padded row
#if defined(__HIP_DEVICE_COMPILE__) && SMX_NT_LOAD
    f32x2 w = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p + u * stride));
#elif defined(__HIP_DEVICE_COMPILE__)
    f32x2 w = *reinterpret_cast<const f32x2*>(p + u * stride);
#else
    float2 w = *reinterpret_cast<const float2*>(p + u * stride);
#endif
    v[u] = mk(w.x, w.y);
  }
}
template <bool PAD = false>
SMX_HD void store_tile16(float* __restrict__ yb, const Geom& g, int t, int tau, bool valid, const cf (&v)[16]) {
  const int r = 16 * tau + t;
  if (!valid || r >= g.P) return;
  const size_t stride = (size_t)g.P * g.D;
  float* p = yb + (size_t)r * g.D;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (PAD && g.P * u + r >= g.R) continue;                                      // cropped row
    st_stream(p + u * stride, v[u].x, v[u].y, u < 4 && u < g.st_plain);
  }
}
// accumulator slot of the bin block s'' in [-8 NB, 8 NB) (f = q + 16 s'') and its table row s'' + 16
template <int NB> SMX_HD constexpr int slot16(int sp) { return sp >= 0 ? sp : 16 * NB + sp; }
// forward, after the barrier: thread q gathers e[t'] = G_r'[q] w_N^{q r'} and accumulates its 16 NB bins.
// V[-m] = conj V[m]: the blocks +m and -m share the two REAL-weighted sums P = sum Re V[m][t'] e[t'] and
// Q = sum Im V[m][t'] e[t'] -- z(+m) = P + i Q, z(-m) = P - i Q -- 4 FMAs per term for the pair instead of 8.
template <int NB>
SMX_HD void fwd16_phase2(TState<NB>& st, const cf* __restrict__ E, const cf* __restrict__ v16,
                         const cf* __restrict__ beta, int q, int j) {
  cf e[16];
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) e[t2] = E[(t2 * 16 + q) * 16 + j];
  {                                               // s'' = 0: V = 1
    cf z = e[0];
#pragma unroll
    for (int t2 = 1; t2 < 16; ++t2) z = cadd(z, e[t2]);
    st.acc[0] = cfma(st.acc[0], beta[16], z);
  }
#pragma unroll
  for (int m = 1; m < 8 * NB; ++m) {
    const cf* vr = v16 + (m + 16) * 16;
    const cf w0 = vr[0];
    cf P = mk(w0.x * e[0].x, w0.x * e[0].y), Q = mk(w0.y * e[0].x, w0.y * e[0].y);
#pragma unroll
    for (int t2 = 1; t2 < 16; ++t2) {
      const cf w = vr[t2];
      P = mk(__builtin_fmaf(w.x, e[t2].x, P.x), __builtin_fmaf(w.x, e[t2].y, P.y));
      Q = mk(__builtin_fmaf(w.y, e[t2].x, Q.x), __builtin_fmaf(w.y, e[t2].y, Q.y));
    }
    st.acc[slot16<NB>(m)] = cfma(st.acc[slot16<NB>(m)], beta[16 + m], mk(P.x - Q.y, P.y + Q.x));
    st.acc[slot16<NB>(-m)] = cfma(st.acc[slot16<NB>(-m)], beta[16 - m], mk(P.x + Q.y, P.y - Q.x));
  }
  {                                               // s'' = -8 NB: no partner among the kept blocks
    constexpr int m = -8 * NB;
    const cf* vr = v16 + (m + 16) * 16;
    cf z = cmul(vr[0], e[0]);
#pragma unroll
    for (int t2 = 1; t2 < 16; ++t2) z = cfma(z, vr[t2], e[t2]);
    st.acc[slot16<NB>(m)] = cfma(st.acc[slot16<NB>(m)], beta[16 + m], z);
  }
}
// (Round 4 tried these tables through the scalar cache as well (ld_uniform): 45 s_load per tile, each waited for with
//  lgkmcnt(0) beside the LDS traffic -- (64, 4000, 256) fwd+bwd 0.344 -> 0.369 ms.  Per-lane loads stay here.)
// inverse, before the barrier: h[t'] = sum over blocks of conj(V[s''][t']) conj(beta[s'']) S[q + 16 s''], scattered
// for thread t'.  The pair +-m: a(+m) conj V + a(-m) V = (a(+m) + a(-m)) Re V + i (a(-m) - a(+m)) Im V.
template <int NB>
SMX_HD void inv16_phase1(TState<NB>& st, const cf* __restrict__ v16, const cf* __restrict__ beta,
                         cf* __restrict__ E, int q, int j) {
  // (block loop outside, row loop inside: 16 running sums instead of 16 NB sums and differences held at once --
  //  the two-band kernels spilled 44 registers the other way round)
  cf h[16];
  {
    const cf a0 = cmulc(st.acc[0], beta[16]);                                        // s'' = 0: V = 1
    const cf ae = cmulc(st.acc[slot16<NB>(-8 * NB)], beta[16 - 8 * NB]);             // s'' = -8 NB: no partner
#pragma unroll
    for (int p = 0; p < 16; ++p) h[p] = cadd(a0, cmulc(ae, v16[(16 - 8 * NB) * 16 + p]));
  }
#pragma unroll
  for (int m = 1; m < 8 * NB; ++m) {
    const cf ap = cmulc(st.acc[slot16<NB>(m)], beta[16 + m]), an = cmulc(st.acc[slot16<NB>(-m)], beta[16 - m]);
    const cf sm = cadd(ap, an), df = csub(an, ap);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const cf v = v16[(m + 16) * 16 + p];
      h[p] = mk(__builtin_fmaf(sm.x, v.x, __builtin_fmaf(-df.y, v.y, h[p].x)),
                __builtin_fmaf(sm.y, v.x, __builtin_fmaf(df.x, v.y, h[p].y)));
    }
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) E[(q * 16 + p) * 16 + j] = h[p];
}

// ---- unpack + filter (once per workgroup, between the two loops) ------------------------------
// The exchange that lets a thread fetch Z[-f] goes through LDS in ROUNDS of at most 32 slots per
// thread (64 KiB, the size of the double exchange buffer of the loops): one round for NB <= 2, two
// for NB == 4 -- bands (0,-1) then (+1,-2), which are each other's mirror images -- so the four-band
// kernels also fit two workgroups per CU.  The only pair that crosses the rounds is f = -256 <-> +256
// (slots 16 and 32 of the q == 0 threads, which are their own partners): served from registers.
template <int NB> struct UnpackRounds {
  static constexpr int N = NB >= 4 ? NB / 2 : 1;     // rounds of bands (beta, -beta-1): mirror images
  static constexpr int SLOTS = 16 * NB / N;          // slots published per round
};
// The bins that are multiples of 256 (slot 16 bi of the q == 0 threads) mirror into ANOTHER round -- or
// into themselves (DC, and f = -128 NB) -- but always into a slot of the same thread: their Z is taken
// aside before round 0 and served from registers.
template <int NB> struct ZSave { cf z[NB]; };
template <int NB>
SMX_HD ZSave<NB> save_z(const TState<NB>& st) {
  ZSave<NB> r;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) r.z[bi] = st.acc[16 * bi];
  return r;
}

// phase U1: publish the accumulators of one round   U[slot - first][q][j]
template <int NB, int ROUND = 0>
SMX_HD void unpack_phase1(const TState<NB>& st, cf* __restrict__ U, int q, int j) {
  constexpr int S0 = ROUND * UnpackRounds<NB>::SLOTS;
#pragma unroll
  for (int sl = S0; sl < S0 + UnpackRounds<NB>::SLOTS; ++sl) U[((sl - S0) * 16 + q) * 16 + j] = st.acc[sl];
}

// Z[-f] for slot sl during round ROUND.  zs: save_z() taken before round 0 (used for NB >= 4 only).
template <int NB, int ROUND>
SMX_HD cf unpack_partner(const TState<NB>& st, const cf* __restrict__ U, int q, int qp, int j, int sl,
                         const ZSave<NB>& zs) {
  constexpr int S0 = ROUND * UnpackRounds<NB>::SLOTS;
  const int ps = partner_slot<NB>(q, sl);
  if (NB >= 4 && (sl & 15) == 0) {
    const bool own_bin = q == 0;                      // f = 256 beta: the mirror image is one of this thread's slots
    const int pb = -band_beta(sl >> 4);
    const bool in_range = pb >= -(NB / 2) && pb < NB / 2;
    const cf own = zs.z[in_range ? band_index(pb) : (sl >> 4)];
    const cf u = U[(((own_bin ? sl : ps) - S0) * 16 + qp) * 16 + j];
    return own_bin ? own : u;
  }
  return U[((ps - S0) * 16 + qp) * 16 + j];
}

struct FilterArgs {
  const float* w_re;      // (D,F)
  const float* w_im;      // (D,F)
  const float* wt;        // (k,D) complex = the two above transposed and interleaved (launch_pack_w), or
                          // null: one 16-B load per channel pair and bin, coalesced over the 16 j lanes,
                          // instead of four scalar gathers at stride F
  const float* bias;      // (D) or null          (forward only)
  float* xk_out;          // (B,k,D) c64 or null  (forward: saved spectrum; also "spectrum only" API)
  const float* xk_in;     // (B,k,D) c64          (backward: spectrum saved by forward)
  float* pslab;           // (B,k,D) c64          (backward: X*conj(G)/N per batch row)
  float* gb_part;         // (B,D)                (backward: sum_n g per batch row)
  int conj_w;             // multiply by conj(W)
  int slab_agent;         // backward: slab rows / bias partials are read back by workgroups of the SAME launch
                          // (gradw_tail) -> written through with agent-scope stores
  // per-(batch row, channel) real factor on the filter, W_eff[b,d,f] = W[d,f] sc[b,d] -- the context gate of
  // fft_lm's FixedSpectralBlock (reference train_fixed_full.py:532-536) without an extra pass over y.
  // Backward: the slab rows carry the factor (so their batch sum is grad_W) and gsc[b,d] receives
  // d/dsc = sum_f Re(W X conj(G)) / N  (= sum_n g y0, y0 = the unscaled output).
  const float* sc;        // (B,D) or null
  float* gsc;             // (B,D) or null        (backward, one workgroup per (b, d-tile))
  cf* gsc_part;           // four-step path: [B*ndt][9][16] partial sums of the column-unit blocks
  // band groups (k > 512, four-band kernels only): this launch covers the bins |f| in
  // [goff, goff + 512), goff = 512 * group.  With more than one group (multi) the two bins a
  // band-aligned launch cannot pair inside itself -- local f = -512, and local f = 0 of the groups
  // after the first, i.e. the global bins that are multiples of 512 -- are left to the edge kernels.
  int goff, multi;
  // synthesis from a given one-sided spectrum (smx_irfft_ex; rows read through xk_in):
  //   y = sp_scale * Re sum_f c_f Y[f] e^{+2 pi i f n / N},  c_f = 2 for 0 < f < N/2 when sp_herm, else 1
  float sp_scale;
  int sp_herm;
};
SMX_HD bool group_edge_slot(const FilterArgs& fa, int fs) {
  return (fa.multi && fs == -512) || (fa.goff > 0 && fs == 0);
}

// NB == 1 fused kernels: the workgroup's slice of the filter (32 channels x 128 bins) is read from the
// reference's (D,F) arrays with coalesced row reads at the START of the launch (32 registers per thread),
// parked in the half of the exchange buffer the one-round unpack leaves idle, and read back as one
// 16-byte LDS load per bin and channel pair.  Replaces the separate transpose launch (k_pack_w, 5 us of
// every step at C2).  Row pitch 34 complex (272 B): 16-B aligned rows, conflict-free ds_read_b128.
constexpr int WL_PITCH = 34;
constexpr int WL_ELEMS = 128 * WL_PITCH;
struct WPre { float re[16], im[16]; };
// thread tid reads bin f = tid & 127 of channels d0 + (tid >> 7) + 2 i
SMX_HD void prefetch_w(WPre& w, const Geom& g, const float* __restrict__ w_re,
                       const float* __restrict__ w_im, int d0, int tid) {
  const int f = tid & 127, h = tid >> 7;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = d0 + h + 2 * i;
    const bool ok = d < g.D && f < g.k;
    const size_t o = ok ? (size_t)d * g.F + f : 0;
    const float a = w_re[o], b = w_im[o];
    w.re[i] = ok ? a : 0.f;
    w.im[i] = ok ? b : 0.f;
  }
}
SMX_HD void stage_w(const WPre& w, cf* __restrict__ wl, int tid, int conj_w) {
  const int f = tid & 127, h = tid >> 7;
#pragma unroll
  for (int i = 0; i < 16; ++i) wl[f * WL_PITCH + h + 2 * i] = mk(w.re[i], conj_w ? -w.im[i] : w.im[i]);
}

// ---- synthesis: packed spectrum from one row of a given one-sided spectrum ------------------------
// r = (Ya.re, Ya.im, Yb.re, Yb.im) of the channel pair at bin |f|; the packed sequence z = a + i b has
// Z[+f] = (Ya + i Yb) h, Z[-f] = (conj Ya + i conj Yb) h, h = weight / 2; a bin that is its own mirror image
// (DC, Nyquist) contributes Re(Y) only -- what torch.fft.irfft does with the imaginary parts there.
// dbl: the bin has a distinct mirror image and the Hermitian weight c_f = 2 applies.  (A Nyquist bin held in TWO
// slots, +N/2 and -N/2 -- three tiles under the four-band kernels -- is not `self`: each slot carries half of it.)
SMX_HD void synth_pair(const float (&r)[4], bool self, float scale, bool dbl, cf& Spos, cf& Sneg) {
  if (self) {
    Spos = mk(r[0] * scale, r[2] * scale);
    Sneg = Spos;
  } else {
    const float h = dbl ? scale : 0.5f * scale;
    Spos = mk((r[0] - r[3]) * h, (r[1] + r[2]) * h);
    Sneg = mk((r[0] + r[3]) * h, (-r[1] + r[2]) * h);
  }
}
// every accumulator slot of thread (q, j) from the rows of fa.xk_in (loads in batches of eight)
template <int NB>
SMX_HD void synth_fill(TState<NB>& st, const Geom& g, const FilterArgs& fa, int b, int d, bool valid, int q) {
  const int dl = valid ? d : g.D - 2;
  constexpr int CH = NB == 8 ? 16 : 8;     // eight bands: one workgroup per CU, more rows in flight per thread
#pragma unroll
  for (int c0 = 0; c0 < 16 * NB; c0 += CH) {
    float r[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int fs = slot_fs<NB>(q, c0 + i);
      const int af = fs < 0 ? -fs : fs;
      const int afc = af < g.k ? af : 0;
      ld4(fa.xk_in + (((size_t)b * g.k + afc) * g.D + dl) * 2, r[i][0], r[i][1], r[i][2], r[i][3]);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int fs = slot_fs<NB>(q, c0 + i);
      const int af = fs < 0 ? -fs : fs;
      const bool self = af == 0 || self_nyquist<NB>(g, fs);
      cf sp, sn;
      synth_pair(r[i], self, fa.sp_scale, fa.sp_herm && 2 * af != g.N, sp, sn);
      const cf S = (fs >= 0 || self) ? sp : sn;
      st.acc[c0 + i] = (valid && af < g.k) ? S : mk(0.f, 0.f);
    }
  }
}

// phase U2: fetch Z[-f], split the packed pair into (A,B), apply W, rebuild the packed spectrum S.
// MODE 0 = forward, 1 = backward (also emits the grad_w slab row and the grad_bias partial),
// 2 = spectrum only (no weights are read; S is left zero).
template <int NB, int MODE, int ROUND = 0>
SMX_HD void unpack_phase2(TState<NB>& st, const cf* __restrict__ U, const Geom& g,
                          const FilterArgs& fa, int b, int d, bool valid, int q, int j,
                          const ZSave<NB>& zsave = ZSave<NB>{}, const cf* __restrict__ wl = nullptr,
                          cf* gs = nullptr) {
  const int qp = (16 - q) & 15;
  constexpr int S0 = ROUND * UnpackRounds<NB>::SLOTS;
  constexpr int NS = UnpackRounds<NB>::SLOTS;
  // Multi-band kernels read the filter (and, in backward, the saved spectrum) from global memory, one
  // 16-byte row per slot.  The loads of slot sl + PF are issued while slot sl is processed (clamped,
  // always-valid addresses), so the 32 slots of a round cost a few memory latencies instead of 32 --
  // which matters most where little else is resident to hide them (four and eight bands).
  // Depth 4; 8 for two bands in backward (A/B at (64,4096,512), k = 256: backward launch 272 -> 265 us; forward
  // unchanged at any depth -- what the "unpack phase" costs there is its 67 MB of saved-spectrum stores, not latency).
#ifndef SMX_PF2_FWD
#define SMX_PF2_FWD 4
#endif
#ifndef SMX_PF2_BWD
#define SMX_PF2_BWD 8
#endif
  constexpr int PF = NB == 2 ? (MODE == 1 ? SMX_PF2_BWD : SMX_PF2_FWD) : NB >= 2 ? 4 : 0;
  constexpr bool XQ = MODE == 1 && !io_regs<NB, MODE>() && NB >= 2;     // saved-spectrum rows by prefetch
  float wq[PF ? PF : 1][4], xq[PF ? PF : 1][4];
  const int dl = valid ? d : g.D - 2;
  float sca = 1.f, scb = 1.f;
  if (fa.sc) { sca = fa.sc[(size_t)b * g.D + dl]; scb = fa.sc[(size_t)b * g.D + dl + 1]; }
  float gsx = 0.f, gsy = 0.f;           // row-scale gradient terms of this round (registers; *gs updated once)
  auto issue = [&](int sl2, int ring) {
    const int fs2 = slot_fs<NB>(q, sl2);
    const int af2 = (fs2 < 0 ? -fs2 : fs2) + fa.goff;
    const int afc = af2 < g.k ? af2 : 0;
    if (MODE != 2) {
      if (fa.wt) {
        ld4(fa.wt + ((size_t)afc * g.D + dl) * 2, wq[ring][0], wq[ring][1], wq[ring][2], wq[ring][3]);
      } else {
        const size_t wo = (size_t)dl * g.F + afc;
        wq[ring][0] = fa.w_re[wo]; wq[ring][1] = fa.w_im[wo];
        wq[ring][2] = fa.w_re[wo + g.F]; wq[ring][3] = fa.w_im[wo + g.F];
      }
    }
    if (XQ && (slot_pos<NB>(sl2) || sl2 == 16 * (NB - 1)))
      ld4(fa.xk_in + (((size_t)b * g.k + afc) * g.D + dl) * 2, xq[ring][0], xq[ring][1], xq[ring][2],
          xq[ring][3]);
  };
  if (PF > 0) {
#pragma unroll
    for (int i = 0; i < PF && i < NS; ++i) issue(S0 + i, i);
  }
#pragma unroll
  for (int sl = S0; sl < S0 + UnpackRounds<NB>::SLOTS; ++sl) {
    const int fs = slot_fs<NB>(q, sl);
    const int af = (fs < 0 ? -fs : fs) + fa.goff;           // global |bin|
    const bool snyq = self_nyquist<NB>(g, fs);
    const cf zo = st.acc[sl];
    const cf zp = unpack_partner<NB, ROUND>(st, U, q, qp, j, sl, zsave);
    const int ring = PF ? (sl - S0) % (PF ? PF : 1) : 0;
    float wv[4] = {0.f, 0.f, 0.f, 0.f}, xv[4] = {0.f, 0.f, 0.f, 0.f};
    if (PF > 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { wv[c] = wq[ring][c]; xv[c] = xq[ring][c]; }
      if (sl + PF < S0 + NS) issue(sl + PF, ring);
    }
    // Z[+af], Z[-af]
    const cf zpos = fs >= 0 ? zo : zp;
    const cf zneg = fs >= 0 ? zp : zo;
    // A = (Z+ + conj Z-)/2 ; B = (Z+ - conj Z-)/(2i)
    const cf A = mk(0.5f * (zpos.x + zneg.x), 0.5f * (zpos.y - zneg.y));
    const cf Bc = mk(0.5f * (zpos.y + zneg.y), -0.5f * (zpos.x - zneg.x));
    cf S = mk(0.f, 0.f);
    if (valid && af < g.k && !(NB == 4 && group_edge_slot(fa, fs))) {
      cf wa = mk(0.f, 0.f), wb = mk(0.f, 0.f);
      if (MODE != 2) {
        if (NB == 1 && wl && af < 128) {     // staged tile (conj already applied), see stage_w
          float a0, a1, a2, a3;
          ld4(reinterpret_cast<const float*>(wl + af * WL_PITCH + 2 * j), a0, a1, a2, a3);
          wa = mk(a0, a1); wb = mk(a2, a3);
        } else {
          if (PF > 0) {
            wa = mk(wv[0], wv[1]); wb = mk(wv[2], wv[3]);
          } else if (fa.wt) {
            float a0, a1, a2, a3;
            ld4(fa.wt + ((size_t)af * g.D + d) * 2, a0, a1, a2, a3);
            wa = mk(a0, a1); wb = mk(a2, a3);
          } else {
            const size_t wo = (size_t)d * g.F + af;
            wa = mk(fa.w_re[wo], fa.w_im[wo]);
            wb = mk(fa.w_re[wo + g.F], fa.w_im[wo + g.F]);
          }
          if (fa.conj_w) { wa = cconj(wa); wb = cconj(wb); }
        }
        const cf ya = cscale(cmul(wa, A), sca), yb = cscale(cmul(wb, Bc), scb);
        if (af == 0 || snyq) {
          S = mk(ya.x * g.inv_n, yb.x * g.inv_n);
          if (MODE == 0 && fa.bias && af == 0) S = mk(S.x + fa.bias[d], S.y + fa.bias[d + 1]);
        } else if (fs > 0) {
          // (Ya + i Yb) / (2N)
          S = mk((ya.x - yb.y) * 0.5f * g.inv_n, (ya.y + yb.x) * 0.5f * g.inv_n);
        } else {
          // (conj Ya + i conj Yb) / (2N)
          S = mk((ya.x + yb.y) * 0.5f * g.inv_n, (-ya.y + yb.x) * 0.5f * g.inv_n);
        }
      }
      if (fs >= 0 || snyq) {
        const size_t xo = (((size_t)b * g.k + af) * g.D + d) * 2;
        if (io_regs<NB, MODE>() && !snyq) {  // spectrum IO happens in prefetch_io / store_io
          constexpr int IM = io_bins<NB>() - 1;
          if (MODE != 1) {
            st.io[2 * (sl & IM)] = A; st.io[2 * (sl & IM) + 1] = Bc;
          } else {
            const cf pa = cscale(cmulc(st.io[2 * (sl & IM)], A), g.inv_n);
            const cf pb = cscale(cmulc(st.io[2 * (sl & IM) + 1], Bc), g.inv_n);
            gsx += wa.x * pa.x + wa.y * pa.y; gsy += wb.x * pb.x + wb.y * pb.y;
            st.io[2 * (sl & IM)] = cscale(pa, sca); st.io[2 * (sl & IM) + 1] = cscale(pb, scb);
            if (af == 0) {
              st1_agent(fa.gb_part + (size_t)b * g.D + d, A.x);
              st1_agent(fa.gb_part + (size_t)b * g.D + d + 1, Bc.x);
            }
          }
        } else if (MODE != 1) {
          if (fa.xk_out) st4(fa.xk_out + xo, A.x, A.y, Bc.x, Bc.y);
        } else {
          float x0, x1, x2, x3;
          if (XQ) { x0 = xv[0]; x1 = xv[1]; x2 = xv[2]; x3 = xv[3]; }
          else ld4(fa.xk_in + xo, x0, x1, x2, x3);
          const cf pa = cscale(cmulc(mk(x0, x1), A), g.inv_n);
          const cf pb = cscale(cmulc(mk(x2, x3), Bc), g.inv_n);
          gsx += wa.x * pa.x + wa.y * pa.y; gsy += wb.x * pb.x + wb.y * pb.y;
          st4(fa.pslab + xo, pa.x * sca, pa.y * sca, pb.x * scb, pb.y * scb);
          if (af == 0) {
            fa.gb_part[(size_t)b * g.D + d] = A.x;
            fa.gb_part[(size_t)b * g.D + d + 1] = Bc.x;
          }
        }
      }
    }
    st.acc[sl] = S;
  }
  if (MODE == 1 && gs) { gs->x += gsx; gs->y += gsy; }
}

// Spectrum IO of the fused kernels where io_regs<NB, MODE>() (no-ops otherwise; the thread's non-negative
// bins are slots 0..7 (one band) or 0..15 (two bands) = bins q + 16 s).  The saved spectrum X is read at the very START of the backward launch
// -- right after the forward launch stored it at its very END -- and the grad slab is stored at the end
// of the backward launch, right before k_gradw reads it.  Measured ~1 % per step;
// the stores also leave the latency-bound unpack phase that way.
template <int NB, int MODE>
SMX_HD void prefetch_io(TState<NB>& st, const Geom& g, const FilterArgs& fa, int b, int d, bool valid,
                        int q) {
  if (!io_regs<NB, MODE>()) return;
#pragma unroll
  for (int s = 0; s < io_bins<NB>(); ++s) {
    const int af = q + 16 * s;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
    if (MODE == 1 && valid && af < g.k)
      ld4(fa.xk_in + (((size_t)b * g.k + af) * g.D + d) * 2, x0, x1, x2, x3);
    st.io[2 * s] = mk(x0, x1); st.io[2 * s + 1] = mk(x2, x3);
  }
}
template <int NB, int MODE>
SMX_HD void store_io(const TState<NB>& st, const Geom& g, const FilterArgs& fa, int b, int d,
                     bool valid, int q) {
  if (!io_regs<NB, MODE>()) return;
  float* dst = MODE == 1 ? fa.pslab : fa.xk_out;
  if (!dst) return;
#pragma unroll
  for (int s = 0; s < io_bins<NB>(); ++s) {
    const int af = q + 16 * s;
    if (valid && af < g.k) {
      float* o = dst + (((size_t)b * g.k + af) * g.D + d) * 2;
      if (MODE == 1 && fa.slab_agent) st4_agent(o, st.io[2 * s].x, st.io[2 * s].y, st.io[2 * s + 1].x, st.io[2 * s + 1].y);
      else st4(o, st.io[2 * s].x, st.io[2 * s].y, st.io[2 * s + 1].x, st.io[2 * s + 1].y);
    }
  }
}

// Same computation as unpack_phase2 with the loads of a whole batch of slots issued first (clamped,
// always-valid addresses; out-of-range bins zeroed by select).  One memory latency per batch instead
// of one per slot: right for k_split_f, where only B*ceil(D/32) workgroups exist and nothing else
// hides the latency (24 -> see DESIGN.md); inside the fused kernels the co-resident workgroup already
// hides it and the extra registers cost more than they save.
template <int NB, int MODE, int ROUND = 0>
SMX_HD void unpack_phase2_batched(TState<NB>& st, const cf* __restrict__ U, const Geom& g,
                                  const FilterArgs& fa, int b, int d, bool valid, int q, int j,
                                  const ZSave<NB>& zsave = ZSave<NB>{}, cf* gs = nullptr,
                                  const cf* __restrict__ wl = nullptr) {
  const int qp = (16 - q) & 15;
  const int dl = valid ? d : g.D - 2;                  // channel pair used for loads
  const bool staged = NB == 1 && wl != nullptr;        // filter slice in LDS, conj already applied (stage_w)
  const bool cj = fa.conj_w && !staged;
  float sca = 1.f, scb = 1.f;
  if (fa.sc) { sca = fa.sc[(size_t)b * g.D + dl]; scb = fa.sc[(size_t)b * g.D + dl + 1]; }
  float gsx = 0.f, gsy = 0.f;
  constexpr int CH = NB == 1 ? 16 : NB == 2 ? 8 : 4;   // slots per batch (register budget: 64 / 128 accumulators live)
  constexpr int S0 = ROUND * UnpackRounds<NB>::SLOTS;
#pragma unroll
  for (int c0 = S0; c0 < S0 + UnpackRounds<NB>::SLOTS; c0 += CH) {
    cf zp[CH];
    float war[CH], wai[CH], wbr[CH], wbi[CH];
    float xs[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int sl = c0 + i;
      const int fs = slot_fs<NB>(q, sl);
      const int af = (fs < 0 ? -fs : fs) + fa.goff;
      const int afc = af < g.k ? af : 0;
      zp[i] = unpack_partner<NB, ROUND>(st, U, q, qp, j, sl, zsave);
      if (MODE != 2) {
        if (staged) {
          ld4(reinterpret_cast<const float*>(wl + (afc < 128 ? afc : 0) * WL_PITCH + 2 * j), war[i], wai[i], wbr[i],
              wbi[i]);
        } else if (fa.wt) {
          ld4(fa.wt + ((size_t)afc * g.D + dl) * 2, war[i], wai[i], wbr[i], wbi[i]);
        } else {
          const size_t wo = (size_t)dl * g.F + afc;
          war[i] = fa.w_re[wo]; wai[i] = fa.w_im[wo];
          wbr[i] = fa.w_re[wo + g.F]; wbi[i] = fa.w_im[wo + g.F];
        }
      }
      const bool pos = slot_pos<NB>(sl) || self_nyquist<NB>(g, fs);   // slots that own a spectrum row
      if (MODE == 1 && pos) {
        const size_t xo = (((size_t)b * g.k + afc) * g.D + dl) * 2;
        ld4(fa.xk_in + xo, xs[i][0], xs[i][1], xs[i][2], xs[i][3]);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int sl = c0 + i;
      const int fs = slot_fs<NB>(q, sl);
      const int af = (fs < 0 ? -fs : fs) + fa.goff;
      const bool ok = valid && af < g.k && !(NB == 4 && group_edge_slot(fa, fs));
      const bool snyq = self_nyquist<NB>(g, fs);
      const bool pos = slot_pos<NB>(sl) || snyq;
      const cf zo = st.acc[sl];
      const cf zpos = fs >= 0 ? zo : zp[i];
      const cf zneg = fs >= 0 ? zp[i] : zo;
      const cf A = mk(0.5f * (zpos.x + zneg.x), 0.5f * (zpos.y - zneg.y));
      const cf Bc = mk(0.5f * (zpos.y + zneg.y), -0.5f * (zpos.x - zneg.x));
      cf S = mk(0.f, 0.f);
      if (MODE != 2) {
        const cf wa = mk(war[i], cj ? -wai[i] : wai[i]);
        const cf wb = mk(wbr[i], cj ? -wbi[i] : wbi[i]);
        const cf ya = cscale(cmul(wa, A), sca), yb = cscale(cmul(wb, Bc), scb);
        const float h = 0.5f * g.inv_n;
        const cf sp = mk((ya.x - yb.y) * h, (ya.y + yb.x) * h);
        const cf sn = mk((ya.x + yb.y) * h, (-ya.y + yb.x) * h);
        cf s0 = mk(ya.x * g.inv_n, yb.x * g.inv_n);
        if (MODE == 0 && fa.bias && af == 0) s0 = mk(s0.x + fa.bias[dl], s0.y + fa.bias[dl + 1]);
        S = (af == 0 || snyq) ? s0 : (fs > 0 ? sp : sn);
        if (!ok) S = mk(0.f, 0.f);
      }
      st.acc[sl] = S;
      if (pos && ok) {
        const size_t xo = (((size_t)b * g.k + af) * g.D + d) * 2;
        if (MODE != 1) {
          if (fa.xk_out) st4(fa.xk_out + xo, A.x, A.y, Bc.x, Bc.y);
        } else {
          const cf pa = cscale(cmulc(mk(xs[i][0], xs[i][1]), A), g.inv_n);
          const cf pb = cscale(cmulc(mk(xs[i][2], xs[i][3]), Bc), g.inv_n);
          {
            const float wai_ = cj ? -wai[i] : wai[i], wbi_ = cj ? -wbi[i] : wbi[i];
            gsx += war[i] * pa.x + wai_ * pa.y; gsy += wbr[i] * pb.x + wbi_ * pb.y;
          }
          st4(fa.pslab + xo, pa.x * sca, pa.y * sca, pb.x * scb, pb.y * scb);
          if (af == 0) {
            fa.gb_part[(size_t)b * g.D + d] = A.x;
            fa.gb_part[(size_t)b * g.D + d + 1] = Bc.x;
          }
        }
      }
    }
  }
  if (MODE == 1 && gs) { gs->x += gsx; gs->y += gsy; }
}

}  // namespace smx

// =====================================================================================================
// Four-step path for the FULL spectrum of long transforms (N = 256 L, 5 <= L <= 16 or L = 32; more than 512 bins;
// L = 64, 128, 256: the two-level column transform further down)
//   (A) per residue r: the tile's 256-point spectrum, twiddled by w_N^{fu r}, goes to a workspace
//   (F) per pair of columns {fu, 256 - fu}: an L-point transform across the residues gives the bins
//       fu + 256 f2 -- a set closed under f -> -f, so ONE thread unpacks, filters and repacks all of them in
//       registers (no exchange between threads) -- and the inverse L-point transform goes back in place
//   (B) per residue r: inverse 256-point transform of the filtered tile, store
// x and y stream once; the packed spectrum makes one round trip through the workspace (same bytes as x
// for an unpadded transform).  Replaces the band groups (one pass over x and y per 512 bins + edge-bin
// passes) for these lengths.
// Workspace layout: ws[((wg L + r) 16 + s) 256 + tid], wg = b ndt + dt, tid = q 16 + j, bin fu = q + 16 s.
// =====================================================================================================
namespace smx {

// (A) second half of the tile transform, result to the workspace instead of accumulators
SMX_HD void fwd_phase2_out(const cf* __restrict__ E, const cf* __restrict__ bt_r, int t, int j,
                           cf* __restrict__ dst) {
  cf e[16];
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) e[t2] = E[(t2 * 16 + t) * 16 + j];
  fft16<-1>(e);
#pragma unroll
  for (int s = 0; s < 16; ++s) dst[s * TPB] = cmul(ld_uniform(bt_r, s + BT_HALF), e[s]);
}
// (B) first half of the inverse tile transform from values loaded out of the workspace
SMX_HD void inv_phase1_in(cf (&v)[16], const cf* __restrict__ bt_r, cf* __restrict__ E, int q, int j) {
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = cmulc(v[s], ld_uniform(bt_r, s + BT_HALF));
  fft16<+1>(v);
#pragma unroll
  for (int p = 0; p < 16; ++p) E[(q * 16 + p) * 16 + j] = v[p];
}

// natural-order L-point DFT across the residues; w_L^k = tw[STRIDE k] (tw = w_N^n table, N = 256 L at STRIDE 256)
template <int SGN, int L, int STRIDE = 256>
SMX_HD void fft_residues(cf (&a)[L], const cf* __restrict__ tw) {
  if constexpr (L == 8) {
    fft8<SGN>(a);
  } else if constexpr (L == 16) {
    fft16<SGN>(a);
  } else if constexpr (L > 16 && L % 2 == 0) {
    // even L above 16 (18 ... 32): one radix-2 step over two half-length transforms
    constexpr int H = L / 2;
    cf ev[H], od[H];
#pragma unroll
    for (int i = 0; i < H; ++i) { ev[i] = a[2 * i]; od[i] = a[2 * i + 1]; }
    fft_residues<SGN, H, 2 * STRIDE>(ev, tw);
    fft_residues<SGN, H, 2 * STRIDE>(od, tw);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const cf w = tw[STRIDE * k];                                // w_L^k
      const cf o = (SGN < 0) ? cmul(od[k], w) : cmulc(od[k], w);
      a[k] = cadd(ev[k], o);
      a[k + H] = csub(ev[k], o);
    }
  } else if constexpr (L == 4) {
    radix4<SGN>(a[0], a[1], a[2], a[3]);
  } else if constexpr (L == 2) {
    const cf s0 = cadd(a[0], a[1]), d0 = csub(a[0], a[1]);
    a[0] = s0; a[1] = d0;
  } else {
    // any other L <= 16 (N = 768, 1280, 1536, ... 3840): the L x L product with w_L^m = tw[STRIDE (m mod L)]
    cf o[L];
#pragma unroll
    for (int f2 = 0; f2 < L; ++f2) {
      cf acc = a[0];
#pragma unroll
      for (int r = 1; r < L; ++r) {
        const cf w = tw[STRIDE * ((r * f2) % L)];
        acc = (SGN < 0) ? cfma(acc, a[r], w) : cfmac(acc, a[r], w);
      }
      o[f2] = acc;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) a[i] = o[i];
  }
}

// ---- one bin pair of a column unit -----------------------------------------------------------------------
// What a column thread needs to know about its place; fs_pair is the unpack / filter / repack of ONE pair
// (bin f = u + 256 f2 held in z0, its mirror image N - f held in z1), shared by the register-resident column
// kernel (L <= 32) and the two-level one (L = 64, 128, 256).
struct FsCtx {
  int b, d, dl, u;
  bool valid, one_col;
  float sca, scb;
};
SMX_HD FsCtx fs_ctx(const Geom& g, const FilterArgs& fa, int b, int d, bool valid, int u) {
  FsCtx c;
  c.b = b; c.d = d; c.u = u; c.valid = valid;
  c.dl = valid ? d : g.D - 2;
  c.one_col = (u == 0 || u == 128);
  c.sca = 1.f; c.scb = 1.f;
  if (fa.sc) { c.sca = fa.sc[(size_t)b * g.D + c.dl]; c.scb = fa.sc[(size_t)b * g.D + c.dl + 1]; }
  return c;
}
SMX_HD int fs_bin(const Geom& g, int u, int f2, bool& pos) {
  const int f = u + 256 * f2;
  pos = 2 * f <= g.N;
  return pos ? f : g.N - f;
}
// the 16-byte rows this pair reads: filter (MODE != 2), saved spectrum (MODE == 1)
template <int MODE>
SMX_HD void fs_pair_issue(const Geom& g, const FilterArgs& fa, const FsCtx& c, int f2, float (&wq)[4], float (&xq)[4]) {
  bool pos;
  const int af = fs_bin(g, c.u, f2, pos);
  const int afc = af < g.k ? af : 0;
  if (MODE != 2) {
    if (fa.wt) {
      ld4(fa.wt + ((size_t)afc * g.D + c.dl) * 2, wq[0], wq[1], wq[2], wq[3]);
    } else {
      const size_t wo = (size_t)c.dl * g.F + afc;
      wq[0] = fa.w_re[wo]; wq[1] = fa.w_im[wo];
      wq[2] = fa.w_re[wo + g.F]; wq[3] = fa.w_im[wo + g.F];
    }
  }
  if (MODE == 1) ld4(fa.xk_in + (((size_t)c.b * g.k + afc) * g.D + c.dl) * 2, xq[0], xq[1], xq[2], xq[3]);
}
// pacc2 (backward, optional): instead of storing this batch row's slab row, add it to pacc2[0..1] (and the
// grad_bias term to *gbacc) -- k_fs_f_grouped walks a GROUP of batch rows with one thread
template <int MODE>
SMX_HD void fs_pair(const Geom& g, const FilterArgs& fa, const FsCtx& c, int f2, const float (&wv)[4],
                    const float (&xv)[4], cf& z0, cf& z1, float& gsx, float& gsy, cf* pacc2 = nullptr,
                    cf* gbacc = nullptr) {
  bool pos;
  const int af = fs_bin(g, c.u, f2, pos);
  const int d = c.d;
  const bool self = af == 0 || 2 * af == g.N;                 // DC / Nyquist: their own mirror image
  const cf zpos = pos ? z0 : z1, zneg = pos ? z1 : z0;
  const cf A = mk(0.5f * (zpos.x + zneg.x), 0.5f * (zpos.y - zneg.y));
  const cf Bc = mk(0.5f * (zpos.y + zneg.y), -0.5f * (zpos.x - zneg.x));
  cf Spos = mk(0.f, 0.f), Sneg = mk(0.f, 0.f);
  if (c.valid && af < g.k) {
    cf wa = mk(wv[0], wv[1]), wb = mk(wv[2], wv[3]);
    if (fa.conj_w) { wa = cconj(wa); wb = cconj(wb); }
    if (MODE != 2) {
      const cf ya = cscale(cmul(wa, A), c.sca), yb = cscale(cmul(wb, Bc), c.scb);
      if (self) {
        Spos = mk(ya.x * g.inv_n, yb.x * g.inv_n);
        if (MODE == 0 && fa.bias && af == 0) Spos = mk(Spos.x + fa.bias[d], Spos.y + fa.bias[d + 1]);
        Sneg = Spos;
      } else {
        const float h = 0.5f * g.inv_n;
        Spos = mk((ya.x - yb.y) * h, (ya.y + yb.x) * h);       // (Ya + i Yb) / (2N)
        Sneg = mk((ya.x + yb.y) * h, (-ya.y + yb.x) * h);      // (conj Ya + i conj Yb) / (2N)
      }
    }
    if (pos || !c.one_col) {          // a single-column unit meets every pair from both ends: IO once
      const size_t xo = (((size_t)c.b * g.k + af) * g.D + d) * 2;
      if (MODE != 1) {
        if (fa.xk_out) st4(fa.xk_out + xo, A.x, A.y, Bc.x, Bc.y);
      } else {
        const cf pa = cscale(cmulc(mk(xv[0], xv[1]), A), g.inv_n);
        const cf pb = cscale(cmulc(mk(xv[2], xv[3]), Bc), g.inv_n);
        gsx += wa.x * pa.x + wa.y * pa.y; gsy += wb.x * pb.x + wb.y * pb.y;
        if (pacc2) {
          pacc2[0] = cadd(pacc2[0], cscale(pa, c.sca));
          pacc2[1] = cadd(pacc2[1], cscale(pb, c.scb));
          if (af == 0) *gbacc = cadd(*gbacc, mk(A.x, Bc.x));
        } else {
          st4(fa.pslab + xo, pa.x * c.sca, pa.y * c.sca, pb.x * c.scb, pb.y * c.scb);
          if (af == 0) {
            fa.gb_part[(size_t)c.b * g.D + d] = A.x;
            fa.gb_part[(size_t)c.b * g.D + d + 1] = Bc.x;
          }
        }
      }
    }
  }
  z0 = pos ? Spos : Sneg;
  z1 = pos ? Sneg : Spos;
}

// (F) one thread: columns fu = u and 256 - u of one channel pair.  MODE as in unpack_phase2.
// pacc (backward, optional): instead of storing this batch row's slab rows, add them to pacc[f2][channel]
// (and the grad_bias term to *gbacc) -- the caller walks a GROUP of batch rows with one thread and stores the
// sums once (fs_store_slab), so the slab and the k_gradw pass shrink from B rows to the number of groups.
template <int L, int MODE>
SMX_HD void fs_columns(cf* __restrict__ wsb, const Geom& g, const FilterArgs& fa,
                       const cf* __restrict__ tw, int b, int d, bool valid, int u, int j, cf* gs = nullptr,
                       cf (*pacc)[2] = nullptr, cf* gbacc = nullptr) {
  const int fum = (256 - u) & 255;
  const int offp = ((u >> 4) * 256) + (u & 15) * 16 + j;
  const int offm = ((fum >> 4) * 256) + (fum & 15) * 16 + j;
  const bool one_col = (u == 0 || u == 128);
  cf zp[L], zm[L];
#pragma unroll
  for (int r = 0; r < L; ++r) { zp[r] = wsb[(size_t)r * EX + offp]; zm[r] = wsb[(size_t)r * EX + offm]; }
  fft_residues<-1, L>(zp, tw);
  fft_residues<-1, L>(zm, tw);
  if (MODE == 3) {
    // complex sequence FFT (reference frequency_ops.py:201): the channel pair IS one complex channel, the
    // packed spectrum is the answer -- bins u + 256 f2 and (256 - u) + 256 f2 go straight to out (B, N, D)
    if (valid) {
      float* o = fa.xk_out + (size_t)b * g.N * g.D + d;
#pragma unroll
      for (int f2 = 0; f2 < L; ++f2) {
        float* p = o + (size_t)(u + 256 * f2) * g.D;
        p[0] = zp[f2].x; p[1] = zp[f2].y;
      }
      if (!one_col) {
#pragma unroll
        for (int f2 = 0; f2 < L; ++f2) {
          float* p = o + (size_t)(fum + 256 * f2) * g.D;
          p[0] = zm[f2].x; p[1] = zm[f2].y;
        }
      }
    }
    return;
  }
  if (u == 0) {                       // column 0 mirrors into itself: -(256 f2) = 256 ((L - f2) mod L)
#pragma unroll
    for (int i = 0; i < L; ++i) zm[i] = zp[(i + 1) % L];
  }
  const FsCtx c = fs_ctx(g, fa, b, d, valid, u);
  constexpr int PF = 4;
  float wq[PF][4], xq[PF][4];
  float gsx = 0.f, gsy = 0.f;
#pragma unroll
  for (int i = 0; i < PF && i < L; ++i) fs_pair_issue<MODE>(g, fa, c, i, wq[i], xq[i]);
#pragma unroll
  for (int f2 = 0; f2 < L; ++f2) {
    const int ring = f2 % PF;
    float wv[4], xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { wv[e] = wq[ring][e]; xv[e] = xq[ring][e]; }
    if (f2 + PF < L) fs_pair_issue<MODE>(g, fa, c, f2 + PF, wq[ring], xq[ring]);
    fs_pair<MODE>(g, fa, c, f2, wv, xv, zp[f2], zm[L - 1 - f2], gsx, gsy, pacc ? pacc[f2] : nullptr, gbacc);
  }
  if (MODE == 1 && gs) { gs->x += gsx; gs->y += gsy; }
  if (MODE == 2) return;
  fft_residues<+1, L>(zp, tw);
#pragma unroll
  for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offp] = zp[r];
  if (!one_col) {
    fft_residues<+1, L>(zm, tw);
#pragma unroll
    for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offm] = zm[r];
  }
}

// the sums of a batch group (fs_columns with pacc) -> row `grp` of the (groups, k, D) slab
template <int L>
SMX_HD void fs_store_slab(const cf (*pacc)[2], cf gbacc, const Geom& g, const FilterArgs& fa, int grp, int d,
                          bool valid, int u) {
  if (!valid) return;
  const bool one_col = (u == 0 || u == 128);
#pragma unroll
  for (int f2 = 0; f2 < L; ++f2) {
    const int f = u + 256 * f2;
    const bool pos = 2 * f <= g.N;
    const int af = pos ? f : g.N - f;
    if (af < g.k && (pos || !one_col)) {
      st4(fa.pslab + (((size_t)grp * g.k + af) * g.D + d) * 2, pacc[f2][0].x, pacc[f2][0].y, pacc[f2][1].x,
          pacc[f2][1].y);
      if (af == 0) {
        fa.gb_part[(size_t)grp * g.D + d] = gbacc.x;
        fa.gb_part[(size_t)grp * g.D + d + 1] = gbacc.y;
      }
    }
  }
}

// (F) of the synthesis (smx_irfft_ex): the columns u and 256 - u of the packed spectrum come from the rows of a
// given one-sided spectrum instead of from tile spectra; the inverse L-point transforms leave them in the
// workspace for k_fs_b.
template <int L>
SMX_HD void fs_synth_columns(cf* __restrict__ wsb, const Geom& g, const FilterArgs& fa,
                             const cf* __restrict__ tw, int b, int d, bool valid, int u, int j) {
  const int fum = (256 - u) & 255;
  const int offp = ((u >> 4) * 256) + (u & 15) * 16 + j;
  const int offm = ((fum >> 4) * 256) + (fum & 15) * 16 + j;
  const bool one_col = (u == 0 || u == 128);
  const int dl = valid ? d : g.D - 2;
  cf zp[L], zm[L];
  constexpr int CH = L < 8 ? L : 8;
#pragma unroll
  for (int c0 = 0; c0 < L; c0 += CH) {
    float r[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (c0 + i < L) {
        const int f = u + 256 * (c0 + i);
        const int af = 2 * f <= g.N ? f : g.N - f;
        const int afc = af < g.k ? af : 0;
        ld4(fa.xk_in + (((size_t)b * g.k + afc) * g.D + dl) * 2, r[i][0], r[i][1], r[i][2], r[i][3]);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (c0 + i < L) {
        const int f2 = c0 + i, f = u + 256 * f2;
        const bool pos = 2 * f <= g.N;
        const int af = pos ? f : g.N - f;
        cf sp, sn;
        synth_pair(r[i], af == 0 || 2 * af == g.N, fa.sp_scale, fa.sp_herm != 0, sp, sn);
        if (!(valid && af < g.k)) { sp = mk(0.f, 0.f); sn = sp; }
        zp[f2] = pos ? sp : sn;
        zm[L - 1 - f2] = pos ? sn : sp;
      }
    }
  }
  fft_residues<+1, L>(zp, tw);
#pragma unroll
  for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offp] = zp[r];
  if (!one_col) {
    fft_residues<+1, L>(zm, tw);
#pragma unroll
    for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offm] = zm[r];
  }
}

// ---- two-level column transform: L = 16 L2 residues, L2 in {2, 4, 8, 16} (N = 8192 ... 65536) -----------------
// 2 x L complex per column pair no longer fit one thread, so L2 threads share it: thread t2 holds the residues
// r = r1 L2 + t2 (16 of each column), transforms them over r1 (fft16), multiplies by w_L^{t2 q1} and publishes
// them in LDS; after the barrier it gathers, for its 16 / L2 values of q1, the L2 entries of sub-array q1 and
// transforms over t2: bin f2 = q1 + 16 q2.  The mirror image of bin u + 256 f2 is (256 - u) + 256 (L - 1 - f2),
// L - 1 - f2 = (15 - q1) + 16 (L2 - 1 - q2): the same thread gathers sub-array 15 - q1 of the other column, so
// every pair meets in one thread's registers exactly as in fs_columns (fs_pair).  Column 0 mirrors into itself,
// -(256 f2) = 256 ((L - f2) mod L): sub-array (16 - q1) mod 16 of the same column.  The way back mirrors it.
// A block = 16 / L2 column units x L2 x 16 channel pairs = 256 threads.  The two columns go through the SAME
// 32 KiB of LDS one after the other, X[unit][q1][t2][j] (conflict-free both ways): half the LDS of publishing
// both at once, i.e. four workgroups per CU instead of two where the registers allow it, for three more barriers.
// Round 3: L = L1 L2 with ANY first-level length L1 <= 16 (5 ... 16; the default 16 is the round-2 scheme) -- tile
// counts 36, 40, ... 60 (L2 = 4), 72 ... 120 (L2 = 8), 144 ... 240 (L2 = 16) leave the band groups.  Residue
// r = r1 L2 + t2 with r1 < L1, bin f2 = q1 + L1 q2, mirror image L - 1 - f2 = (L1 - 1 - q1) + L1 (L2 - 1 - q2); the
// first level is fft_residues<L1> with w_L1 = w_N^{256 L2}.  A thread takes the q1 values t2 NQ ... t2 NQ + NQ - 1,
// NQ = ceil(L1 / L2); those >= L1 (L1 not a multiple of L2) are padding: their gathers read unused LDS rows, their
// pairs are skipped, their way back writes unused rows.
template <int L2>
SMX_HD int big_idx(int ul, int q1, int t2, int j) { return (((ul * 16 + q1) * L2 + t2) * 16) + j; }
struct BigState { cf zp[16], zm[16]; };
SMX_HD int big_off(int u, int j) { return ((u >> 4) * 256) + (u & 15) * 16 + j; }
template <int L1 = 16>
SMX_HD int big_qm(int u, int q1) { return u == 0 ? ((L1 - q1) % L1) : L1 - 1 - q1; }
template <int L2, int L1 = 16> SMX_HD constexpr int big_nq() { return (L1 + L2 - 1) / L2; }

// the residues of thread t2, both columns (all loads in flight before the first exchange)
template <int L2, int L1 = 16>
SMX_HD void fsb_load(BigState& st, const cf* __restrict__ wsb, int u, int t2, int j) {
  const int offp = big_off(u, j), offm = big_off((256 - u) & 255, j);
#pragma unroll
  for (int r1 = 0; r1 < L1; ++r1) {
    st.zp[r1] = wsb[(size_t)(r1 * L2 + t2) * EX + offp];
    st.zm[r1] = wsb[(size_t)(r1 * L2 + t2) * EX + offm];
  }
}
// first-level transform over r1 (w_L1^k = w_N^{256 L2 k}) on the first L1 entries of z
template <int SGN, int L2, int L1>
SMX_HD void big_fft_l1(cf (&z)[16], const cf* __restrict__ tw) {
  if constexpr (L1 == 16) {
    fft16<SGN>(z);
  } else {
    cf a[L1];
#pragma unroll
    for (int i = 0; i < L1; ++i) a[i] = z[i];
    fft_residues<SGN, L1, 256 * L2>(a, tw);
#pragma unroll
    for (int i = 0; i < L1; ++i) z[i] = a[i];
  }
}
// forward, step 1 (one column): transform over r1 -> twiddle w_L^{t2 q1} -> LDS
template <int L2, int L1 = 16>
SMX_HD void fsb_pub(cf (&z)[16], const cf* __restrict__ tw, cf* __restrict__ X, int ul, int t2, int j) {
  big_fft_l1<-1, L2, L1>(z, tw);
#pragma unroll
  for (int q1 = 0; q1 < L1; ++q1) X[big_idx<L2>(ul, q1, t2, j)] = cmul(z[q1], tw[256 * (t2 * q1)]);
}
// forward, step 2 (one column): gather the sub-arrays of this thread's q1 values, transform over t2.
// MIRROR = false: z[a L2 + q2] = bin q1 + L1 q2 of column u.  MIRROR = true (the other column is in X): entry q2
// of the partner sub-array big_qm(u, q1), so that the mirror image of zp[a L2 + q2] is zm[a L2 + L2 - 1 - q2].
template <int L2, bool MIRROR, int L1 = 16>
SMX_HD void fsb_gather(cf (&z)[16], const cf* __restrict__ X, const cf* __restrict__ tw, int u, int ul, int t2,
                       int j) {
  constexpr int NQ = big_nq<L2, L1>();
#pragma unroll
  for (int a = 0; a < NQ; ++a) {
    const int q1 = t2 * NQ + a;
    const int q1c = (L1 % L2 == 0 || q1 < L1) ? q1 : 0;          // padding slot: any row, the result is not used
    const int qs = MIRROR ? big_qm<L1>(u, q1c) : q1c;
    cf t[L2];
#pragma unroll
    for (int i = 0; i < L2; ++i) t[i] = X[big_idx<L2>(ul, qs, i, j)];
    fft_residues<-1, L2>(t, tw);
    // the multiples of 256 L1 of column 0 mirror into themselves: -(L1 q2) = L1 ((L2 - q2) mod L2)
    const bool rot = MIRROR && u == 0 && q1 == 0;
#pragma unroll
    for (int i = 0; i < L2; ++i) z[a * L2 + i] = rot ? t[(i + 1) % L2] : t[i];
  }
}

// unpack / filter / repack of the thread's pairs (MODE 0, 1, 2) or the packed bins straight out (MODE 3)
template <int L2, int MODE, int L1 = 16>
SMX_HD void fsb_pairs(BigState& st, const Geom& g, const FilterArgs& fa, int b, int d, bool valid, int u, int t2,
                      cf* gs) {
  constexpr int NQ = big_nq<L2, L1>();
  constexpr int NS = NQ * L2;                                     // slots of a thread (16 when L2 divides L1)
  constexpr bool PADDED = L1 % L2 != 0;
  if (MODE == 3) {
    if (valid) {
      float* o = fa.xk_out + (size_t)b * g.N * g.D + d;
      const int fum = (256 - u) & 255;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int q1 = t2 * NQ + i / L2, q2 = i % L2;
        if (PADDED && q1 >= L1) continue;
        float* p = o + (size_t)(u + 256 * (q1 + L1 * q2)) * g.D;
        p[0] = st.zp[i].x; p[1] = st.zp[i].y;
        if (u != 0 && u != 128) {
          float* pm = o + (size_t)(fum + 256 * ((L1 - 1 - q1) + L1 * q2)) * g.D;
          pm[0] = st.zm[i].x; pm[1] = st.zm[i].y;
        }
      }
    }
    return;
  }
  const FsCtx c = fs_ctx(g, fa, b, d, valid, u);
  float gsx = 0.f, gsy = 0.f;
  auto f2_of = [&](int i) { return t2 * NQ + i / L2 + L1 * (i % L2); };
  if constexpr (!PADDED) {
    constexpr int PF = 4;
    float wq[PF][4], xq[PF][4];
#pragma unroll
    for (int i = 0; i < PF; ++i) fs_pair_issue<MODE>(g, fa, c, f2_of(i), wq[i], xq[i]);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int ring = i % PF;
      float wv[4], xv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { wv[e] = wq[ring][e]; xv[e] = xq[ring][e]; }
      if (i + PF < NS) fs_pair_issue<MODE>(g, fa, c, f2_of(i + PF), wq[ring], xq[ring]);
      fs_pair<MODE>(g, fa, c, f2_of(i), wv, xv, st.zp[i], st.zm[(i / L2) * L2 + (L2 - 1 - i % L2)], gsx, gsy);
    }
  } else {
    // the thread's last q1 values may be padding (uniform per group of L2 slots): no loads, no pair for those
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
      if (t2 * NQ + a >= L1) continue;
      float wq[L2][4], xq[L2][4];
#pragma unroll
      for (int i2 = 0; i2 < L2; ++i2) fs_pair_issue<MODE>(g, fa, c, f2_of(a * L2 + i2), wq[i2], xq[i2]);
#pragma unroll
      for (int i2 = 0; i2 < L2; ++i2) {
        const int i = a * L2 + i2;
        fs_pair<MODE>(g, fa, c, f2_of(i), wq[i2], xq[i2], st.zp[i], st.zm[a * L2 + (L2 - 1 - i2)], gsx, gsy);
      }
    }
  }
  if (MODE == 1 && gs) { gs->x += gsx; gs->y += gsy; }
}
// synthesis (smx_irfft_ex): the thread's pairs from the rows of a given one-sided spectrum
template <int L2, int L1 = 16>
SMX_HD void fsb_synth(BigState& st, const Geom& g, const FilterArgs& fa, int b, int d, bool valid, int u, int t2) {
  constexpr int NQ = big_nq<L2, L1>();
  constexpr int NS = NQ * L2;
  constexpr int CHK = NS < 8 ? NS : 8;
  const int dl = valid ? d : g.D - 2;
#pragma unroll
  for (int c0 = 0; c0 < NS; c0 += CHK) {
    float r[CHK][4];
#pragma unroll
    for (int e = 0; e < CHK; ++e) {
      const int i = c0 + e;
      if (i < NS) {
        const int q1 = t2 * NQ + i / L2;
        const int f = u + 256 * ((q1 < L1 ? q1 : 0) + L1 * (i % L2));
        const int af = 2 * f <= g.N ? f : g.N - f;
        const int afc = af < g.k ? af : 0;
        ld4(fa.xk_in + (((size_t)b * g.k + afc) * g.D + dl) * 2, r[e][0], r[e][1], r[e][2], r[e][3]);
      }
    }
#pragma unroll
    for (int e = 0; e < CHK; ++e) {
      const int i = c0 + e;
      if (i < NS) {
        const int q1 = t2 * NQ + i / L2;
        const int f = u + 256 * ((q1 < L1 ? q1 : 0) + L1 * (i % L2));
        const bool pos = 2 * f <= g.N;
        const int af = pos ? f : g.N - f;
        cf sp, sn;
        synth_pair(r[e], af == 0 || 2 * af == g.N, fa.sp_scale, fa.sp_herm != 0, sp, sn);
        if (!(valid && af < g.k)) { sp = mk(0.f, 0.f); sn = sp; }
        st.zp[i] = pos ? sp : sn;
        st.zm[(i / L2) * L2 + (L2 - 1 - i % L2)] = pos ? sn : sp;
      }
    }
  }
}
// inverse, step 1 (one column): transform over q2, conjugate twiddle, publish.  MIRROR: the values belong to
// sub-array L1 - 1 - q1 of the other column.
template <int L2, bool MIRROR, int L1 = 16>
SMX_HD void fsb_unpub(const cf (&z)[16], cf* __restrict__ X, const cf* __restrict__ tw, int ul, int t2, int j) {
  constexpr int NQ = big_nq<L2, L1>();
#pragma unroll
  for (int a = 0; a < NQ; ++a) {
    const int q1 = t2 * NQ + a;
    // padding slots publish into the unused rows L1 ... 15 (q1 >= L1; mirrored: the row of q1 itself is unused too)
    const bool real = L1 % L2 == 0 || q1 < L1;
    const int qs = (MIRROR && real) ? L1 - 1 - q1 : q1;
    const int qt = real ? qs : 0;                                   // (table index of a padding slot: anything valid)
    cf t[L2];
#pragma unroll
    for (int i = 0; i < L2; ++i) t[i] = z[a * L2 + i];
    fft_residues<+1, L2>(t, tw);
#pragma unroll
    for (int i = 0; i < L2; ++i) X[big_idx<L2>(ul, qs, i, j)] = cmulc(t[i], tw[256 * (i * qt)]);
  }
}
// inverse, step 2 (one column): gather over q1, transform to the residues of thread t2, store at column `col`
template <int L2, int L1 = 16>
SMX_HD void fsb_ungather(cf (&z)[16], cf* __restrict__ wsb, const cf* __restrict__ X, const cf* __restrict__ tw,
                         int col, int ul, int t2, int j) {
  const int off = big_off(col, j);
#pragma unroll
  for (int q1 = 0; q1 < L1; ++q1) z[q1] = X[big_idx<L2>(ul, q1, t2, j)];
  big_fft_l1<+1, L2, L1>(z, tw);
#pragma unroll
  for (int r1 = 0; r1 < L1; ++r1) wsb[(size_t)(r1 * L2 + t2) * EX + off] = z[r1];
}

}  // namespace smx

// =====================================================================================================
// Rank-one filter on the four-step path: the causal FFT convolution of fft_lm.FixedSpectralBlock
// (reference fft_lm/train_fixed_full.py:507-555)
//     y[b, n, c] = s[b, c] * irfft( rfft(zero-pad(x[b, :, c]), N) * H, N )[n],   n < rows
// Every channel sees the SAME complex response H[f] (kernel spectrum x frequency gate x cutoff mask) and a
// real factor s[b, c] (gain x context gate).  A real kernel convolves the packed pair z = a + i b as it
// convolves a and b, so the packed spectrum is multiplied by the Hermitian extension of H directly -- no
// unpack, no (D, F) filter, no saved one-sided spectrum, no gradient slab:
//   forward   (A) tile spectra of x -> xs (kept for backward);  (F) Y = Z Hfull / N per column, back to ws;
//             (B) inverse tiles, s applied to the two channels at the store
//   backward  (A) tile spectra of g -> ws;  (F) per column pair, with the columns of xs beside them:
//             grad_x:  Zg conj(Hfull) / N back to ws ((B) stores s * ...);
//             P[f] += Zg[f] (sigma conj(Zx[f]) + delta Zx[-f]),  sigma/delta = (s_a +/- s_b) / 2: its Hermitian
//                      part, summed over (b, pairs), is sum_c s_c conj(X_c) G_c -> grad_H (host: c_f / N, section
//                      comment in fixed_spectral.py);
//             R1 += Re(Zg conj(W)), R2 += Re(Zg[f] W[-f]),  W = Hfull Zx:  (R1 +/- R2) / (2N) = sum_n g y0 of the
//                      two channels = grad_s.
// =====================================================================================================
namespace smx {

struct ConvArgs {
  const float* h_re;      // (N/2 + 1) response, real / imaginary parts (imaginary parts of DC / Nyquist ignored,
  const float* h_im;      //  as torch.fft.irfft does)
  const float* sc;        // (B, D) or null
  const cf* xs;           // backward: tile spectra of x saved by forward, workspace layout
  cf* p_part;             // backward: [B*ndt][N] partial sums of P over the workgroup's 16 channel pairs
  cf* r_part;             // backward: [B*ndt][9][16] partial (R1, R2) of the column-unit blocks
};

SMX_HD cf conv_hfull(const ConvArgs& ca, int f, int N) {
  const bool up = 2 * f <= N;
  const int i = up ? f : N - f;
  const float im = (i == 0 || 2 * i == N) ? 0.f : ca.h_im[i];
  return mk(ca.h_re[i], up ? im : -im);
}

// forward (DIR = 0): columns u and 256 - u of ws  *=  Hfull / N  (in the residue-transformed domain)
// backward (DIR = 1): the same with conj(Hfull) on the columns of g, plus the P / R sums against xs
// src: where the columns are read (may be wsb itself); wsb: where the filtered columns go
template <int L, int DIR>
SMX_HD void fs_conv_columns(const cf* src, cf* wsb, const cf* __restrict__ xsb, const Geom& g,
                            const ConvArgs& ca, const cf* __restrict__ tw, int b, int d, bool valid, int u,
                            int j, cf* __restrict__ pp, cf* __restrict__ pm, cf* rr) {
  const int fum = (256 - u) & 255;
  const int offp = ((u >> 4) * 256) + (u & 15) * 16 + j;
  const int offm = ((fum >> 4) * 256) + (fum & 15) * 16 + j;
  const bool one_col = (u == 0 || u == 128);
  cf zp[L], zm[L];
#pragma unroll
  for (int r = 0; r < L; ++r) { zp[r] = src[(size_t)r * EX + offp]; zm[r] = src[(size_t)r * EX + offm]; }
  fft_residues<-1, L>(zp, tw);
  fft_residues<-1, L>(zm, tw);
  cf xp[DIR ? L : 1], xm[DIR ? L : 1];
  if constexpr (DIR == 1) {
#pragma unroll
    for (int r = 0; r < L; ++r) { xp[r] = xsb[(size_t)r * EX + offp]; xm[r] = xsb[(size_t)r * EX + offm]; }
    fft_residues<-1, L>(xp, tw);
    fft_residues<-1, L>(xm, tw);
  }
  float sig = 1.f, del = 0.f;
  if (DIR == 1 && ca.sc) {
    const int dl = valid ? d : g.D - 2;
    const float sa = ca.sc[(size_t)b * g.D + dl], sb = ca.sc[(size_t)b * g.D + dl + 1];
    sig = 0.5f * (sa + sb); del = 0.5f * (sa - sb);
  }
  float r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int f2 = 0; f2 < L; ++f2) {
    const int fp = u + 256 * f2, fm = fum + 256 * f2;
    const cf hp = conv_hfull(ca, fp, g.N), hm = conv_hfull(ca, fm, g.N);
    if constexpr (DIR == 1) {
      // mirror images: -(u + 256 f2) = (256 - u) + 256 (L - 1 - f2); column 0 mirrors into itself
      const int mi = (u == 0) ? (L - f2) % L : L - 1 - f2;
      const cf xneg_p = (u == 0) ? xp[mi] : xm[mi];            // Zx[-fp]
      const cf gp = zp[f2];
      pp[f2] = cmul(gp, cadd(cscale(cconj(xp[f2]), sig), cscale(xneg_p, del)));
      const cf wp = cmul(hp, xp[f2]);                          // W[fp]
      const cf wneg_p = cmulc(xneg_p, hp);                     // W[-fp] = conj(Hfull[fp]) Zx[-fp]
      r1 += gp.x * wp.x + gp.y * wp.y;                         // Re(Zg conj W)
      r2 += gp.x * wneg_p.x - gp.y * wneg_p.y;                 // Re(Zg W[-f])
      if (!one_col) {
        const cf xneg_m = xp[mi];                              // Zx[-fm]: column u, index L - 1 - f2
        const cf gm = zm[f2];
        pm[f2] = cmul(gm, cadd(cscale(cconj(xm[f2]), sig), cscale(xneg_m, del)));
        const cf wm = cmul(hm, xm[f2]);
        const cf wneg_m = cmulc(xneg_m, hm);
        r1 += gm.x * wm.x + gm.y * wm.y;
        r2 += gm.x * wneg_m.x - gm.y * wneg_m.y;
      }
    }
    const cf hpe = cscale(DIR ? cconj(hp) : hp, g.inv_n), hme = cscale(DIR ? cconj(hm) : hm, g.inv_n);
    zp[f2] = valid ? cmul(zp[f2], hpe) : mk(0.f, 0.f);
    zm[f2] = valid ? cmul(zm[f2], hme) : mk(0.f, 0.f);
  }
  if constexpr (DIR == 1) {
    if (rr) *rr = valid ? mk(r1, r2) : mk(0.f, 0.f);
    if (!valid) {
#pragma unroll
      for (int f2 = 0; f2 < L; ++f2) { pp[f2] = mk(0.f, 0.f); pm[f2] = mk(0.f, 0.f); }
    } else if (one_col) {
#pragma unroll
      for (int f2 = 0; f2 < L; ++f2) pm[f2] = mk(0.f, 0.f);
    }
  }
  fft_residues<+1, L>(zp, tw);
#pragma unroll
  for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offp] = zp[r];
  if (!one_col) {
    fft_residues<+1, L>(zm, tw);
#pragma unroll
    for (int r = 0; r < L; ++r) wsb[(size_t)r * EX + offm] = zm[r];
  }
}

// ---- rank-one filter on the two-level columns (L = 32, 64, 128, 256: n_fft 8192 ... 65536) -------------------
// The exchanges are those of the generic two-level transform (fsb_load / fsb_pub / fsb_gather and back); after
// the gathers a thread holds, at index i = a L2 + q2, bin fp = u + 256 (q1 + 16 q2) of column u in zp[i] and bin
// fm = (256 - u) + 256 ((15 - q1) + 16 q2) of the mirror column in zm[i], and the mirror image of zp[i] sits in
// zm[big_pi(i)] (of zm[i] in zp[big_pi(i)]) -- what fs_conv_columns reads as xm[L - 1 - f2] / xp[L - 1 - f2].
template <int L2>
SMX_HD constexpr int big_pi(int i) { return (i / L2) * L2 + (L2 - 1 - i % L2); }
template <int L2>
SMX_HD void big_bins(int u, int t2, int i, int& fp, int& fm) {
  const int q1 = t2 * (16 / L2) + i / L2, q2 = i % L2;
  fp = u + 256 * (q1 + 16 * q2);
  fm = ((256 - u) & 255) + 256 * ((15 - q1) + 16 * q2);
}
// columns *= Hfull / N (DIR 0) or conj(Hfull) / N (DIR 1)
template <int L2, int DIR>
SMX_HD void fsb_conv_scale(BigState& st, const Geom& g, const ConvArgs& ca, bool valid, int u, int t2) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int fp, fm;
    big_bins<L2>(u, t2, i, fp, fm);
    const cf hp = conv_hfull(ca, fp, g.N), hm = conv_hfull(ca, fm, g.N);
    const cf hpe = cscale(DIR ? cconj(hp) : hp, g.inv_n), hme = cscale(DIR ? cconj(hm) : hm, g.inv_n);
    st.zp[i] = valid ? cmul(st.zp[i], hpe) : mk(0.f, 0.f);
    st.zm[i] = valid ? cmul(st.zm[i], hme) : mk(0.f, 0.f);
  }
}
// backward sums of the thread's bins (sg: columns of g, sx: columns of x), as fs_conv_columns: (R1, R2) are
// returned, each P term is handed to `emit(bin, value)` as soon as it exists (64 registers less than two arrays)
template <int L2, typename Emit>
SMX_HD void fsb_conv_sums(const BigState& sg, const BigState& sx, const Geom& g, const ConvArgs& ca, int b, int d,
                          bool valid, int u, int t2, cf& rr, Emit emit) {
  const bool one_col = (u == 0 || u == 128);
  float sig = 1.f, del = 0.f;
  if (ca.sc) {
    const int dl = valid ? d : g.D - 2;
    const float sa = ca.sc[(size_t)b * g.D + dl], sb = ca.sc[(size_t)b * g.D + dl + 1];
    sig = 0.5f * (sa + sb); del = 0.5f * (sa - sb);
  }
  float r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int fp, fm;
    big_bins<L2>(u, t2, i, fp, fm);
    const int pi = big_pi<L2>(i);
    const cf hp = conv_hfull(ca, fp, g.N), hm = conv_hfull(ca, fm, g.N);
    const cf xneg_p = sx.zm[pi], gp = sg.zp[i];                   // Zx[-fp]
    cf pp = cmul(gp, cadd(cscale(cconj(sx.zp[i]), sig), cscale(xneg_p, del)));
    const cf wp = cmul(hp, sx.zp[i]), wneg_p = cmulc(xneg_p, hp);
    r1 += gp.x * wp.x + gp.y * wp.y;
    r2 += gp.x * wneg_p.x - gp.y * wneg_p.y;
    cf pm = mk(0.f, 0.f);
    if (!one_col) {
      const cf xneg_m = sx.zp[pi], gm = sg.zm[i];                 // Zx[-fm]
      pm = cmul(gm, cadd(cscale(cconj(sx.zm[i]), sig), cscale(xneg_m, del)));
      const cf wm = cmul(hm, sx.zm[i]), wneg_m = cmulc(xneg_m, hm);
      r1 += gm.x * wm.x + gm.y * wm.y;
      r2 += gm.x * wneg_m.x - gm.y * wneg_m.y;
    }
    if (!valid) { pp = mk(0.f, 0.f); pm = mk(0.f, 0.f); }
    emit(fp, fm, pp, pm, one_col);
  }
  rr = valid ? mk(r1, r2) : mk(0.f, 0.f);
}

}  // namespace smx

// =====================================================================================================
// Rank-one filter in ONE launch per direction (round 3): n_fft = N = 512 LP, LP in {1, 2, 4} (rows <= N / 2, or folded) --
// fft_lm's default causal convolution (seq_len 1024, 128 taps: n_fft 2048; reference
// fft_lm/train_fixed_full.py:507-555).  The three-launch four-step form above moves the packed tile spectra
// through HBM twice per direction (5.6 x the algorithmic bytes at (64, 1024, 512)); here x is read once, y is
// written once, and the only other traffic is the packed spectrum of x kept for backward.
//
// The zero padding is what makes it fit: with rows <= N / 2 the N-point spectrum splits by PARITY of the bin into
// two N' = N / 2 point transforms of the same rows,
//     Z[2 f']     = DFT_N'( z[n] )[f'],            Z[2 f' + 1] = DFT_N'( z[n] w_N^n )[f'],
//     y[n]        = ( IDFT_N'(Y[2 .])[n] + w_N^{-n} IDFT_N'(Y[2 . + 1])[n] ) / N          for n < N / 2,
// and a workgroup of 512 threads runs the two halves side by side: half p = tid >> 8 (waves 0-3 / 4-7, so every
// branch on p is wave-uniform) is a 256-thread team exactly like the other kernels' workgroup -- 16 row groups t x
// 16 channel pairs j, tiles of 256 rows n = LP (t + 16 u) + r, two radix-16 passes around one LDS exchange -- with
// the per-residue spectra kept apart (16 LP complex per thread) and an LP-point transform across the residues, as
// in the eight-band kernel.  Thread (p, q, j) ends up with the bins f = 2 (q + 16 sl) + p, sl < 16 LP.  The odd
// half's modulation w_N^n = w_N^{LP t + r} w_32^u costs 15 constant multiplies before the first pass (the w_32^u)
// and rides in the inter-pass twiddle otherwise.  The mirror image of a bin has the same parity, so backward finds
// Zx[-f] in the saved spectrum of the same half (slot / thread of c1_mirror).  At the store the halves swap the
// eight rows the other one writes through LDS, so each stores half of every tile.
// LDS: 2 halves x 2 exchange buffers (128 KiB) + the response Hfull (N complex): one workgroup per CU, eight
// waves -- the occupancy of the two-workgroup kernels.  NJ = 8: the same on 16 channels with 256 threads, 80 KiB,
// two workgroups per CU.  Rows beyond N / 2: folded onto the lower half at the load (c1_fold_in), two output rows
// per value at the store (c1_comb_store<.., FOLD>).
// =====================================================================================================
namespace smx {

constexpr int C1_TPB = 512;
// NJ = channel pairs per workgroup (16: the 512 threads above; 8: 256 threads owning 16 channels -- two workgroups per CU)
template <int NJ> SMX_HD constexpr int c1_tpb() { return 32 * NJ; }      // threads of a workgroup: 2 teams x 16 t x NJ j

// w_64^k = exp(-2 pi i k / 64) as a function of a compile-time k (flat selects, no table in memory, no recursion:
// once the loops are unrolled every use is a literal operand)
SMX_HD constexpr float c1_q64(int i) {           // cos(2 pi i / 64), 0 <= i <= 16
  return i == 0 ? 1.f : i == 1 ? 0.99518472667219688624f : i == 2 ? 0.98078528040323044913f
       : i == 3 ? 0.95694033573220886494f : i == 4 ? 0.92387953251128675613f : i == 5 ? 0.88192126434835502971f
       : i == 6 ? 0.83146961230254523708f : i == 7 ? 0.77301045336273696081f : i == 8 ? 0.70710678118654752440f
       : i == 9 ? 0.63439328416364549822f : i == 10 ? 0.55557023301960222474f : i == 11 ? 0.47139673682599764856f
       : i == 12 ? 0.38268343236508977173f : i == 13 ? 0.29028467725446236764f : i == 14 ? 0.19509032201612826785f
       : i == 15 ? 0.09801714032956060199f : 0.f;
}
SMX_HD constexpr float c1_cos64(int k) {
  const int m = k & 63;
  return m <= 16 ? c1_q64(m) : m <= 32 ? -c1_q64(32 - m) : m <= 48 ? -c1_q64(m - 32) : c1_q64(64 - m);
}
SMX_HD constexpr float c1_sin64(int k) { return c1_cos64(k - 16); }
// exp(SGN 2 pi i k / 64)
template <int SGN> SMX_HD constexpr cf c1_w64(int k) { return cf{c1_cos64(k), (float)SGN * c1_sin64(k)}; }

// the odd half's modulation of a tile: v[u] *= w_32^u (SGN -1) or its conjugate (SGN +1)
template <int SGN>
SMX_HD void c1_mod32(cf (&v)[16]) {
#pragma unroll
  for (int u = 1; u < 16; ++u) {
    if (u == 8) v[u] = (SGN < 0) ? mul_mi(v[u]) : mul_pi(v[u]);
    else v[u] = cmul(v[u], c1_w64<SGN>(2 * u));
  }
}

// residue twiddle w_N'^{16 s r} = exp(-2 pi i s r / (16 LP)): a literal
template <int LP> SMX_HD constexpr cf c1_bt(int s, int r) { return c1_w64<-1>(s * r * (4 / LP)); }

// forward tile, before the barrier: [odd half: x w_32^u] -> radix-16 over u -> x w_N'^{q (LP t + r)} [x w_N^{LP t + r}]
// -> E[t][q][j]      (E: this half's exchange buffer)
// D[q] = d c^q for q < 16 (c = d^2 for the odd half's twiddles d^{2q+1}): three squarings + 15 products, depth <= 5
// -- four products more than powers16, against sixteen for powers16 followed by a multiplication by d
SMX_HD void c1_powers16_shifted(cf d, cf c, cf (&D)[16]) {
  const cf c2 = cmul(c, c), c4 = cmul(c2, c2), c8 = cmul(c4, c4);
  D[0] = d;
  D[1] = cmul(d, c);
  D[2] = cmul(D[0], c2); D[3] = cmul(D[1], c2);
#pragma unroll
  for (int q = 0; q < 4; ++q) D[4 + q] = cmul(D[q], c4);
#pragma unroll
  for (int q = 0; q < 8; ++q) D[8 + q] = cmul(D[q], c8);
}
// we = w_N^e, w2e = w_N^{2e}, e = LP t + r: handed in by the caller, who loads the LP pairs of a thread at launch start --
// read here, per tile, the two loads sat BEHIND the next tile's 16 prefetch loads in the in-order vector-memory
// queue, and waiting for them meant waiting for the whole prefetch (round 4)
template <int LP, int NJ = 16>
SMX_HD void c1_fwd_phase1(cf (&v)[16], cf we, cf w2e, cf* __restrict__ E, int p, int t, int j) {
  cf cp[16];
  if (p) {
    c1_powers16_shifted(we, w2e, cp);
    c1_mod32<-1>(v);
    fft16<-1>(v);
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], cp[q]);
  } else {
    powers16(w2e, cp);
    fft16<-1>(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], cp[q]);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) E[(t * 16 + q) * NJ + j] = v[q];
}
template <int LP, int NJ = 16>
SMX_HD void c1_fwd_phase1(cf (&v)[16], const cf* __restrict__ tw, cf* __restrict__ E, int p, int t, int j, int r) {
  const int e = LP * t + r;
  c1_fwd_phase1<LP, NJ>(v, tw[e], tw[2 * e], E, p, t, j);
}
// after the barrier: thread q = t gathers, radix-16 over t', keeps residue R's spectrum in acc[16 R + s]
template <int LP, int R, int NJ = 16>
SMX_HD void c1_fwd_phase2(cf (&acc)[16 * LP], const cf* __restrict__ E, int t, int j) {
  cf e[16];
#pragma unroll
  for (int t2 = 0; t2 < 16; ++t2) e[t2] = E[(t2 * 16 + t) * NJ + j];
  fft16<-1>(e);
#pragma unroll
  for (int s = 0; s < 16; ++s) acc[16 * R + s] = (R == 0 || s == 0) ? e[s] : cmul(c1_bt<LP>(s, R), e[s]);
}
// residues <-> bins f' = q + 16 s + 256 f2 (slot 16 f2 + s), in place
template <int LP, int SGN>
SMX_HD void c1_residues(cf (&acc)[16 * LP]) {
  if constexpr (LP == 4) {
#pragma unroll
    for (int s = 0; s < 16; ++s) radix4<SGN>(acc[s], acc[16 + s], acc[32 + s], acc[48 + s]);
  } else if constexpr (LP == 2) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const cf a = acc[s], b = acc[16 + s];
      acc[s] = cadd(a, b);
      acc[16 + s] = csub(a, b);
    }
  }
}
// inverse tile R, before the barrier
template <int LP, int R, int NJ = 16>
SMX_HD void c1_inv_phase1(const cf (&acc)[16 * LP], cf (&v)[16], cf* __restrict__ E, int q, int j) {
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = (R == 0 || s == 0) ? acc[16 * R + s] : cmulc(acc[16 * R + s], c1_bt<LP>(s, R));
  fft16<+1>(v);
#pragma unroll
  for (int tp = 0; tp < 16; ++tp) E[(q * 16 + tp) * NJ + j] = v[tp];
}
// after the barrier: v[u] = this half's part of row n = LP (t + 16 u) + r (not yet divided by N: Hfull carries it)
template <int LP, int NJ = 16>
SMX_HD void c1_inv_phase2(cf (&v)[16], cf we, cf w2e, const cf* __restrict__ E, int p, int t, int j) {
  cf cp[16];
#pragma unroll
  for (int q2 = 0; q2 < 16; ++q2) v[q2] = E[(q2 * 16 + t) * NJ + j];
  if (p) {
    c1_powers16_shifted(we, w2e, cp);
#pragma unroll
    for (int q2 = 0; q2 < 16; ++q2) v[q2] = cmulc(v[q2], cp[q2]);
    fft16<+1>(v);
    c1_mod32<+1>(v);
  } else {
    powers16(w2e, cp);
#pragma unroll
    for (int q2 = 1; q2 < 16; ++q2) v[q2] = cmulc(v[q2], cp[q2]);
    fft16<+1>(v);
  }
}
template <int LP, int NJ = 16>
SMX_HD void c1_inv_phase2(cf (&v)[16], const cf* __restrict__ tw, const cf* __restrict__ E, int p, int t, int j,
                          int r) {
  const int e = LP * t + r;
  c1_inv_phase2<LP, NJ>(v, tw[e], tw[2 * e], E, p, t, j);
}

// the response in LDS: Hs[f] = Hfull[f] / N, f < N (both directions: backward's (R1, R2) come out divided by N)
template <int NJ = 16>
SMX_HD void c1_stage_h(const ConvArgs& ca, int N, float inv_n, cf* __restrict__ Hs, int tid) {
  for (int f = tid; f < N; f += c1_tpb<NJ>()) Hs[f] = cscale(conv_hfull(ca, f, N), inv_n);
}
SMX_HD int c1_bin(int p, int q, int sl) { return 2 * q + p + 32 * sl; }
// Pins the accumulators in registers at this point (an empty asm that "modifies" each of them): the compiler can
// neither move their producers below nor their consumers above, which keeps the phases of the launch apart in the
// schedule (interleaved, the register allocator spills hundreds of values).
template <int CNT>
SMX_HD void c1_pin(cf (&a)[CNT]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < CNT; ++i) asm volatile("" : "+v"(a[i].x), "+v"(a[i].y));
#endif
}
// keeps the compiler from hoisting every chunk's loads to the top of an unrolled slot loop (hundreds of spills)
SMX_HD void c1_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}
// A lane whose channel pair lies past D (ragged last d-tile) transforms a copy of a valid pair and never stores a
// row.  Forward needs no predicate at all; backward hands such a lane sig = del = inv_n = 0 (per-thread scalars: a
// per-bin predicate becomes a branch per bin), so it adds nothing to the sums over the channel pairs.
// between the loops, forward: keep the packed spectrum of x (xsave: this workgroup's 16 LP x 512 block, or null),
// then Y = Z Hfull / N
template <int LP, int NJ = 16>
SMX_HD void c1_mid_fwd(cf (&acc)[16 * LP], const cf* __restrict__ Hs, cf* __restrict__ xsave, int p, int q,
                       int tid) {
  c1_pin(acc);
#pragma unroll
  for (int c0 = 0; c0 < 16 * LP; c0 += 16) {
    cf h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = Hs[c1_bin(p, q, c0 + i)];
    if (xsave) {                       // 16 rows of the saved spectrum leave per batch of bins (streamed: the next
#pragma unroll                         //  reader is the backward launch)
      for (int i = 0; i < 16; ++i) {
        cf* dst = xsave + ((unsigned)((c0 + i) * c1_tpb<NJ>()) + (unsigned)tid);
        st_stream(reinterpret_cast<float*>(dst), acc[c0 + i].x, acc[c0 + i].y, false);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c0 + i] = cmul(acc[c0 + i], h[i]);
    c1_pin(acc);
  }
}
// between the loops, backward (acc = Zg): the P / (R1, R2) terms of fs_conv_columns against the saved spectrum of x,
// then Y = Zg conj(Hfull) / N.  With Y in hand, Re(Zg conj W[f]) = Re(Y conj Zx[f]) N and Re(Zg W[-f]) = Re(Y Zx[-f]) N
// (W = Hfull Zx): the sums are formed from Y and come out as (R1, R2) / N.
// Order: slot m next to slot 16 LP - 1 - m.  The mirror image of (q, sl) lives in thread q'' at slot 16 LP - 1 - sl
// (bins 32 sl of thread (0, 0): its own slot 16 LP - sl), so the rows of the saved spectrum a step reads as mirror
// images are the rows its partner threads read directly in the same step: the second read is served by the caches,
// not by HBM (in ascending slot order the two reads of a row are the whole 256 KiB block apart, times 32 CUs per L2).
// emit16(g, px, py): the P terms of the 16 bins c1_bin(p, q, c1_group_slot<LP>(g, i)), i < 16, to be summed over j.
template <int LP> SMX_HD constexpr int c1_group_slot(int g, int i) { return i < 8 ? 8 * g + i : 16 * LP - 16 - 8 * g + i; }
template <int LP, int NJ = 16, typename Emit16>
SMX_HD void c1_mid_bwd(cf (&acc)[16 * LP], const cf* __restrict__ Hs, const cf* __restrict__ xs, float sig,
                       float del, int p, int q, int j, int tid, cf& rr, Emit16 emit16) {
  float r1 = 0.f, r2 = 0.f;
  constexpr int CH = 4;                                                  // pairs of slots per batch of loads
  const unsigned mt = (unsigned)(p * (16 * NJ) + (p ? 15 - q : (16 - q) & 15) * NJ + j);   // thread of the mirror images
  const bool self0 = p == 0 && q == 0;
  c1_pin(acc);
#pragma unroll
  for (int g = 0; g < LP; ++g) {
    float px[16], py[16];
#pragma unroll
    for (int c1 = 0; c1 < 8; c1 += CH) {
      cf x1[2 * CH], x2[2 * CH];
#pragma unroll
      for (int i = 0; i < 2 * CH; ++i) {
        const int gi = i < CH ? c1 + i : 15 - c1 - (i - CH);             // c1 .. c1+CH-1, then their partners
        const int sl = c1_group_slot<LP>(g, gi);
        const unsigned sl2 = self0 ? (unsigned)((16 * LP - sl) & (16 * LP - 1)) : (unsigned)(16 * LP - 1 - sl);
        x1[i] = xs[(unsigned)(sl * c1_tpb<NJ>()) + (unsigned)tid];
        x2[i] = xs[sl2 * (unsigned)c1_tpb<NJ>() + mt];
      }
#pragma unroll
      for (int i = 0; i < 2 * CH; ++i) {
        const int gi = i < CH ? c1 + i : 15 - c1 - (i - CH);
        const int sl = c1_group_slot<LP>(g, gi);
        const cf h = Hs[c1_bin(p, q, sl)];
        const cf gz = acc[sl];
        const cf pp = cmul(gz, cadd(cscale(cconj(x1[i]), sig), cscale(x2[i], del)));
        const cf y = cmulc(gz, h);                           // Zg conj(Hfull) / N
        r1 += y.x * x1[i].x + y.y * x1[i].y;                 // Re(Y conj Zx[f])
        r2 += y.x * x2[i].x - y.y * x2[i].y;                 // Re(Y Zx[-f])
        px[gi] = pp.x;
        py[gi] = pp.y;
        acc[sl] = y;
      }
      c1_fence();
    }
    emit16(g, px, py);
    c1_pin(acc);
  }
  rr = mk(r1, r2);
}
// the two halves of a row: each half hands over the eight values the other one stores (C: 2 x 8 x 256 complex)
template <int NJ = 16>
SMX_HD void c1_comb_write(const cf (&v)[16], cf* __restrict__ C, int p, int lt) {
  // (value selects, not two branches with different register indices: the compiler merges such branches into
  //  one dynamically indexed access and the tile lands in scratch memory)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cf w = mk(p ? v[k].x : v[8 + k].x, p ? v[k].y : v[8 + k].y);
    C[(8 * p + k) * (16 * NJ) + lt] = w;
  }
}
// rows u = 8 p + k of tile r: sum of the halves, scaled by (sa, sb), stored (g: the N' tile geometry: L = LP)
// rows > N / 2 (FOLD): the transforms ran on x[n] +/- x[n + N/2] (c1_fold_in), and the same two halves give two output
// rows each: y[n] = e + o', y[n + N/2] = e - o' (e: the even team's value, o': the odd team's, already demodulated)
SMX_HD void c1_fold_in(cf (&v)[16], const cf (&hi)[16], int p) {
  const float sg = p ? -1.f : 1.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = mk(__builtin_fmaf(sg, hi[u].x, v[u].x), __builtin_fmaf(sg, hi[u].y, v[u].y));
}
template <bool PAD, int NJ = 16, bool FOLD = false>
SMX_HD void c1_comb_store(const cf (&v)[16], const cf* __restrict__ C, float* __restrict__ yb, const Geom& g,
                          int p, int t, int lt, int r, bool valid, float sa, float sb) {
  if (!valid) return;
  const size_t stride = (size_t)16 * g.L * g.D;
  float* ptr = yb + ((size_t)t * g.L + r) * g.D;
  cf o[8], o2[FOLD ? 8 : 1];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cf own = mk(p ? v[8 + k].x : v[k].x, p ? v[8 + k].y : v[k].y);
    const cf oth = C[(8 * (1 - p) + k) * (16 * NJ) + lt];
    o[k] = cadd(own, oth);
    if constexpr (FOLD) {
      const cf df = csub(own, oth);                    // even team: e - o'; odd team: o' - e -> negate
      o2[k] = mk(p ? -df.x : df.x, p ? -df.y : df.y);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int u = 8 * p + k;
    const int n = (t + 16 * u) * g.L + r;              // row of the lower half; FOLD: always present, its partner n + N'
    if (!FOLD && PAD && n >= g.R) continue;            //   is present while n + N' < R (g: the N' tile geometry, R rows)
    float* dst = ptr + (size_t)u * stride;
    // (each team stores 8 of the 16 rows: the first st_plain / 2 of them write-back, as store_tile's first st_plain)
    st_stream(dst, o[k].x * sa, o[k].y * sb, k < 2 && 2 * k < g.st_plain);
    if constexpr (FOLD) {
      if (PAD && n + g.N >= g.R) continue;
      float* dst2 = dst + (size_t)g.N * g.D;
      st_stream(dst2, o2[k].x * sa, o2[k].y * sb, k < 2 && 2 * k < g.st_plain);
    }
  }
}

}  // namespace smx
