// smx_tables.h -- host-side twiddle tables (computed in fp64, rounded once to fp32).
#pragma once
#include <cmath>
#include <vector>
#include "smx_core.h"

namespace smx {

// tw[n] = exp(-2 pi i n / N), n in [0, N)
inline std::vector<cf> make_tw(int N) {
  std::vector<cf> t((size_t)(N > 0 ? N : 1));
  const double step = -2.0 * M_PI / (double)N;
  for (int n = 0; n < N; ++n) {
    // reduce to the first octant-free form: plain cos/sin in double is accurate to <1e-16 here
    double a = step * (double)n;
    t[n] = mk((float)std::cos(a), (float)std::sin(a));
  }
  return t;
}

// tq[e*16 + q] = exp(-2 pi i q e / N), e in [0, N/16), q in [0, 16): row e = t L + r holds the sixteen inter-pass
// twiddles (w_N^{L t + r})^q of row-group t and residue r (N = 256 L), each rounded once from fp64
inline std::vector<cf> make_tq(int N) {
  const int rows = N / 16;
  std::vector<cf> t((size_t)(rows > 0 ? rows : 1) * 16);
  for (int e = 0; e < rows; ++e)
    for (int q = 0; q < 16; ++q) {
      const long long m = ((long long)q * e) % N;
      const double a = -2.0 * M_PI * (double)m / (double)N;
      t[(size_t)e * 16 + q] = mk((float)std::cos(a), (float)std::sin(a));
    }
  return t;
}

// bt[r*64 + (s'+32)] = exp(-2 pi i * 16 s'' r / N), r in [0,L), s' in [-32,32).
// Band group g (k > 512: the bins are covered 512 at a time, smx_api.hip) shifts the four bands
// outwards by 512 g bins: s'' = s' + 32 g for the positive bands (s' >= 0), s' - 32 g for the negative.
inline std::vector<cf> make_bt(int N, int L, int group = 0) {
  std::vector<cf> t((size_t)L * BT_STRIDE);
  for (int r = 0; r < L; ++r)
    for (int i = 0; i < BT_STRIDE; ++i) {
      const int sp = i - BT_HALF;
      long long e = (long long)16 * (sp >= 0 ? sp + 32 * group : sp - 32 * group) * r;   // may be negative
      long long m = ((e % N) + N) % N;
      double a = -2.0 * M_PI * (double)m / (double)N;
      t[(size_t)r * BT_STRIDE + i] = mk((float)std::cos(a), (float)std::sin(a));
    }
  return t;
}

// ---- sixteen-row decimation (N = 16 P, any P: smx_core.h, "N % 16 == 0") -----------------------------------
// v16[(s'' + 16) * 16 + t'] = w_P^{s'' t'}: row s'' in [-16, 16) is the block of 16 bins f = q + 16 s'' an accumulator
// slot holds (one band: s'' in [-8, 8); two bands: all 32 rows), t' < 16
inline std::vector<cf> make_v16(int N) {
  const int P = N / 16;
  std::vector<cf> t(32 * 16);
  for (int si = 0; si < 32; ++si)
    for (int tp = 0; tp < 16; ++tp) {
      const long long e = (long long)(si - 16) * tp;
      const long long m = ((e % P) + P) % P;
      const double a = -2.0 * M_PI * (double)m / (double)P;
      t[(size_t)si * 16 + tp] = mk((float)std::cos(a), (float)std::sin(a));
    }
  return t;
}
// b16[tau * 32 + (s'' + 16)] = w_P^{16 s'' tau}: the part of the residue twiddle common to a tile of 16 residues
inline std::vector<cf> make_b16(int N) {
  const int P = N / 16, T = (P + 15) / 16;
  std::vector<cf> t((size_t)T * 32);
  for (int tau = 0; tau < T; ++tau)
    for (int si = 0; si < 32; ++si) {
      const long long e = (long long)16 * (si - 16) * tau;
      const long long m = ((e % P) + P) % P;
      const double a = -2.0 * M_PI * (double)m / (double)P;
      t[(size_t)tau * 32 + si] = mk((float)std::cos(a), (float)std::sin(a));
    }
  return t;
}

}  // namespace smx
