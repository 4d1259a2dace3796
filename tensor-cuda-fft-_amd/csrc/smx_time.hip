// smx_time.hip -- native pieces of fft_lm's spectrum-domain twin blocks that are NOT transforms: the time path of
// BicameralBlock on the (B, T, C) layout the rest of the block lives in, and SpectralLayerNorm (further down).
//
// Replaces: reference fft_lm/bicameral.py:214-223 -- transpose to (B, C, T), shift right by one and drop the last
// position (F.pad(x[:, :, :-1], (1, 0))), nn.Conv1d(C, C, kernel_size = 3, padding = 1, groups = C), transpose back --
// and :226-227, the time gate sigmoid(gate_time(pooled))[b, c] on the result; and their autograd backward.
// Written out, with x the (B, T, C) input and w = conv1d.weight[:, 0, :] (C, 3):
//     y0[b, t, c] = bias[c] + w[c,0] x[t-2] + w[c,1] x[t-1] + w[c,2] x[t] [t <= T-2]        (x[<0] = 0)
//     y = scale[b, c] y0                                                                     (scale optional)
// (the tap on x[t] is missing from the LAST output row: the shifted sequence dropped x[T-1] and the convolution's own
// right padding supplies a zero there).  Through MIOpen the depthwise Conv1d and its two transposes were 40 % of all
// kernel time of the twin blocks (naive weight-gradient kernels of 9.6 ms each; gpurun_out/r04i): the op itself is a
// streaming pass -- read x, write y; backward read g and x, write grad_x, reduce four sums per channel.
//
// Thread = one channel (coalesced along C, the fastest axis), walking TR consecutive rows of one batch row with a
// sliding window in registers (four channels per thread, 16-byte accesses, when C % 4 == 0); grid = channel groups x
// B * ceil(T / TR) blocks.  Backward: the per-channel sums of a block go to a partial buffer [block][5][C]; two small launches
// add them in block order -- fixed order, bitwise reproducible.
#include "smx_kernels.h"

namespace smx {

namespace {

constexpr int DW_TR = 32;            // rows per block
constexpr int DW_TPB = 256;

// V channels per thread: 4 (one 16-byte access per row; C % 4 == 0 and 16-byte aligned bases) or 1
template <int V> struct Vec { float v[V]; };
template <int V> __device__ __forceinline__ Vec<V> ldv(const float* p) {
  Vec<V> r;
  if constexpr (V == 4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int V> __device__ __forceinline__ void stv(float* p, const Vec<V>& a) {
  if constexpr (V == 4) {
    f32x4 q; q.x = a.v[0]; q.y = a.v[1]; q.z = a.v[2]; q.w = a.v[3];
    *reinterpret_cast<f32x4*>(p) = q;
  } else {
    *p = a.v[0];
  }
}
template <int V> __device__ __forceinline__ Vec<V> zerov() {
  Vec<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = 0.f;
  return r;
}

template <int V>
__global__ __launch_bounds__(DW_TPB) void k_dwconv3_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ scale, float* __restrict__ y,
                                                       int B, int T, int C) {
  const int ncg = (C + DW_TPB * V - 1) / (DW_TPB * V);                  // channel groups; blockIdx.x = group + ncg * chunk
  const int blk = blockIdx.x / ncg;
  const int c = ((blockIdx.x % ncg) * DW_TPB + threadIdx.x) * V;
  if (c >= C) return;
  const int nch = (T + DW_TR - 1) / DW_TR;
  const int b = blk / nch, t0 = (blk % nch) * DW_TR, t1 = min(T, t0 + DW_TR);
  float w0[V], w1[V], w2[V], bi[V], s[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    w0[i] = w[(c + i) * 3]; w1[i] = w[(c + i) * 3 + 1]; w2[i] = w[(c + i) * 3 + 2];
    bi[i] = bias ? bias[c + i] : 0.f;
    s[i] = scale ? scale[(size_t)b * C + c + i] : 1.f;
  }
  const float* xb = x + (size_t)b * T * C + c;
  float* yb = y + (size_t)b * T * C + c;
  Vec<V> xm2 = t0 >= 2 ? ldv<V>(xb + (size_t)(t0 - 2) * C) : zerov<V>();
  Vec<V> xm1 = t0 >= 1 ? ldv<V>(xb + (size_t)(t0 - 1) * C) : zerov<V>();
#pragma unroll 4
  for (int t = t0; t < t1; ++t) {
    const Vec<V> xt = ldv<V>(xb + (size_t)t * C);
    const bool tap2 = t <= T - 2;
    Vec<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float v = __builtin_fmaf(w0[i], xm2.v[i], bi[i]);
      v = __builtin_fmaf(w1[i], xm1.v[i], v);
      if (tap2) v = __builtin_fmaf(w2[i], xt.v[i], v);
      o.v[i] = s[i] * v;
    }
    stv<V>(yb + (size_t)t * C, o);
    xm2 = xm1; xm1 = xt;
  }
}

// grad_x[t] = w0 gs[t+2] + w1 gs[t+1] + w2 gs[t] [t <= T-2],  gs = scale g  (rows beyond T-1 contribute nothing)
// partial sums of the block: part[blk][0..2][c] = sum_t gs[t] x[t-2+k] (k-th tap, with the rule above for k = 2),
// part[blk][3][c] = sum_t gs[t], part[blk][4][c] = sum_t g[t] y0[t]   (-> grad_scale of batch row b)
template <int V>
__global__ __launch_bounds__(DW_TPB) void k_dwconv3_bwd(const float* __restrict__ g, const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ scale,
                                                       float* __restrict__ gx, float* __restrict__ part, int B,
                                                       int T, int C) {
  const int ncg = (C + DW_TPB * V - 1) / (DW_TPB * V);                  // channel groups; blockIdx.x = group + ncg * chunk
  const int blk = blockIdx.x / ncg;
  const int c = ((blockIdx.x % ncg) * DW_TPB + threadIdx.x) * V;
  if (c >= C) return;
  const int nch = (T + DW_TR - 1) / DW_TR;
  const int b = blk / nch, t0 = (blk % nch) * DW_TR, t1 = min(T, t0 + DW_TR);
  float w0[V], w1[V], w2[V], bi[V], s[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    w0[i] = w[(c + i) * 3]; w1[i] = w[(c + i) * 3 + 1]; w2[i] = w[(c + i) * 3 + 2];
    bi[i] = bias ? bias[c + i] : 0.f;
    s[i] = scale ? scale[(size_t)b * C + c + i] : 1.f;
  }
  const float* gb = g + (size_t)b * T * C + c;
  const float* xb = x + (size_t)b * T * C + c;
  float* gxb = gx ? gx + (size_t)b * T * C + c : nullptr;
  Vec<V> xm2 = t0 >= 2 ? ldv<V>(xb + (size_t)(t0 - 2) * C) : zerov<V>();
  Vec<V> xm1 = t0 >= 1 ? ldv<V>(xb + (size_t)(t0 - 1) * C) : zerov<V>();
  Vec<V> g0 = ldv<V>(gb + (size_t)t0 * C);                       // g[t], g[t+1], g[t+2] (0 past the end)
  Vec<V> g1 = t0 + 1 < T ? ldv<V>(gb + (size_t)(t0 + 1) * C) : zerov<V>();
  Vec<V> a0 = zerov<V>(), a1 = zerov<V>(), a2 = zerov<V>(), ab = zerov<V>(), as = zerov<V>();
#pragma unroll 4
  for (int t = t0; t < t1; ++t) {
    const Vec<V> g2 = t + 2 < T ? ldv<V>(gb + (size_t)(t + 2) * C) : zerov<V>();
    const Vec<V> xt = ldv<V>(xb + (size_t)t * C);
    const bool tap2 = t <= T - 2;
    Vec<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float v = w0[i] * g2.v[i];
      v = __builtin_fmaf(w1[i], g1.v[i], v);
      if (tap2) v = __builtin_fmaf(w2[i], g0.v[i], v);
      o.v[i] = s[i] * v;
      const float gs = s[i] * g0.v[i];
      a0.v[i] = __builtin_fmaf(gs, xm2.v[i], a0.v[i]);
      a1.v[i] = __builtin_fmaf(gs, xm1.v[i], a1.v[i]);
      if (tap2) a2.v[i] = __builtin_fmaf(gs, xt.v[i], a2.v[i]);
      ab.v[i] += gs;
      float y0 = __builtin_fmaf(w0[i], xm2.v[i], bi[i]);
      y0 = __builtin_fmaf(w1[i], xm1.v[i], y0);
      if (tap2) y0 = __builtin_fmaf(w2[i], xt.v[i], y0);
      as.v[i] = __builtin_fmaf(g0.v[i], y0, as.v[i]);
    }
    if (gxb) stv<V>(gxb + (size_t)t * C, o);
    xm2 = xm1; xm1 = xt; g0 = g1; g1 = g2;
  }
  float* p = part + (size_t)blk * 5 * C + c;
  stv<V>(p, a0); stv<V>(p + (size_t)C, a1); stv<V>(p + (size_t)2 * C, a2); stv<V>(p + (size_t)3 * C, ab);
  stv<V>(p + (size_t)4 * C, as);
}

// Two small launches add the partial sums in a fixed order: (a) per batch row b the nch row chunks -> grad_scale[b, c]
// and part2[b][0..3][c]; (b) the batch rows -> grad_w[c, k], grad_bias[c].
__global__ __launch_bounds__(DW_TPB) void k_dwconv3_sum_a(const float* __restrict__ part, float* __restrict__ part2,
                                                         float* __restrict__ gscale, int nch, int C) {
  const int c = blockIdx.x * DW_TPB + threadIdx.x, b = blockIdx.y, what = blockIdx.z;
  if (c >= C || (what == 4 && !gscale)) return;
  const float* p = part + ((size_t)b * nch * 5 + what) * C + c;
  float acc = 0.f;
  int ch = 0;
  for (; ch + 8 <= nch; ch += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(ch + u) * 5 * C];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; ch < nch; ++ch) acc += p[(size_t)ch * 5 * C];
  if (what == 4) gscale[(size_t)b * C + c] = acc;
  else part2[((size_t)b * 4 + what) * C + c] = acc;
}
__global__ __launch_bounds__(DW_TPB) void k_dwconv3_sum_b(const float* __restrict__ part2, float* __restrict__ gw,
                                                         float* __restrict__ gbias, int B, int C) {
  const int c = blockIdx.x * DW_TPB + threadIdx.x, what = blockIdx.y;
  if (c >= C || (what == 3 ? !gbias : !gw)) return;
  const float* p = part2 + (size_t)what * C + c;
  float acc = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * 4 * C];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; b < B; ++b) acc += p[(size_t)b * 4 * C];
  if (what == 3) gbias[c] = acc; else gw[c * 3 + what] = acc;
}

// ---- SpectralLayerNorm (reference fft_lm/frequency_native.py:203-239) -------------------------------------------------
// Per (batch row, bin): magnitudes normalised across the C channels, phases kept:
//     m = |z|,  mu = mean_c m,  var = mean_c (m - mu)^2,  s = (m - mu) rsqrt(var + eps) gamma[f] + beta[f],  out = s u
// with u = z / m (m = 0: u = cos / sin of atan2(+-0, +-0), i.e. +-1 by the sign of the real zero -- what
// exp(i angle(z)) gives there).  The reference spells this as abs, mean, var, four elementwise steps, angle, cos, sin,
// complex() and a product: about twelve passes over the (B, F, C) spectrum forward and twice that backward, 3 ms of
// FrequencyNativeBlock's 16.5 at (64, 1024, 512).  Here: one wavefront per row, the row in registers (CH complex per
// lane), two wave-wide sums.  Backward (real calculus in (Re, Im); G = dL/dRe + i dL/dIm as torch hands it over):
//     ds = Re(G conj u),  dtheta = s Im(G conj u),  a = ds gamma,  dm = r (a - mean a - shat mean(a shat)),
//     grad_z = u (dm + i dtheta / m)      (0 where m = 0, as torch's abs / angle backward),
//     grad_gamma[f, c] = sum_b ds shat,   grad_beta[f, c] = sum_b ds
// one workgroup per bin walks the batch rows (wave w takes b = w, w + 4, ...), the four waves' sums meet in LDS in a
// fixed order -- bitwise reproducible, no second launch.
constexpr int SLN_WAVES = 4;

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float sln_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float sln_wave_sum(float v) {       // as wave_sum in smx_block.hip
  v += sln_dpp<0x111, 0xf, true>(v);
  v += sln_dpp<0x112, 0xf, true>(v);
  v += sln_dpp<0x114, 0xf, true>(v);
  v += sln_dpp<0x118, 0xf, true>(v);
  v += sln_dpp<0x142, 0xa, false>(v);
  v += sln_dpp<0x143, 0xc, false>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int CH>
struct SlnRow {
  cf u[CH];            // unit phases
  float m[CH], sh[CH]; // magnitudes, normalised magnitudes
  float r;             // rsqrt(var + eps)
};
template <int CH>
__device__ __forceinline__ void sln_stats(const cf (&z)[CH], int lane, int C, float eps, SlnRow<CH>& o) {
  float sm = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const bool in = lane + 64 * k < C;
    const float m = sqrtf(__builtin_fmaf(z[k].x, z[k].x, z[k].y * z[k].y));
    o.m[k] = in ? m : 0.f;
    if (m > 0.f) o.u[k] = mk(z[k].x / m, z[k].y / m);
    else o.u[k] = mk(__builtin_signbit(z[k].x) ? -1.f : 1.f, 0.f);
    sm += o.m[k];
  }
  const float inv_c = 1.f / (float)C;
  const float mu = sln_wave_sum(sm) * inv_c;
  float v2 = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (lane + 64 * k < C) { const float d = o.m[k] - mu; v2 = __builtin_fmaf(d, d, v2); }
  o.r = 1.f / sqrtf(sln_wave_sum(v2) * inv_c + eps);
#pragma unroll
  for (int k = 0; k < CH; ++k) o.sh[k] = (o.m[k] - mu) * o.r;
}

// PLANAR: out / g are (2, B, F, C) float32 -- plane 0 the real parts, plane 1 the imaginary parts -- the layout
// SpectralFFN's nn.Linear wants for "the same weights on the real and on the imaginary part" (reference :167-172):
// the (de)interleaving copies around the two Linear layers disappear.
template <int CH, bool PLANAR>
__global__ __launch_bounds__(64 * SLN_WAVES) void k_sln_fwd(const cf* __restrict__ z, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          cf* __restrict__ out, long long rows, int F, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * SLN_WAVES + wv;
  if (row >= rows) return;
  const int f = (int)(row % F);
  cf v[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) v[k] = lane + 64 * k < C ? z[(size_t)row * C + lane + 64 * k] : mk(0.f, 0.f);
  SlnRow<CH> st;
  sln_stats<CH>(v, lane, C, eps, st);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      const float s = __builtin_fmaf(st.sh[k], gamma[(size_t)f * C + c], beta[(size_t)f * C + c]);
      if constexpr (PLANAR) {
        float* o = reinterpret_cast<float*>(out);
        o[(size_t)row * C + c] = s * st.u[k].x;
        o[((size_t)rows + (size_t)row) * C + c] = s * st.u[k].y;
      } else {
        out[(size_t)row * C + c] = mk(s * st.u[k].x, s * st.u[k].y);
      }
    }
  }
}

template <int CH, bool PLANAR>
__global__ __launch_bounds__(64 * SLN_WAVES) void k_sln_bwd(const cf* __restrict__ g, const cf* __restrict__ z,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          cf* __restrict__ gz, float* __restrict__ ggamma,
                                                          float* __restrict__ gbeta, int B, int F, int C) {
  __shared__ float red[2][SLN_WAVES][CH * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, f = blockIdx.x;
  float gm[CH], bt[CH], dg[CH], db[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = lane + 64 * k;
    gm[k] = c < C ? gamma[(size_t)f * C + c] : 0.f;
    bt[k] = c < C ? beta[(size_t)f * C + c] : 0.f;
    dg[k] = 0.f; db[k] = 0.f;
  }
  const float inv_c = 1.f / (float)C;
  for (int b = wv; b < B; b += SLN_WAVES) {
    const size_t row = (size_t)b * F + f;
    cf v[CH], gg[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const bool in = lane + 64 * k < C;
      v[k] = in ? z[row * C + lane + 64 * k] : mk(0.f, 0.f);
      if constexpr (PLANAR) {
        const float* gp = reinterpret_cast<const float*>(g);
        gg[k] = in ? mk(gp[row * C + lane + 64 * k], gp[((size_t)B * F + row) * C + lane + 64 * k]) : mk(0.f, 0.f);
      } else {
        gg[k] = in ? g[row * C + lane + 64 * k] : mk(0.f, 0.f);
      }
    }
    SlnRow<CH> st;
    sln_stats<CH>(v, lane, C, eps, st);
    float ds[CH], dth[CH], sa = 0.f, sas = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const bool in = lane + 64 * k < C;
      ds[k] = in ? __builtin_fmaf(gg[k].x, st.u[k].x, gg[k].y * st.u[k].y) : 0.f;
      const float s = __builtin_fmaf(st.sh[k], gm[k], bt[k]);
      dth[k] = s * __builtin_fmaf(gg[k].y, st.u[k].x, -(gg[k].x * st.u[k].y));
      const float a = ds[k] * gm[k];
      sa += a;
      sas = __builtin_fmaf(a, st.sh[k], sas);
      dg[k] = __builtin_fmaf(ds[k], st.sh[k], dg[k]);
      db[k] += ds[k];
    }
    const float ma = sln_wave_sum(sa) * inv_c, mas = sln_wave_sum(sas) * inv_c;
    if (gz) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          const float dm = st.r * (ds[k] * gm[k] - ma - st.sh[k] * mas);
          const float q = st.m[k] > 0.f ? dth[k] / st.m[k] : 0.f;
          const float dmz = st.m[k] > 0.f ? dm : 0.f;
          gz[row * C + c] = mk(st.u[k].x * dmz - st.u[k].y * q, st.u[k].y * dmz + st.u[k].x * q);
        }
      }
    }
  }
  if (!ggamma && !gbeta) return;
#pragma unroll
  for (int k = 0; k < CH; ++k) { red[0][wv][k * 64 + lane] = dg[k]; red[1][wv][k * 64 + lane] = db[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < CH * 64; i += 64 * SLN_WAVES) {
    const int c = (i & 63) + 64 * (i >> 6);
    if (c >= C) continue;
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int w = 0; w < SLN_WAVES; ++w) { a += red[0][w][i]; b2 += red[1][w][i]; }
    if (ggamma) ggamma[(size_t)f * C + c] = a;
    if (gbeta) gbeta[(size_t)f * C + c] = b2;
  }
}

template <typename F1, typename F2, typename F4, typename F8, typename F16>
bool sln_dispatch(int C, F1 f1, F2 f2, F4 f4, F8 f8, F16 f16) {
  const int ch = (C + 63) / 64;
  if (ch <= 1) f1(); else if (ch <= 2) f2(); else if (ch <= 4) f4(); else if (ch <= 8) f8(); else if (ch <= 16) f16();
  else return false;
  return true;
}

}  // namespace

bool spectral_ln_supported(int C) { return C >= 1 && C <= 1024; }

hipError_t launch_spectral_ln_fwd(const cf* z, const float* gamma, const float* beta, float eps, cf* out, int planar,
                                  int B, int F, int C, hipStream_t s) {
  const long long rows = (long long)B * F;
  const dim3 grid((unsigned)((rows + SLN_WAVES - 1) / SLN_WAVES)), block(64 * SLN_WAVES);
#define SLN_F(CH) [&] { if (planar) hipLaunchKernelGGL((k_sln_fwd<CH, true>), grid, block, 0, s, z, gamma, beta, eps, out, rows, F, C); \
                        else hipLaunchKernelGGL((k_sln_fwd<CH, false>), grid, block, 0, s, z, gamma, beta, eps, out, rows, F, C); }
  if (!sln_dispatch(C, SLN_F(1), SLN_F(2), SLN_F(4), SLN_F(8), SLN_F(16))) return hipErrorInvalidValue;
#undef SLN_F
  return hipGetLastError();
}
hipError_t launch_spectral_ln_bwd(const cf* g, const cf* z, const float* gamma, const float* beta, float eps, cf* gz,
                                  float* ggamma, float* gbeta, int planar, int B, int F, int C, hipStream_t s) {
  const dim3 grid(F), block(64 * SLN_WAVES);
#define SLN_B(CH) [&] { if (planar) hipLaunchKernelGGL((k_sln_bwd<CH, true>), grid, block, 0, s, g, z, gamma, beta, eps, gz, ggamma, gbeta, B, F, C); \
                        else hipLaunchKernelGGL((k_sln_bwd<CH, false>), grid, block, 0, s, g, z, gamma, beta, eps, gz, ggamma, gbeta, B, F, C); }
  if (!sln_dispatch(C, SLN_B(1), SLN_B(2), SLN_B(4), SLN_B(8), SLN_B(16))) return hipErrorInvalidValue;
#undef SLN_B
  return hipGetLastError();
}

// ---- the planar side of SpectralFFN (reference fft_lm/frequency_native.py:167-189) --------------------------------------
// h (2, B, F, H): real plane, imaginary plane.  PhaseShift there (:62-77 = one complex factor per (bin, channel)):
//     out = h (fr + i fi):  out_re = h_re fr - h_im fi,  out_im = h_re fi + h_im fr
// backward: gh = g conj(f);  d fr[f, c] = sum_b (g_re h_re + g_im h_im),  d fi[f, c] = sum_b (g_im h_re - g_re h_im)
// (workgroup per bin walking the batch rows, as k_sln_bwd: fixed order).
__global__ __launch_bounds__(256) void k_pcmul_fwd(const float* __restrict__ h, const float* __restrict__ fr,
                                                  const float* __restrict__ fi, float* __restrict__ out,
                                                  long long rows, int F, int C) {
  const long long total = rows * C, plane = total;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / C;
    const int c = (int)(i - row * C), f = (int)(row % F);
    const float a = h[i], b = h[plane + i], r = fr[(size_t)f * C + c], q = fi[(size_t)f * C + c];
    out[i] = __builtin_fmaf(a, r, -(b * q));
    out[plane + i] = __builtin_fmaf(a, q, b * r);
  }
}
__global__ __launch_bounds__(256) void k_pcmul_bwd(const float* __restrict__ g, const float* __restrict__ h,
                                                  const float* __restrict__ fr, const float* __restrict__ fi,
                                                  float* __restrict__ gh, float* __restrict__ gfr,
                                                  float* __restrict__ gfi, int B, int F, int C) {
  __shared__ float red[2][4][64];
  const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t plane = (size_t)B * F * C;
  for (int c0 = 0; c0 < C; c0 += 64) {                 // 64 channels at a time: wave w takes b = w, w + 4, ...
    const int c = c0 + lane;
    const float r = c < C ? fr[(size_t)f * C + c] : 0.f, q = c < C ? fi[(size_t)f * C + c] : 0.f;
    float ar = 0.f, ai = 0.f;
    for (int b = wv; b < B; b += 4) {
      if (c < C) {
        const size_t i = ((size_t)b * F + f) * C + c;
        const float gr = g[i], gi = g[plane + i], hr = h[i], hi = h[plane + i];
        if (gh) {
          gh[i] = __builtin_fmaf(gr, r, gi * q);
          gh[plane + i] = __builtin_fmaf(gi, r, -(gr * q));
        }
        ar = __builtin_fmaf(gr, hr, __builtin_fmaf(gi, hi, ar));
        ai = __builtin_fmaf(gi, hr, __builtin_fmaf(-gr, hi, ai));
      }
    }
    if (gfr || gfi) {
      red[0][wv][lane] = ar; red[1][wv][lane] = ai;
      __syncthreads();
      if (wv == 0 && c < C) {
        float a = 0.f, b2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[0][w][lane]; b2 += red[1][w][lane]; }
        if (gfr) gfr[(size_t)f * C + c] = a;
        if (gfi) gfi[(size_t)f * C + c] = b2;
      }
      __syncthreads();
    }
  }
}
// Four channels per lane (C % 4 == 0, 16-byte aligned planes): the forms the blocks run.  Forward: a workgroup takes
// 256 >> lg rows per pass (2^lg threads sweep one row), no division per element; backward: workgroup = (bin, 256-channel
// tile), a wave access is 1 KiB of one row, the four waves split the batch rows and meet in LDS in wave order.
// cache policy of the big streams of the kernels below (bit 0: loads nt, bit 1: stores nt); the small factor tables stay cached
#ifndef SMX_TIME_NT
#define SMX_TIME_NT 1        // loads nt: the blocks measured 9.60 -> 9.50 ms (FrequencyNativeBlock), stores nt on top: no change
#endif
__device__ __forceinline__ f32x4 ld4s(const float* p) {
  if constexpr ((SMX_TIME_NT & 1) != 0) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  else return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ void st4s(float* p, f32x4 v) {
  if constexpr ((SMX_TIME_NT & 2) != 0) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
constexpr int PC_PASSES = 8;
__global__ __launch_bounds__(256) void k_pcmul_fwd4(const float* __restrict__ h, const float* __restrict__ fr,
                                                   const float* __restrict__ fi, float* __restrict__ out, int rows,
                                                   int F, int C, int lg) {
  const size_t plane = (size_t)rows * C;
  const int tpr = 1 << lg, rpp = 256 >> lg, cv = C >> 2;
  const int rin = threadIdx.x >> lg, j0 = threadIdx.x & (tpr - 1);
#pragma unroll 2
  for (int ps = 0; ps < PC_PASSES; ++ps) {
    const long long row = ((long long)blockIdx.x * PC_PASSES + ps) * rpp + rin;
    if (row >= rows) break;
    const int f = (int)(row % F);
    for (int j = j0; j < cv; j += tpr) {
      const size_t i = (size_t)row * C + 4 * j, k = (size_t)f * C + 4 * j;
      const f32x4 a = ld4s(h + i), b = ld4s(h + plane + i);
      const f32x4 r = *reinterpret_cast<const f32x4*>(fr + k), q = *reinterpret_cast<const f32x4*>(fi + k);
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = __builtin_fmaf(a[e], r[e], -(b[e] * q[e]));
        o1[e] = __builtin_fmaf(a[e], q[e], b[e] * r[e]);
      }
      st4s(out + i, o0);
      st4s(out + plane + i, o1);
    }
  }
}
__global__ __launch_bounds__(256) void k_pcmul_bwd4(const float* __restrict__ g, const float* __restrict__ h,
                                                   const float* __restrict__ fr, const float* __restrict__ fi,
                                                   float* __restrict__ gh, float* __restrict__ gfr,
                                                   float* __restrict__ gfi, int B, int F, int C) {
  __shared__ f32x4 red[2][4][64];
  const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + 4 * lane;
  const size_t plane = (size_t)B * F * C;
  const bool on = c < C;
  f32x4 r = {0.f, 0.f, 0.f, 0.f}, q = r, ar = r, ai = r;
  if (on) { r = *reinterpret_cast<const f32x4*>(fr + (size_t)f * C + c); q = *reinterpret_cast<const f32x4*>(fi + (size_t)f * C + c); }
  if (on) {
    for (int b = wv; b < B; b += 4) {
      const size_t i = ((size_t)b * F + f) * C + c;
      const f32x4 gr = ld4s(g + i), gi = ld4s(g + plane + i);
      const f32x4 hr = ld4s(h + i), hi = ld4s(h + plane + i);
      if (gh) {
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = __builtin_fmaf(gr[e], r[e], gi[e] * q[e]);
          o1[e] = __builtin_fmaf(gi[e], r[e], -(gr[e] * q[e]));
        }
        st4s(gh + i, o0);
        st4s(gh + plane + i, o1);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ar[e] = __builtin_fmaf(gr[e], hr[e], __builtin_fmaf(gi[e], hi[e], ar[e]));
        ai[e] = __builtin_fmaf(gi[e], hr[e], __builtin_fmaf(-gr[e], hi[e], ai[e]));
      }
    }
  }
  if (gfr || gfi) {
    red[0][wv][lane] = ar; red[1][wv][lane] = ai;
    __syncthreads();
    if (wv == 0 && on) {
      f32x4 a = red[0][0][lane], b2 = red[1][0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) { a += red[0][w][lane]; b2 += red[1][w][lane]; }
      if (gfr) *reinterpret_cast<f32x4*>(gfr + (size_t)f * C + c) = a;
      if (gfi) *reinterpret_cast<f32x4*>(gfi + (size_t)f * C + c) = b2;
    }
  }
}
// y = a + (p_re + i p_im): the residual around the feed-forward (reference :355-356) with the planar result folded in;
// and its backward half, planar planes of a complex gradient
__global__ __launch_bounds__(256) void k_add_planar(const cf* __restrict__ a, const float* __restrict__ p,
                                                   cf* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const cf v = a ? a[i] : mk(0.f, 0.f);
    y[i] = mk(v.x + p[i], v.y + p[n + i]);
  }
}
__global__ __launch_bounds__(256) void k_to_planar(const cf* __restrict__ g, float* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const cf v = g[i];
    p[i] = v.x; p[n + i] = v.y;
  }
}

// ---- the gate chain between the two transforms of the twin blocks --------------------------------------------------------
// reference fft_lm/frequency_native.py:95 (FrequencyConvFunc.forward), :338 (frequency gate x context gate), :351 (cutoff
// mask) -- three (B, F, C) complex passes forward, eleven backward through torch -- in one launch each way:
//     y[b,f,c] = ((((x[b,f,c] a[f]) u[c]) p[f]) q[b,c]) m[f]            a complex (F); u (C), p (F), q (B,C), m (F) real,
// multiplied in the reference's order (a masked bin keeps the signs of its zeros, which SpectralLayerNorm's arg() reads);
// each of u, p, q, m may be absent (= 1).  Backward, with G the gradient of y and W = u p q m:
//     grad_x  = G conj(a) W
//     S1[f]   = sum_{b,c} G conj(x) u q          complex; the caller forms grad_a = p m S1 (= :111), grad_p = m Re(conj(a) S1)
//     Rc[b,c] = sum_f p m Re(conj(a) G conj(x))  grad_q = u Rc (autograd of :338), and grad_u = sum_b q Rc where autograd is meant
//     Rp[b,c] = sum_f p m Re(a G x)              the reference's hand-written grad_gain = sum_b q Rp (:115: no conjugate)
// Workgroup of the backward = (batch row, 128-channel tile), eight waves taking the bins round-robin: Rc / Rp are complete
// inside it (LDS, wave order); S1's per-workgroup partials [B ctiles][F] are added in a fixed order by k_gate_s1.
// complex times real AS TORCH DOES IT: the real factor is promoted to (s + 0i) and the two are multiplied as complex numbers,
// (t.x s - t.y 0) + i (t.x 0 + t.y s).  For s != 0 that is t s; for a masked bin (s = 0) it decides the SIGNS of the zeros --
// real part -0 only for t.x < 0 < t.y, imaginary part -0 only for both negative -- which SpectralLayerNorm's arg() turns into
// a phase of 0 or +-pi (reference :223, :236).  The golden block fixtures (T02, T03) hold exactly these signs.
__device__ __forceinline__ cf gate_scale(cf t, float s) {
  return mk(__builtin_fmaf(t.x, s, -(t.y * 0.f)), __builtin_fmaf(t.x, 0.f, t.y * s));
}
constexpr int GT_ROWS = 16;          // bins per forward workgroup
constexpr int GT_WAVES = 8;
__global__ __launch_bounds__(256) void k_gate_fwd(const cf* __restrict__ x, const cf* __restrict__ a,
                                                 const float* __restrict__ u, const float* __restrict__ p,
                                                 const float* __restrict__ q, const float* __restrict__ m,
                                                 cf* __restrict__ y, int F, int C) {
  const int b = blockIdx.y, f0 = blockIdx.x * GT_ROWS, f1 = min(F, f0 + GT_ROWS);
  for (int j = threadIdx.x; 2 * j < C; j += 256) {
    const int c = 2 * j;
    const float u0 = u ? u[c] : 1.f, u1 = u ? u[c + 1] : 1.f;
    const float q0 = q ? q[(size_t)b * C + c] : 1.f, q1 = q ? q[(size_t)b * C + c + 1] : 1.f;
#pragma unroll 4
    for (int f = f0; f < f1; ++f) {
      const size_t i = ((size_t)b * F + f) * C + c;
      const f32x4 v = ld4s(reinterpret_cast<const float*>(&x[i]));
      const cf af = a[f];
      cf t0 = cmul(mk(v.x, v.y), af), t1 = cmul(mk(v.z, v.w), af);
      if (u) { t0 = gate_scale(t0, u0); t1 = gate_scale(t1, u1); }
      if (p) { const float pf = p[f]; t0 = gate_scale(t0, pf); t1 = gate_scale(t1, pf); }
      if (q) { t0 = gate_scale(t0, q0); t1 = gate_scale(t1, q1); }
      if (m) { const float mf = m[f]; t0 = gate_scale(t0, mf); t1 = gate_scale(t1, mf); }
      f32x4 o; o.x = t0.x; o.y = t0.y; o.z = t1.x; o.w = t1.y;
      st4s(reinterpret_cast<float*>(&y[i]), o);
    }
  }
}
__global__ __launch_bounds__(64 * GT_WAVES) void k_gate_bwd(const cf* __restrict__ g, const cf* __restrict__ x,
                                                           const cf* __restrict__ a, const float* __restrict__ u,
                                                           const float* __restrict__ p, const float* __restrict__ q,
                                                           const float* __restrict__ m, cf* __restrict__ gx,
                                                           cf* __restrict__ part, float* __restrict__ rc_out,
                                                           float* __restrict__ rp_out, int F, int C) {
  __shared__ f32x4 red[GT_WAVES][64];
  const int b = blockIdx.y, ct = blockIdx.x, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = ct * 128 + 2 * lane;
  const bool on = c < C;
  float uq0 = 0.f, uq1 = 0.f;
  if (on) {
    uq0 = (u ? u[c] : 1.f) * (q ? q[(size_t)b * C + c] : 1.f);
    uq1 = (u ? u[c + 1] : 1.f) * (q ? q[(size_t)b * C + c + 1] : 1.f);
  }
  float rc0 = 0.f, rc1 = 0.f, rp0 = 0.f, rp1 = 0.f;
  cf* prt = part + ((size_t)b * gridDim.x + ct) * F;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto bin = [&](int f, const f32x4 xv, const f32x4 gv) {
    const cf af = a[f];
    const float pm = (p ? p[f] : 1.f) * (m ? m[f] : 1.f);
    float s_re = 0.f, s_im = 0.f;
    if (on) {
      // P = G conj(x), Q = G x, per channel
      const float p0r = __builtin_fmaf(gv.x, xv.x, gv.y * xv.y), p0i = __builtin_fmaf(gv.y, xv.x, -(gv.x * xv.y));
      const float p1r = __builtin_fmaf(gv.z, xv.z, gv.w * xv.w), p1i = __builtin_fmaf(gv.w, xv.z, -(gv.z * xv.w));
      const float q0r = __builtin_fmaf(gv.x, xv.x, -(gv.y * xv.y)), q0i = __builtin_fmaf(gv.x, xv.y, gv.y * xv.x);
      const float q1r = __builtin_fmaf(gv.z, xv.z, -(gv.w * xv.w)), q1i = __builtin_fmaf(gv.z, xv.w, gv.w * xv.z);
      rc0 = __builtin_fmaf(pm, __builtin_fmaf(af.x, p0r, af.y * p0i), rc0);
      rc1 = __builtin_fmaf(pm, __builtin_fmaf(af.x, p1r, af.y * p1i), rc1);
      rp0 = __builtin_fmaf(pm, __builtin_fmaf(af.x, q0r, -(af.y * q0i)), rp0);
      rp1 = __builtin_fmaf(pm, __builtin_fmaf(af.x, q1r, -(af.y * q1i)), rp1);
      s_re = __builtin_fmaf(p0r, uq0, p1r * uq1);
      s_im = __builtin_fmaf(p0i, uq0, p1i * uq1);
      if (gx) {
        const float w0 = uq0 * pm, w1 = uq1 * pm;
        f32x4 o;                                        // G conj(a) W
        o.x = __builtin_fmaf(gv.x, af.x, gv.y * af.y) * w0; o.y = __builtin_fmaf(gv.y, af.x, -(gv.x * af.y)) * w0;
        o.z = __builtin_fmaf(gv.z, af.x, gv.w * af.y) * w1; o.w = __builtin_fmaf(gv.w, af.x, -(gv.z * af.y)) * w1;
        st4s(reinterpret_cast<float*>(&gx[((size_t)b * F + f) * C + c]), o);
      }
    }
    s_re = sln_wave_sum(s_re); s_im = sln_wave_sum(s_im);      // lane 63 holds the totals
    if (lane == 63) prt[f] = mk(s_re, s_im);
  };
  for (int f = wv; f < F; f += 2 * GT_WAVES) {                 // two bins in flight per wave
    const int f2 = f + GT_WAVES;
    const bool two = f2 < F;
    f32x4 xa = zero4, ga = zero4, xb = zero4, gb = zero4;
    if (on) {
      const size_t i = ((size_t)b * F + f) * C + c;
      xa = ld4s(reinterpret_cast<const float*>(&x[i])); ga = ld4s(reinterpret_cast<const float*>(&g[i]));
      if (two) {
        const size_t i2 = ((size_t)b * F + f2) * C + c;
        xb = ld4s(reinterpret_cast<const float*>(&x[i2])); gb = ld4s(reinterpret_cast<const float*>(&g[i2]));
      }
    }
    bin(f, xa, ga);
    if (two) bin(f2, xb, gb);
  }
  f32x4 mine; mine.x = rc0; mine.y = rc1; mine.z = rp0; mine.w = rp1;
  red[wv][lane] = mine;
  __syncthreads();
  if (wv == 0 && on) {
    f32x4 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < GT_WAVES; ++w) t += red[w][lane];
    if (rc_out) { rc_out[(size_t)b * C + c] = t.x; rc_out[(size_t)b * C + c + 1] = t.y; }
    if (rp_out) { rp_out[(size_t)b * C + c] = t.z; rp_out[(size_t)b * C + c + 1] = t.w; }
  }
}
// S1[f] = the partials of the nw workgroups in a FIXED order: sixteen lanes per bin take w = l, l + 16, ... (four loads in
// flight each), then the sixteen lane sums are added in lane order.  (One thread per bin walking all nw partials was a chain
// of 256 dependent loads: 100 us at (64, 1025, 512) -- as long as the pass that produced them.)
__global__ __launch_bounds__(256) void k_gate_s1(const cf* __restrict__ part, cf* __restrict__ s1, int nw, int F) {
  __shared__ cf red[16][17];
  const int fl = threadIdx.x & 15, wl = threadIdx.x >> 4;
  const int f = blockIdx.x * 16 + fl;
  float re = 0.f, im = 0.f;
  if (f < F) {
    int w = wl;
    for (; w + 48 < nw; w += 64) {
      const cf v0 = part[(size_t)w * F + f], v1 = part[(size_t)(w + 16) * F + f];
      const cf v2 = part[(size_t)(w + 32) * F + f], v3 = part[(size_t)(w + 48) * F + f];
      re = ((re + v0.x) + v1.x) + (v2.x + v3.x);
      im = ((im + v0.y) + v1.y) + (v2.y + v3.y);
    }
    for (; w < nw; w += 16) { const cf v = part[(size_t)w * F + f]; re += v.x; im += v.y; }
  }
  red[wl][fl] = mk(re, im);
  __syncthreads();
  if (wl == 0 && f < F) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) { a += red[l][fl].x; b += red[l][fl].y; }
    s1[f] = mk(a, b);
  }
}

// ---- BicameralBlock's fusion line (reference fft_lm/bicameral.py:237-268) ------------------------------------------------
//     out = r + w1 a + w2 b + c3 c       r the block input (residual), a / b the spectral / time path, c the cross-talk
// projection; w1, w2 learned scalars in device memory, c3 a constant (0.1).  Six elementwise torch launches forward and
// eight backward (15 + 20 tensor passes) become one each: 5 passes forward; backward reads g, a, b, writes w1 g, w2 g, c3 g
// and reduces sum(g a), sum(g b) -- per-workgroup partials added in workgroup order by k_mix_sum (fixed order).
constexpr int MX_BLOCKS = 2048;
__global__ __launch_bounds__(256) void k_mix_fwd(const float* __restrict__ r, const float* __restrict__ a,
                                                const float* __restrict__ b, const float* __restrict__ c,
                                                const float* __restrict__ w, float c3, float* __restrict__ out,
                                                long long n4) {
  const float w1 = w[0], w2 = w[1];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 rv = ld4s(r + 4 * i), av = ld4s(a + 4 * i), bv = ld4s(b + 4 * i);
    f32x4 o;
    if (c) {
      const f32x4 cv = ld4s(c + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rv[e] + (w1 * av[e] + w2 * bv[e] + c3 * cv[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rv[e] + (w1 * av[e] + w2 * bv[e]);
    }
    st4s(out + 4 * i, o);
  }
}
__global__ __launch_bounds__(256) void k_mix_bwd(const float* __restrict__ g, const float* __restrict__ a,
                                                const float* __restrict__ b, const float* __restrict__ w, float c3,
                                                float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ gc,
                                                float* __restrict__ part, long long n4) {
  __shared__ float red[2][4];
  const float w1 = w[0], w2 = w[1];
  float sa = 0.f, sb = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 gv = ld4s(g + 4 * i), av = ld4s(a + 4 * i), bv = ld4s(b + 4 * i);
    f32x4 o1, o2, o3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sa = __builtin_fmaf(gv[e], av[e], sa);
      sb = __builtin_fmaf(gv[e], bv[e], sb);
      o1[e] = w1 * gv[e]; o2[e] = w2 * gv[e]; o3[e] = c3 * gv[e];
    }
    if (ga) st4s(ga + 4 * i, o1);
    if (gb) st4s(gb + 4 * i, o2);
    if (gc) st4s(gc + 4 * i, o3);
  }
  sa = sln_wave_sum(sa); sb = sln_wave_sum(sb);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 63) { red[0][wv] = sa; red[1][wv] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
__global__ __launch_bounds__(64) void k_mix_sum(const float* __restrict__ part, float* __restrict__ gw, int nb) {
  float sa = 0.f, sb = 0.f;                                        // lane-strided, then one wave sum: fixed order
  for (int i = threadIdx.x; i < nb; i += 64) { sa += part[2 * i]; sb += part[2 * i + 1]; }
  sa = sln_wave_sum(sa); sb = sln_wave_sum(sb);
  if (threadIdx.x == 63) { gw[0] = sa; gw[1] = sb; }
}

// [B ceil(T / 32)][5][C] block partials, then [B][4][C] per-batch-row sums (both 16-byte aligned: C % 4 == 0 on the
// vector path, and the scalar path does not care)
static size_t dw_part_floats(int B, int T, int C) { return (size_t)B * ((T + DW_TR - 1) / DW_TR) * 5 * C; }
size_t dwconv3_workspace_bytes(int B, int T, int C) {
  return (dw_part_floats(B, T, C) + (size_t)B * 4 * C) * sizeof(float);
}

hipError_t launch_dwconv3_fwd(const float* x, const float* w, const float* bias, const float* scale, float* y, int B,
                              int T, int C, hipStream_t s) {
  const bool v4 = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  const int per = DW_TPB * (v4 ? 4 : 1);
  const dim3 grid((unsigned)((C + per - 1) / per) * (unsigned)(B * ((T + DW_TR - 1) / DW_TR)));
  if (v4) hipLaunchKernelGGL(k_dwconv3_fwd<4>, grid, dim3(DW_TPB), 0, s, x, w, bias, scale, y, B, T, C);
  else hipLaunchKernelGGL(k_dwconv3_fwd<1>, grid, dim3(DW_TPB), 0, s, x, w, bias, scale, y, B, T, C);
  return hipGetLastError();
}

hipError_t launch_dwconv3_bwd(const float* g, const float* x, const float* w, const float* bias, const float* scale,
                              float* gx, float* gw, float* gbias, float* gscale, float* part, int B, int T, int C,
                              hipStream_t s) {
  const int nch = (T + DW_TR - 1) / DW_TR;
  const bool v4 = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)gx | (uintptr_t)part) & 15) == 0;
  const int per = DW_TPB * (v4 ? 4 : 1);
  const dim3 grid((unsigned)((C + per - 1) / per) * (unsigned)(B * nch));
  if (v4) hipLaunchKernelGGL(k_dwconv3_bwd<4>, grid, dim3(DW_TPB), 0, s, g, x, w, bias, scale, gx, part, B, T, C);
  else hipLaunchKernelGGL(k_dwconv3_bwd<1>, grid, dim3(DW_TPB), 0, s, g, x, w, bias, scale, gx, part, B, T, C);
  if (gw || gbias || gscale) {
    float* part2 = part + dw_part_floats(B, T, C);
    hipLaunchKernelGGL(k_dwconv3_sum_a, dim3((C + DW_TPB - 1) / DW_TPB, B, 5), dim3(DW_TPB), 0, s, part, part2, gscale,
                       nch, C);
    if (gw || gbias)
      hipLaunchKernelGGL(k_dwconv3_sum_b, dim3((C + DW_TPB - 1) / DW_TPB, 4), dim3(DW_TPB), 0, s, part2, gw, gbias, B, C);
  }
  return hipGetLastError();
}


static inline unsigned ew_blocks(long long n) {
  const long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}
hipError_t launch_pcmul_fwd(const float* h, const float* fr, const float* fi, float* out, int B, int F, int C,
                            hipStream_t s) {
  const long long rows = (long long)B * F;
  const bool v4 = C % 4 == 0 && rows < (1ll << 31) &&
                  (((uintptr_t)h | (uintptr_t)fr | (uintptr_t)fi | (uintptr_t)out) & 15) == 0;
  if (v4) {
    int lg = 0;                                        // threads per row = 2^lg >= C / 4, at most 256
    while ((1 << lg) < C / 4 && lg < 8) ++lg;
    const int rpp = 256 >> lg;                         // rows per pass of the workgroup
    const long long blocks = (rows + (long long)rpp * PC_PASSES - 1) / ((long long)rpp * PC_PASSES);
    hipLaunchKernelGGL(k_pcmul_fwd4, dim3((unsigned)blocks), dim3(256), 0, s, h, fr, fi, out, (int)rows, F, C, lg);
  } else {
    hipLaunchKernelGGL(k_pcmul_fwd, dim3(ew_blocks(rows * C)), dim3(256), 0, s, h, fr, fi, out, rows, F, C);
  }
  return hipGetLastError();
}
hipError_t launch_pcmul_bwd(const float* g, const float* h, const float* fr, const float* fi, float* gh, float* gfr,
                            float* gfi, int B, int F, int C, hipStream_t s) {
  const bool v4 = C % 4 == 0 && (((uintptr_t)g | (uintptr_t)h | (uintptr_t)fr | (uintptr_t)fi | (uintptr_t)gh |
                                   (uintptr_t)gfr | (uintptr_t)gfi) & 15) == 0;
  if (v4) hipLaunchKernelGGL(k_pcmul_bwd4, dim3(F, (C + 255) / 256), dim3(256), 0, s, g, h, fr, fi, gh, gfr, gfi, B, F, C);
  else hipLaunchKernelGGL(k_pcmul_bwd, dim3(F), dim3(256), 0, s, g, h, fr, fi, gh, gfr, gfi, B, F, C);
  return hipGetLastError();
}
hipError_t launch_add_planar(const cf* a, const float* p, cf* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_add_planar, dim3(ew_blocks(n)), dim3(256), 0, s, a, p, y, n);
  return hipGetLastError();
}
hipError_t launch_to_planar(const cf* g, float* p, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_to_planar, dim3(ew_blocks(n)), dim3(256), 0, s, g, p, n);
  return hipGetLastError();
}

size_t gate_workspace_bytes(int B, int F, int C) { return (size_t)B * ((C + 127) / 128) * F * sizeof(cf); }
hipError_t launch_gate_fwd(const cf* x, const cf* a, const float* u, const float* p, const float* q, const float* m, cf* y,
                           int B, int F, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_gate_fwd, dim3((F + GT_ROWS - 1) / GT_ROWS, B), dim3(256), 0, s, x, a, u, p, q, m, y, F, C);
  return hipGetLastError();
}
hipError_t launch_gate_bwd(const cf* g, const cf* x, const cf* a, const float* u, const float* p, const float* q,
                           const float* m, cf* gx, cf* s1, float* rc, float* rp, cf* part, int B, int F, int C,
                           hipStream_t s) {
  const int ct = (C + 127) / 128;
  hipLaunchKernelGGL(k_gate_bwd, dim3(ct, B), dim3(64 * GT_WAVES), 0, s, g, x, a, u, p, q, m, gx, part, rc, rp, F, C);
  if (s1) hipLaunchKernelGGL(k_gate_s1, dim3((F + 15) / 16), dim3(256), 0, s, part, s1, B * ct, F);
  return hipGetLastError();
}

static unsigned mix_blocks(long long n4) {
  const long long b = (n4 + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > MX_BLOCKS ? MX_BLOCKS : b);
}
size_t mix_workspace_bytes() { return (size_t)MX_BLOCKS * 2 * sizeof(float); }
hipError_t launch_mix_fwd(const float* r, const float* a, const float* b, const float* c, const float* w, float c3,
                          float* out, long long n, hipStream_t s) {
  hipLaunchKernelGGL(k_mix_fwd, dim3(mix_blocks(n / 4)), dim3(256), 0, s, r, a, b, c, w, c3, out, n / 4);
  return hipGetLastError();
}
hipError_t launch_mix_bwd(const float* g, const float* a, const float* b, const float* w, float c3, float* ga, float* gb,
                          float* gc, float* gw, float* part, long long n, hipStream_t s) {
  const unsigned nb = mix_blocks(n / 4);
  hipLaunchKernelGGL(k_mix_bwd, dim3(nb), dim3(256), 0, s, g, a, b, w, c3, ga, gb, gc, part, n / 4);
  if (gw) hipLaunchKernelGGL(k_mix_sum, dim3(1), dim3(64), 0, s, part, gw, (int)nb);
  return hipGetLastError();
}

}  // namespace smx
