// smx_time.hip -- the time path of fft_lm's BicameralBlock on the (B, T, C) layout the rest of the block lives in.
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

}  // namespace

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

}  // namespace smx
