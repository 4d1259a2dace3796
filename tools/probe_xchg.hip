// probe_xchg.hip -- the exchange between the two radix-16 passes (a 16 x 16 transpose between a thread's register index
// and its row-group index), two ways, alone:
//   A  as the product does it: 256 threads = 16 row groups x 16 channel pairs, through a double-buffered 32 KiB LDS tile,
//      one barrier per exchange;
//   B  wave-local (the review's variant): the 16 row groups inside one wave (16 x 4 channel pairs), four butterfly stages
//      of cross-lane moves (__shfl_xor = ds_bpermute_b32 on gfx950) + selects, no barrier, no LDS allocation.
// Both keep 16 complex values per thread live and do one fma per component between exchanges (so the compiler cannot fold
// two transposes).  hipcc --offload-arch=gfx950 -O3 -o probe_xchg tools/probe_xchg.hip ; ./probe_xchg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct cf { float x, y; };
constexpr int ITER = 2000;

__global__ __launch_bounds__(256, 2) void k_lds(const cf* __restrict__ in, cf* __restrict__ out, float c) {
  __shared__ cf E[2][16 * 16 * 16 + 16];
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  cf v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = in[((size_t)blockIdx.x * 16 + u) * 256 + tid];
  for (int it = 0; it < ITER; ++it) {
    cf* e = E[it & 1];
#pragma unroll
    for (int u = 0; u < 16; ++u) e[(u * 16 + t) * 16 + j] = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const cf w = e[(t * 16 + u) * 16 + j];
      v[u].x = __builtin_fmaf(w.x, c, 1e-9f); v[u].y = __builtin_fmaf(w.y, c, 1e-9f);
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) out[((size_t)blockIdx.x * 16 + u) * 256 + tid] = v[u];
}

// C  a single 32 KiB exchange buffer: a second barrier per exchange (the reads of one tile must finish before the next writes)
__global__ __launch_bounds__(256, 2) void k_lds1(const cf* __restrict__ in, cf* __restrict__ out, float c) {
  __shared__ cf E[16 * 16 * 16 + 16];
  const int tid = threadIdx.x, j = tid & 15, t = tid >> 4;
  cf v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = in[((size_t)blockIdx.x * 16 + u) * 256 + tid];
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) E[(u * 16 + t) * 16 + j] = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const cf w = E[(t * 16 + u) * 16 + j];
      v[u].x = __builtin_fmaf(w.x, c, 1e-9f); v[u].y = __builtin_fmaf(w.y, c, 1e-9f);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) out[((size_t)blockIdx.x * 16 + u) * 256 + tid] = v[u];
}

__global__ __launch_bounds__(256, 2) void k_shfl(const cf* __restrict__ in, cf* __restrict__ out, float c) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = lane >> 2;                                 // row group inside the wave: lanes 4 t .. 4 t + 3
  cf v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = in[((size_t)blockIdx.x * 16 + u) * 256 + tid];
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
      const bool hi = (t & s) != 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u & s) continue;
        const cf a = v[u], b = v[u | s];
        const float sx = hi ? a.x : b.x, sy = hi ? a.y : b.y;
        const float rx = __shfl_xor(sx, 4 * s, 64), ry = __shfl_xor(sy, 4 * s, 64);
        if (hi) { v[u].x = rx; v[u].y = ry; } else { v[u | s].x = rx; v[u | s].y = ry; }
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { v[u].x = __builtin_fmaf(v[u].x, c, 1e-9f); v[u].y = __builtin_fmaf(v[u].y, c, 1e-9f); }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) out[((size_t)blockIdx.x * 16 + u) * 256 + tid] = v[u];
}

int main() {
  const int blocks = 512;
  const size_t n = (size_t)blocks * 16 * 256;
  cf *in, *out;
  hipMalloc(&in, n * sizeof(cf)); hipMalloc(&out, n * sizeof(cf));
  std::vector<cf> h(n);
  for (size_t i = 0; i < n; ++i) { h[i].x = (float)(i % 97) * 0.01f; h[i].y = (float)(i % 89) * 0.02f; }
  hipMemcpy(in, h.data(), n * sizeof(cf), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 3; ++which) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, in, out, 0.999f);
      else if (which == 1) hipLaunchKernelGGL(k_shfl, dim3(blocks), dim3(256), 0, 0, in, out, 0.999f);
      else hipLaunchKernelGGL(k_lds1, dim3(blocks), dim3(256), 0, 0, in, out, 0.999f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // 512 workgroups = 2 per CU resident at once: every CU runs 2 x ITER exchanges of a 256-thread tile concurrently
    printf("%s: %.3f ms for %d exchanges per workgroup, 2 workgroups per CU -> %.1f ns per tile exchange per CU-pair slot\n",
           which == 0 ? "A  LDS tile + barrier     " : which == 1 ? "B  wave-local shuffles    " : "C  one LDS buffer, 2 barriers", best, ITER, best * 1e6f / ITER);
  }
  // correctness of B against A: two transposes are the identity up to the fma chain, so compare the outputs of both kernels
  std::vector<cf> oa(n), ob(n);
  hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, in, out, 0.999f); hipMemcpy(oa.data(), out, n * sizeof(cf), hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_shfl, dim3(blocks), dim3(256), 0, 0, in, out, 0.999f); hipMemcpy(ob.data(), out, n * sizeof(cf), hipMemcpyDeviceToHost);
  // (A transposes across the workgroup's 16 row groups t = tid >> 4, B across a wave's t = lane >> 2: different partners, so
  //  only the value multiset per (block, j-class) is comparable -- check sums)
  double sa = 0, sb = 0;
  for (size_t i = 0; i < n; ++i) { sa += oa[i].x + oa[i].y; sb += ob[i].x + ob[i].y; }
  printf("checksums A %.6e  B %.6e (ITER even: both return every value to its owner, so they agree)\n", sa, sb);
  return 0;
}
