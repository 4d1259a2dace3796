// probe_mall.hip -- does the 256 MiB Infinity Cache keep what a kernel streamed?  For a footprint S:
//   flush (stream 1 GiB of something else) ; pass A = read S forward ; [optional: write S elsewhere] ;
//   pass B = read S again in REVERSE order (most recently read first).  B at >> HBM rate = hits.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT, bool REV>
__global__ void rd(const f32x4* __restrict__ a, float* __restrict__ o, size_t n) {
  f32x4 s = {0, 0, 0, 0};
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const size_t k = REV ? n - 1 - i : i;
    s += NT ? __builtin_nontemporal_load(a + k) : a[k];
  }
  if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1;
}
template <bool NT>
__global__ void wr(f32x4* __restrict__ b, size_t n) {
  f32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
static float ev(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }
int main() {
  const size_t big = (size_t)1 << 30;
  f32x4 *a, *f, *w; hipMalloc(&a, big); hipMalloc(&f, big); hipMalloc(&w, big);
  hipMemset(a, 0, big); hipMemset(f, 0, big); hipMemset(w, 0, big);
  float* o; hipMalloc(&o, 64);
  hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  const dim3 g(2048), b(256);
  const int sizes[] = {32, 64, 128, 192, 256, 320, 512};
  for (int mode = 0; mode < 4; ++mode) {          // 0: plain A, no writes; 1: nt A; 2: plain A + nt writes; 3: plain A + plain writes
    for (int s : sizes) {
      const size_t n = (size_t)s * (1 << 20) / 16;
      float best_a = 1e9, best_b = 1e9;
      for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((rd<false, false>), g, b, 0, 0, f, o, big / 16);     // flush
        hipEventRecord(e0);
        if (mode == 1) hipLaunchKernelGGL((rd<true, false>), g, b, 0, 0, a, o, n);
        else hipLaunchKernelGGL((rd<false, false>), g, b, 0, 0, a, o, n);
        hipEventRecord(e1);
        if (mode == 2) hipLaunchKernelGGL((wr<true>), g, b, 0, 0, w, n);
        if (mode == 3) hipLaunchKernelGGL((wr<false>), g, b, 0, 0, w, n);
        hipEventRecord(e2);
        hipLaunchKernelGGL((rd<true, true>), g, b, 0, 0, a, o, n);
        hipEventRecord(e3); hipEventSynchronize(e3);
        const float ta = ev(e0, e1), tb = ev(e2, e3);
        if (ta < best_a) best_a = ta;
        if (tb < best_b) best_b = tb;
      }
      printf("mode %d  S=%3d MiB  first read %6.0f GB/s   re-read (reverse) %6.0f GB/s\n", mode, s,
             s * 1.048576 / best_a, s * 1.048576 / best_b);
    }
  }
  return 0;
}
