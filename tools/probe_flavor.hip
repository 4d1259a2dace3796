// probe_flavor.hip -- load/store cache-policy flavours on a footprint far beyond the 256 MiB MALL.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ST(name, mods) \
__global__ void st_##name(f32x4* __restrict__ b, size_t n) { \
  f32x4 v = {1, 2, 3, 4}; \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { \
    f32x4* p = b + i; asm volatile("global_store_dwordx4 %0, %1, off " mods :: "v"(p), "v"(v) : "memory"); } }
ST(plain, "")
ST(nt, "nt")
ST(sc0, "sc0")
ST(sc1, "sc1")
ST(sc01, "sc0 sc1")
ST(ntsc1, "sc1 nt")
ST(ntsc01, "sc0 sc1 nt")
#define LD(name, mods) \
__global__ void ld_##name(const f32x4* __restrict__ a, float* __restrict__ o, size_t n) { \
  f32x4 s = {0, 0, 0, 0}; \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { \
    const f32x4* p = a + i; f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off " mods "\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); s += v; } \
  if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1; }
// (the waitcnt per load serialises; loads are covered by the builtin variants below instead)
__global__ void ld_plain(const f32x4* __restrict__ a, float* __restrict__ o, size_t n) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1; }
__global__ void ld_nt(const f32x4* __restrict__ a, float* __restrict__ o, size_t n) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(a + i);
  if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1; }
__global__ void cp_nt(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); }
__global__ void cp_plain(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
template <typename F> float timeit(F f, int iters = 6) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize(); float best = 1e9;
  for (int i = 0; i < iters; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best; }
int main() {
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;     // 2 GiB per buffer
  f32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  float* o; hipMalloc(&o, 64);
  const int blocks = 2048;
#define RUNST(name) { float ms = timeit([&] { hipLaunchKernelGGL(st_##name, dim3(blocks), dim3(256), 0, 0, b, n); }); printf("store %-8s : %6.0f GB/s\n", #name, bytes / ms / 1e6); }
  RUNST(plain) RUNST(nt) RUNST(sc0) RUNST(sc1) RUNST(sc01) RUNST(ntsc1) RUNST(ntsc01)
  { float ms = timeit([&] { hipLaunchKernelGGL(ld_plain, dim3(blocks), dim3(256), 0, 0, a, o, n); }); printf("load  plain    : %6.0f GB/s\n", bytes / ms / 1e6); }
  { float ms = timeit([&] { hipLaunchKernelGGL(ld_nt, dim3(blocks), dim3(256), 0, 0, a, o, n); }); printf("load  nt       : %6.0f GB/s\n", bytes / ms / 1e6); }
  { float ms = timeit([&] { hipLaunchKernelGGL(cp_plain, dim3(blocks), dim3(256), 0, 0, a, b, n); }); printf("copy  plain    : %6.0f GB/s (rd+wr)\n", 2 * bytes / ms / 1e6); }
  { float ms = timeit([&] { hipLaunchKernelGGL(cp_nt, dim3(blocks), dim3(256), 0, 0, a, b, n); }); printf("copy  nt       : %6.0f GB/s (rd+wr)\n", 2 * bytes / ms / 1e6); }
  return 0; }
