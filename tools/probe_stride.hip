// probe_stride.hip -- which part of the decimated tile pattern costs read bandwidth on MI355X?
// 512 workgroups x 256 threads; workgroup (b, dt) reads, for r = 0..15, 256 segments of 128 B:
//   addr = b*BS + dt*DS + m*pitch + r*RS   (m = 0..255), 16 lanes x float2 per segment, nt loads.
// The smx layout is BS = 4 MiB, DS = 128 B, pitch = 16 KiB, RS = 1 KiB.  256 MiB are read in every config.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
// rot: 0 = b-major blocks, all workgroups walk r = 0,1,2..   1 = same blocks, r0 = 7*bid
//      2 = XCD-aware: the 64 workgroups that share an XCD (bid % 8) cover all 8 d-tiles and 4 residue
//          phases, i.e. all 16 values of address bits [8..11], 4 workgroups each
//      3 = like 2 but r0 spreads over all 16 residues
template <int WRITE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ in, float* __restrict__ out, size_t BS, size_t DS,
                                         size_t pitch, size_t RS, int rot) {
  const int tid = threadIdx.x, lane = tid & 15, row0 = tid >> 4;
  int b = blockIdx.x >> 3, dt = blockIdx.x & 7, r0 = 0;
  if (rot == 1) r0 = (blockIdx.x * 7) & 15;
  if (rot >= 2) {
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3, bb = l >> 3;
    dt = l & 7; b = x + 8 * bb;
    r0 = rot == 2 ? (bb & 3) + 4 * (dt & 1) : (bb & 3) + 4 * (dt & 1) + 8 * ((bb >> 2) & 1);
    if (rot == 4) r0 = bb + 8 * (dt & 1);                  // all 64 (dt>>1, r0) pairs distinct per XCD
    if (rot == 5) { b = 8 * x + bb; r0 = bb + 8 * (dt & 1); }   // same, consecutive batch rows per XCD
    if (rot == 6) r0 = (2 * bb + (dt & 1)) & 15;           // interleave the d-tile parity into the low bit
    if (rot == 7) r0 = (bb + 8 * (dt & 1) + 4 * (dt >> 1)) & 15;   // also skew by the d-tile pair
  }
  const char* base = in + (size_t)b * BS + (size_t)dt * DS + (size_t)lane * 8;
  if (WRITE) {
    f32x2 v = {1.f, 2.f};
    for (int i = 0; i < 16; ++i) {
      const int r = (r0 + i) & 15;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        __builtin_nontemporal_store(v, (f32x2*)(const_cast<char*>(base) + (size_t)(row0 + 16 * u) * pitch + (size_t)r * RS));
    }
    return;
  }
  f32x2 acc = {0, 0};
  f32x2 buf[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[u] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * pitch + (size_t)r0 * RS));
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += buf[u];
    const int r = (r0 + i + 1) & 15;
    if (i < 15) {
#pragma unroll
      for (int u = 0; u < 16; ++u) buf[u] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * pitch + (size_t)r * RS));
    }
  }
  if (acc.x + acc.y == 123.456f) out[0] = acc.x;
}
int main() {
  const size_t bytes = (size_t)3 << 30;
  char* in; float* out; hipMalloc(&in, bytes); hipMalloc(&out, 64); hipMemset(in, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t K = 1024, M = 1024 * 1024;
  struct Cfg { const char* name; size_t BS, DS, pitch, RS; };
  Cfg cfgs[] = {
    {"smx layout      pitch 16K  RS 1K   ", 4 * M, 128, 16 * K, 1 * K},
  };
  for (int rep = 0; rep < 2; ++rep)
  for (auto& c : cfgs) for (int rot = 0; rot < 8; ++rot) {
    float best[2] = {1e9f, 1e9f};
    for (int w = 0; w < 2; ++w)
      for (int it = 0; it < 10; ++it) {
        hipEventRecord(a);
        if (w) hipLaunchKernelGGL((k<1>), dim3(512), dim3(256), 0, 0, in, out, c.BS, c.DS, c.pitch, c.RS, rot);
        else hipLaunchKernelGGL((k<0>), dim3(512), dim3(256), 0, 0, in, out, c.BS, c.DS, c.pitch, c.RS, rot);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (it > 1 && ms < best[w]) best[w] = ms;
      }
    printf("%s map=%d : read %6.1f us %5.0f GB/s   write %6.1f us %5.0f GB/s\n", c.name, rot, best[0] * 1e3,
           268.435456 / best[0], best[1] * 1e3, 268.435456 / best[1]);
  }
  return 0;
}
