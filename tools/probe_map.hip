// probe_map.hip -- search over workgroup -> (batch row, d-tile, first residue) placements for the
// decimated access pattern (smx layout: 64 batch rows x 4 MiB, rows 1 KiB, 8 d-tiles of 128 B,
// tile = 256 rows at 16 KiB pitch, 16 residues 1 KiB apart).  Table-driven so many candidates run
// in one process; reports read and write bandwidth of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Item { int b, dt, r0; };
template <int WRITE>
__global__ __launch_bounds__(256) void k(char* __restrict__ buf, float* __restrict__ out, const Item* __restrict__ items) {
  const int tid = threadIdx.x, lane = tid & 15, row0 = tid >> 4;
  const Item it = items[blockIdx.x];
  char* base = buf + (size_t)it.b * (4u << 20) + (size_t)it.dt * 128 + (size_t)lane * 8;
  if (WRITE) {
    f32x2 v = {1.f, 2.f};
    for (int i = 0; i < 16; ++i) {
      const int r = (it.r0 + i) & 15;
#pragma unroll
      for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(v, (f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)r * 1024));
    }
    return;
  }
  f32x2 acc = {0, 0}, b2[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) b2[u] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)it.r0 * 1024));
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += b2[u];
    const int r = (it.r0 + i + 1) & 15;
    if (i < 15) {
#pragma unroll
      for (int u = 0; u < 16; ++u) b2[u] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)r * 1024));
    }
  }
  if (acc.x + acc.y == 123.456f) out[0] = acc.x;
}
// residue PAIRS per step: the workgroup walks 8 steps, each covering rows of residues (2i, 2i+1)
template <int WRITE>
__global__ __launch_bounds__(256) void kp(char* __restrict__ buf, float* __restrict__ out, const Item* __restrict__ items) {
  const int tid = threadIdx.x, lane = tid & 15, row0 = tid >> 4;
  const Item it = items[blockIdx.x];
  char* base = buf + (size_t)it.b * (4u << 20) + (size_t)it.dt * 128 + (size_t)lane * 8;
  f32x2 acc = {0, 0};
  for (int i = 0; i < 8; ++i) {
    const int r = ((it.r0 & 14) + 2 * i) & 15;
    if (WRITE) {
      f32x2 v = {1.f, 2.f};
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        __builtin_nontemporal_store(v, (f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)r * 1024));
        __builtin_nontemporal_store(v, (f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)(r + 1) * 1024));
      }
    } else {
      f32x2 b2[32];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        b2[2 * u] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)r * 1024));
        b2[2 * u + 1] = __builtin_nontemporal_load((const f32x2*)(base + (size_t)(row0 + 16 * u) * 16384 + (size_t)(r + 1) * 1024));
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += b2[u];
    }
  }
  if (!WRITE && acc.x + acc.y == 123.456f) out[0] = acc.x;
}
int main() {
  char* buf; float* out; Item* d_items;
  hipMalloc(&buf, (size_t)256 << 20); hipMalloc(&out, 64); hipMalloc(&d_items, 512 * sizeof(Item)); hipMemset(buf, 0, (size_t)256 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  struct Cand { const char* name; std::function<Item(int)> f; };
  auto xl = [](int bid, int& x, int& dt, int& bb) { x = bid & 7; int l = bid >> 3; dt = l & 7; bb = l >> 3; };
  std::vector<Cand> cands = {
    {"b-major, r0=0 (old default)          ", [](int bid) { return Item{bid >> 3, bid & 7, 0}; }},
    {"xcd: b=x+8bb r0=bb+8(dt&1)  (current)", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (bb + 8 * (dt & 1)) & 15}; }},
    {"xcd: + 2x skew between XCDs           ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (bb + 8 * (dt & 1) + 2 * x) & 15}; }},
    {"xcd: + x skew between XCDs            ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (bb + 8 * (dt & 1) + x) & 15}; }},
    {"xcd: r0 = 2bb + (dt&1)                ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (2 * bb + (dt & 1)) & 15}; }},
    {"xcd: b=(x+bb)%8+8bb                   ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{((x + bb) & 7) + 8 * bb, dt, (bb + 8 * (dt & 1)) & 15}; }},
    {"xcd: dt=(l+x)%8 (rotate dt per XCD)   ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, (dt + x) & 7, (bb + 8 * (dt & 1)) & 15}; }},
    {"xcd: r0 = 4*(dt>>1)+... full 16 spread ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (bb + 8 * (dt & 1) + 4 * (dt >> 1)) & 15}; }},
    {"xcd: r0 = bit-reversed bb              ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); int rb = ((bb & 1) << 2) | (bb & 2) | ((bb >> 2) & 1); return Item{x + 8 * bb, dt, (2 * rb + (dt & 1)) & 15}; }},
    {"xcd: r0 = (bb*5 + 8(dt&1)+3x)          ", [&](int bid) { int x, dt, bb; xl(bid, x, dt, bb); return Item{x + 8 * bb, dt, (bb * 5 + 8 * (dt & 1) + 3 * x) & 15}; }},
    {"dt-major per XCD: dt = x, 64 b         ", [&](int bid) { int x = bid & 7, l = bid >> 3; return Item{l, x, l & 15}; }},
    {"dt-major per XCD, r0 = (l + 4(l>>4))   ", [&](int bid) { int x = bid & 7, l = bid >> 3; return Item{l, x, (l + 4 * (l >> 4)) & 15}; }},
  };
  for (int rep = 0; rep < 2; ++rep)
  for (auto& c : cands) {
    std::vector<Item> h(512); std::vector<int> seen(512, 0);
    for (int i = 0; i < 512; ++i) { h[i] = c.f(i); seen[h[i].b * 8 + h[i].dt]++; }
    bool ok = true; for (int v : seen) ok &= (v == 1);
    hipMemcpy(d_items, h.data(), 512 * sizeof(Item), hipMemcpyHostToDevice);
    float best[2] = {1e9f, 1e9f};
    for (int w = 0; w < 2; ++w) for (int it = 0; it < 10; ++it) {
      hipEventRecord(a);
      if (w) hipLaunchKernelGGL((k<1>), dim3(512), dim3(256), 0, 0, buf, out, d_items);
      else hipLaunchKernelGGL((k<0>), dim3(512), dim3(256), 0, 0, buf, out, d_items);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (it > 1 && ms < best[w]) best[w] = ms;
    }
    float bp[2] = {1e9f, 1e9f};
    for (int w = 0; w < 2; ++w) for (int it = 0; it < 10; ++it) {
      hipEventRecord(a);
      if (w) hipLaunchKernelGGL((kp<1>), dim3(512), dim3(256), 0, 0, buf, out, d_items);
      else hipLaunchKernelGGL((kp<0>), dim3(512), dim3(256), 0, 0, buf, out, d_items);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (it > 1 && ms < bp[w]) bp[w] = ms;
    }
    printf("   residue pairs: read %5.1f us %5.0f GB/s | write %5.1f us %5.0f GB/s\n", bp[0] * 1e3, 268.435456 / bp[0], bp[1] * 1e3, 268.435456 / bp[1]);
    printf("%s %s read %5.1f us %5.0f GB/s | write %5.1f us %5.0f GB/s | sum %5.1f us\n", c.name, ok ? "  " : "!!", best[0] * 1e3,
           268.435456 / best[0], best[1] * 1e3, 268.435456 / best[1], (best[0] + best[1]) * 1e3);
  }
  return 0;
}
