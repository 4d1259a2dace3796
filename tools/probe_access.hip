// probe_access.hip -- HBM access-pattern probe for the decimated layout (tuning tool, not product).
// Measures read / write / copy bandwidth when each 256-thread workgroup owns (batch row, SEG bytes
// of every 1 KiB row) and walks the rows with stride L, exactly like smx_decim.hip does.
//   hipcc --offload-arch=gfx950 -O3 -o probe_access probe_access.hip && ./probe_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int B = 64, N = 4096, D = 256, L = N / 256;

__global__ void k_copy4(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    b[i] = a[i];
}
__global__ void k_read4(const f32x4* __restrict__ a, float* __restrict__ o, size_t n) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += __builtin_nontemporal_load(a + i);
  if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1;
}
__global__ void k_write4(f32x4* __restrict__ b, size_t n) {
  f32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, b + i);
}

// MODE 0 read, 1 write, 2 read-all-then-write-all (like the fused kernel)
template <typename V, int SEGB, int PF, int MODE, int NT>
__global__ __launch_bounds__(256) void k_pat(const float* __restrict__ in, float* __restrict__ out,
                                             int stagger) {
  constexpr int VB = sizeof(V), LPR = SEGB / VB, RP = 256 / LPR, NL = 256 / RP;   // loads per tile
  constexpr int NSEG = 1024 / SEGB;
  const int tid = threadIdx.x, lane = tid % LPR, row0 = tid / LPR;
  int bid = blockIdx.x, b = bid / NSEG, seg = bid % NSEG;
  if ((stagger & 2) && NSEG == 8) { seg = (bid / 8) % 8; b = (bid % 8) + 8 * (bid / 64); }   // siblings share an XCD
  const size_t base = (size_t)b * N * D + (size_t)seg * (SEGB / 4) + (size_t)lane * (VB / 4);
  const int r0 = (stagger & 1) ? (bid * 7) % L : 0;
  V acc = {};
  if (MODE == 0 || MODE == 2) {
    V buf[PF][NL];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      int r = (r0 + p) % L;
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        const V* q = (const V*)(in + base + ((size_t)(row0 + RP * u) * L + r) * D);
        buf[p][u] = (NT & 1) ? __builtin_nontemporal_load(q) : *q;
      }
    }
    for (int i = 0; i < L; i += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int u = 0; u < NL; ++u) acc += buf[p][u];
        int r = (r0 + i + p + PF) % L;
        if (i + p + PF < L) {
#pragma unroll
          for (int u = 0; u < NL; ++u) {
            const V* q = (const V*)(in + base + ((size_t)(row0 + RP * u) * L + r) * D);
            buf[p][u] = (NT & 1) ? __builtin_nontemporal_load(q) : *q;
          }
        }
      }
    }
  }
  if (MODE == 0) {
    float s = 0;
    for (int i = 0; i < VB / 4; ++i) s += acc[i];
    if (s == 123.456f) out[0] = s;
    return;
  }
  if (MODE == 2) __syncthreads();
  for (int i = 0; i < L; ++i) {
    int r = (r0 + i) % L;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      V* q = (V*)(out + base + ((size_t)(row0 + RP * u) * L + r) * D);
      V v = acc + (float)i;
      if (NT & 2) __builtin_nontemporal_store(v, q); else *q = v;
    }
  }
}


// H2: one 512-thread workgroup = two 256-thread halves owning adjacent 128-B column slices.
// Reads: both halves walk the SAME residue r (each row is touched as 256 contiguous bytes by two waves
// at about the same time).  Writes: half h walks residue r + h*L/2 (128-B segments, like the 1-half kernel).
template <int MODE>
__global__ __launch_bounds__(512) void k_pat_h2(const float* __restrict__ in, float* __restrict__ out, int stagger) {
  const int tid = threadIdx.x & 255, h = threadIdx.x >> 8, lane = tid & 15, row0 = tid >> 4;
  const int bid = blockIdx.x, b = bid / 4, seg = (bid % 4) * 2 + h;
  const size_t base = (size_t)b * N * D + (size_t)seg * 32 + (size_t)lane * 2;
  const int r0 = stagger ? (bid * 7) % L : 0;
  f32x2 acc = {0, 0};
  if (MODE == 0 || MODE == 2) {
    f32x2 buf[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) buf[u] = __builtin_nontemporal_load((const f32x2*)(in + base + ((size_t)(row0 + 16 * u) * L + r0) * D));
    for (int i = 0; i < L; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += buf[u];
      int r = (r0 + i + 1) % L;
      if (i + 1 < L) {
#pragma unroll
        for (int u = 0; u < 16; ++u) buf[u] = __builtin_nontemporal_load((const f32x2*)(in + base + ((size_t)(row0 + 16 * u) * L + r) * D));
      }
      __syncthreads();      // keeps the two halves in lockstep, like the exchange barrier would
    }
  }
  if (MODE == 0) { if (acc.x + acc.y == 123.456f) out[0] = acc.x; return; }
  for (int i = 0; i < L; ++i) {
    int r = (i + h * (L / 2)) % L;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x2 v = acc + (float)i;
      __builtin_nontemporal_store(v, (f32x2*)(out + base + ((size_t)(row0 + 16 * u) * L + r) * D));
    }
    __syncthreads();
  }
}

template <typename F>
float timeit(F f, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); f();
  hipDeviceSynchronize();
  float best = 1e9;
  for (int i = 0; i < iters; ++i) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

template <typename V, int SEGB, int PF, int NT>
void run_pat(const float* in, float* out, const char* vname) {
  constexpr int NSEG = 1024 / SEGB;
  const double bytes = (double)B * N * D * 4;
  for (int st = 0; st < 4; ++st) {
    float r = timeit([&] { hipLaunchKernelGGL((k_pat<V, SEGB, PF, 0, NT>), dim3(B * NSEG), dim3(256), 0, 0, in, out, st); });
    float w = timeit([&] { hipLaunchKernelGGL((k_pat<V, SEGB, PF, 1, NT>), dim3(B * NSEG), dim3(256), 0, 0, in, out, st); });
    float c = timeit([&] { hipLaunchKernelGGL((k_pat<V, SEGB, PF, 2, NT>), dim3(B * NSEG), dim3(256), 0, 0, in, out, st); });
    printf("pat vec=%s seg=%4dB pf=%d nt=%d stagger=%d wgs=%4d : read %6.0f GB/s  write %6.0f GB/s  read+write %6.0f GB/s (%.1f us)\n",
           vname, SEGB, PF, NT, st, B * NSEG, bytes / r / 1e6, bytes / w / 1e6, 2 * bytes / c / 1e6, c * 1e3);
  }
}

int main() {
  const size_t n = (size_t)B * N * D;
  float *in, *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
  CK(hipMemset(in, 0, n * 4)); CK(hipMemset(out, 0, n * 4));
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 0.001f;
  for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(in + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const double bytes = (double)n * 4;
  for (int blocks : {1024, 2048, 4096, 8192}) {
    float c = timeit([&] { hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, 0, (const f32x4*)in, (f32x4*)out, n / 4); });
    float r = timeit([&] { hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, (const f32x4*)in, out, n / 4); });
    float w = timeit([&] { hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(256), 0, 0, (f32x4*)out, n / 4); });
    printf("linear float4 blocks=%5d : copy %6.0f GB/s  read %6.0f GB/s  write %6.0f GB/s\n", blocks,
           2 * bytes / c / 1e6, bytes / r / 1e6, bytes / w / 1e6);
  }
  // NT bit0: nontemporal loads, bit1: nontemporal stores
  run_pat<f32x2, 128, 1, 3>(in, out, "f2");
  run_pat<f32x2, 128, 1, 3>(in, out, "f2");
  return 0;
}
