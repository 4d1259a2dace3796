// probe_valu.hip -- VALU issue/throughput probe: cycles per v_fma_f32 vs ILP and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, bool PK>
__global__ void k(float* out, int iters, float a, float b) {
  float x[ILP * 2];
#pragma unroll
  for (int i = 0; i < ILP * 2; ++i) x[i] = threadIdx.x * 0.001f + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if (PK) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = {x[2 * i], x[2 * i + 1]}, aa = {a, a}, bb = {b, b};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(aa), "v"(bb));
          x[2 * i] = v.x; x[2 * i + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < ILP * 2; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(a), "v"(b));
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP * 2; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int ILP, bool PK>
void run(float* out, int waves_per_simd) {
  // one block per CU: block = waves_per_simd*4 waves
  int threads = waves_per_simd * 4 * 64, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ILP, PK>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, PK>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, (long long*)out + (1 << 20), 8, hipMemcpyDeviceToHost);
  double ninstr = (double)iters * 8 * (PK ? ILP : ILP * 2);       // per wave
  printf("%s ILP=%d waves/SIMD=%d : %.2f clk64-ticks/instr/wave, %.3f ms, %.1f TFLOP/s\n", PK ? "pk_fma" : "fma   ", ILP * 2 / (PK ? 2 : 1),
         waves_per_simd, (double)cyc / ninstr, ms, 256.0 * threads * ninstr * (PK ? 4 : 2) / 64 * 64 / (ms * 1e-3) / 1e12 / 64 * 1);
}
int main() {
  float* out; hipMalloc(&out, (1 << 23) + 64);
  for (int w : {1, 2, 4}) { run<1, false>(out, w); run<2, false>(out, w); run<4, false>(out, w); run<8, false>(out, w); }
  for (int w : {1, 2, 4}) { run<1, true>(out, w); run<2, true>(out, w); run<4, true>(out, w); run<8, true>(out, w); }
  return 0;
}
