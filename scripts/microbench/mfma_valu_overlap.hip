// Microbenchmark: how does one wavefront per SIMD overlap its own VALU work with its own bf16 MFMAs on gfx950?
// Per iteration: 6 MFMAs (32x32x16 bf16) over NACC independent accumulators, each followed by NV VALU ops of a kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu(float& x, float c) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x) : "v"(*(double*)&c));
  if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
}

template <int NV, int KIND, int NACC, int NM>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      if (m < NM) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NV; ++n) valu<KIND>(v[(2 * n) % 16], v[15]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND, int NACC, int NM>
void run(float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND, NACC, NM>), dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, NACC, NM>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const char* kinds[] = {"fma", "exp", "pk_mul", "cvt_pk_bf16"};
  printf("mfma x%d (acc %d) + %2d %-11s per slot: %6.1f clk/slot @2.4GHz\n", NM, NACC, NV, kinds[KIND], ms * 1e6 / iters * 2.4 / 6);
}

template <int KIND, int NACC, int NM>
void sweep(float* out) {
  run<0, KIND, NACC, NM>(out); run<2, KIND, NACC, NM>(out); run<4, KIND, NACC, NM>(out); run<6, KIND, NACC, NM>(out);
  run<8, KIND, NACC, NM>(out); run<12, KIND, NACC, NM>(out); run<16, KIND, NACC, NM>(out);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  sweep<0, 1, 0>(out);  // VALU alone
  sweep<0, 1, 6>(out);
  sweep<0, 2, 6>(out);
  sweep<0, 3, 6>(out);
  sweep<1, 1, 0>(out);
  sweep<1, 2, 6>(out);
  sweep<2, 2, 6>(out);
  sweep<3, 2, 6>(out);
  return 0;
}
