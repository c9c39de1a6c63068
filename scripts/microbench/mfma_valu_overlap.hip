// Microbenchmark (gfx950): how much VALU work does a SIMD issue in the shadow of its own bf16 MFMAs?
// Per slot: one v_mfma_f32_32x32x16_bf16 (4 independent accumulators round-robin, operands in registers) followed by NV
// independent VALU ops (16 independent chains, so the VALU stream is throughput- not latency-bound).
// Run with one and with two wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu(float& x, float c) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
}

template <int NV, int KIND, int NM, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < NM) acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 4], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NV; ++n) valu<KIND>(v[(m * NV + n) % 16], v[(m * NV + n + 5) % 16]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND, int NM, int OCC>
float run(float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND, NM, OCC>), dim3(256 * OCC), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, NM, OCC>), dim3(256 * OCC), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters / 8;  // ns per slot (per SIMD: OCC waves share it)
}

template <int KIND, int OCC>
void sweep(float* out) {
  const char* kinds[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32"};
  printf("%s, %d wave(s)/SIMD: ns per slot of one wave  [NV: valu only | mfma + valu]\n", kinds[KIND], OCC);
#define ROW(NV) printf("  NV=%2d : %6.2f | %6.2f\n", NV, run<NV, KIND, 0, OCC>(out), run<NV, KIND, 8, OCC>(out))
  ROW(0); ROW(2); ROW(4); ROW(6); ROW(8); ROW(12); ROW(16);
#undef ROW
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  sweep<0, 1>(out);
  sweep<0, 2>(out);
  sweep<1, 1>(out);
  sweep<2, 1>(out);
  return 0;
}
