// Microbenchmark (gfx950), round 3: the SDF kernel's k-step in isolation.  One wavefront per SIMD, four per workgroup.
// Per k-step: six dependent v_mfma_f32_32x32x16_bf16 on ONE accumulator whose A operands were read from LDS one k-step
// earlier (three ds_read_b128, double-buffered), B operands in registers; NV v_fma_f32 fillers (independent chains) after each
// MFMA.  Variants: ACC in AGPRs or VGPRs (compile with / without -mllvm -amdgpu-mfma-vgpr-form=1), B in AGPRs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool LDSA, bool MF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)lds)[i] = 1e-3f * i;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  u32x4 b[3], a[2][3];
  for (int p = 0; p < 3; ++p) { b[p] = u32x4{threadIdx.x + p, 2u, 3u, 4u}; a[0][p] = a[1][p] = b[p]; }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  const char* rd = lds + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        constexpr int X[6] = {2, 0, 1, 1, 0, 0}, Y[6] = {0, 2, 1, 0, 1, 0};
        if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks & 1][X[m]]), __builtin_bit_cast(bf16x8, b[Y[m]]), acc, 0, 0, 0);
        if (LDSA && m < 3) a[(ks + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + ((ks * 3 + m) & 31) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          float& x = v[(m * NV + n) % 16];
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(v[(m * NV + n + 5) % 16]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int j = 0; j < 16; ++j) s += v[j];
  for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, a[0][p][0]) + __builtin_bit_cast(float, a[1][p][1]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool LDSA, bool MF>
float run(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, LDSA, MF>), dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, LDSA, MF>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters / 48;  // ns per MFMA gap
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  printf("ns per MFMA gap: [fillers only | mfma + fillers, A in registers | mfma + fillers, A from LDS]\n");
#define ROW(NV) printf("  NV=%2d : %6.2f | %6.2f | %6.2f\n", NV, run<NV, false, false>(out), run<NV, false, true>(out), run<NV, true, true>(out))
  ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(8);
  return 0;
}
