// Microbenchmark (gfx950), round 5: does ONE dependent accumulator chain of v_mfma_f32_32x32x16_bf16 run at 32 cycles per MFMA, or
// does the SrcC dependency add a bubble that a second, independent chain (alternating accumulators) would hide?
// Question behind it: the split SDF kernel's sweeps run at 37-38 shader clocks per 32-cycle MFMA, and taking 35 % of the VALU
// work out of the gaps (matrix-pipe residuals, round 5) did NOT make them faster - so the gaps are not VALU-issue bound.
// One wavefront per SIMD (256 threads, 1 block per CU), the SDF k-step's shape: six MFMAs per k-step, A pieces from LDS one k-step
// ahead, NV independent v_fma fillers per gap.  NA = number of accumulator chains the six MFMAs of a k-step alternate over.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/microbench/mfma_dep_chain.hip -o /tmp/mfma_dep_chain && /tmp/mfma_dep_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[2];
__device__ __forceinline__ void pin(const f32x16& a) { asm volatile("" ::"v"(a)); }
__device__ __forceinline__ void filler(float& x, float y) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y)); }

template <int NV, int NA, bool LDSA>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)lds)[i] = 1e-3f * i;
  __syncthreads();
  f32x16 acc[NA];
  for (int q = 0; q < NA; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  u32x4 b[3], a[2][3];
  for (int p = 0; p < 3; ++p) { b[p] = u32x4{threadIdx.x + p, 2u, 3u, 4u}; a[0][p] = a[1][p] = b[p]; }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  const char* rd = lds + (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        constexpr int X[6] = {2, 0, 1, 1, 0, 0}, Y[6] = {0, 2, 1, 0, 1, 0};
        acc[m % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks & 1][X[m]]), __builtin_bit_cast(bf16x8, b[Y[m]]), acc[m % NA], 0, 0, 0);
        pin(acc[m % NA]);
        if (LDSA && m < 3) a[(ks + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + ((ks * 3 + m) & 31) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          filler(v[(m * NV + n) % 16], v[(m * NV + n + 5) % 16]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) { g_clk[0] = t1 - t0; }
  float s = 0;
  for (int q = 0; q < NA; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  for (int p = 0; p < 3; ++p) s += __builtin_bit_cast(float, a[0][p][0]) + __builtin_bit_cast(float, a[1][p][1]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NA, bool LDSA>
void run(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NA, LDSA>), dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NA, LDSA>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long clk[2];
  hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
  printf("NV %d  NA %d  LDS-A %d : %.2f ns per MFMA gap, %.2f counter ticks per gap (s_memtime, 100 MHz)\n", NV, NA, (int)LDSA, ms * 1e6f / iters / 48,
         (double)clk[0] / iters / 48);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  run<0, 1, false>(out); run<0, 2, false>(out); run<0, 3, false>(out);
  run<0, 1, true>(out); run<0, 2, true>(out);
  run<2, 1, true>(out); run<2, 2, true>(out);
  run<4, 1, true>(out); run<4, 2, true>(out);
  run<6, 1, true>(out); run<6, 2, true>(out);
  return 0;
}
