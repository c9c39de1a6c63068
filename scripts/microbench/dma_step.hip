// Microbenchmark (gfx950): cost of LDS-DMA pieces inside an MFMA-paced, barrier-per-step loop with two wavefronts per SIMD
// (the regime of sdf_mlp_v2.hip).  Per step and wavefront: NDMA pieces (1 KB each, cyclic 1.25 MB L2-resident stream),
// 12 ds_read_b128, 24 v_mfma_f32_32x32x16_bf16 on 4 accumulators, one workgroup barrier (vmcnt(WAITN) + s_barrier).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int NDMA, int PLACE, int DEP>
__global__ __launch_bounds__(WAVES * 64, 1) void k(const char* src, float* out, int steps) {
  __shared__ __attribute__((aligned(16))) char lds[5 * 24 * 1024];
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1310720, 0x00020000);
  const int lane16 = (threadIdx.x & 63) * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  int off = 0;
  for (int s = 0; s < steps; ++s) {
    const int slot = (s % 5) * 24 * 1024, nslot = ((s + 4) % 5) * 24 * 1024;
    if (PLACE == 0) {
#pragma unroll
      for (int d = 0; d < NDMA; ++d)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + nslot + (wave + WAVES * d) * 1024), 16, lane16, off + (wave + WAVES * d) * 1024, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (PLACE == 1 && (wave & 3) == t) {
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + nslot + (wave + WAVES * d) * 1024), 16, lane16, off + (wave + WAVES * d) * 1024, 0, 0);
      }
      u32x4 a[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const u32x4*>(lds + slot + (t * 3 + p) * 1024 + lane16);
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const int q = DEP ? t : (m & 3);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m % 3]), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
      }
    }
    off += 16384;
    if (off >= 1310720 - 32768) off = 0;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA * 3) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) sum += acc[q][i];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = sum;
}

template <int WAVES, int NDMA, int PLACE, int DEP>
void run(const char* name, const char* src, float* out) {
  const int steps = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WAVES, NDMA, PLACE, DEP>), dim3(256), dim3(WAVES * 64), 0, 0, src, out, 2000);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<WAVES, NDMA, PLACE, DEP>), dim3(256), dim3(WAVES * 64), 0, 0, src, out, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %7.2f ms  %6.0f ns/step  (%d MFMA per SIMD-step: %.1f clk @2.4GHz each)\n", name, ms, ms * 1e6 / steps, WAVES / 4 * 24,
         ms * 1e6 / steps * 2.4 / (WAVES / 4 * 24));
}

int main() {
  char* src; float* out;
  hipMalloc(&src, 2 << 20); hipMemset(src, 0x3f, 2 << 20);
  hipMalloc(&out, 256 * 512 * 4);
  run<8, 0, 0, 1>("8 waves, no DMA, dependent chains", src, out);
  run<8, 0, 0, 0>("8 waves, no DMA, 4 independent acc", src, out);
  run<8, 2, 0, 1>("8 waves, 2 DMA at step head", src, out);
  run<8, 2, 1, 1>("8 waves, 2 DMA staggered by tile", src, out);
  run<8, 1, 0, 1>("8 waves, 1 DMA at step head", src, out);
  run<4, 0, 0, 1>("4 waves, no DMA", src, out);
  run<4, 3, 0, 1>("4 waves, 3 DMA at step head", src, out);
  run<4, 3, 1, 1>("4 waves, 3 DMA staggered", src, out);
  return 0;
}
