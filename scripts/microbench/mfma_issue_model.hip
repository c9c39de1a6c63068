// Microbenchmark (gfx950), round 3: what does ONE wavefront per SIMD pay for instructions placed between the MFMAs of a
// DEPENDENT accumulator chain (the SDF kernel's situation: all six products of a k-step and all k-steps of a tile go into one
// accumulator)?  Per slot: one v_mfma_f32_32x32x16_bf16 (CH accumulator chains round-robin) followed by NV fillers of one kind.
// Prints ns per slot; the MFMA-only row gives the length of 32 matrix-pipe cycles at the clock the chip runs this at.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FMA, EXP, CVT, NOP, SMOV, DSR, DEPFMA, PKADD };
template <int KIND>
__device__ __forceinline__ void filler(float (&v)[16], int n, const char* lds, f32x4& sink) {
  float& x = v[n % 16];
  const float c = v[(n + 5) % 16];
  if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
  if (KIND == NOP) asm volatile("s_nop 0");
  if (KIND == SMOV) { int s; asm volatile("s_mov_b32 %0, 0x1234" : "=s"(s)); }
  if (KIND == DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"((int)(size_t)lds));
  if (KIND == DEPFMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(c));  // one serial chain
  if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[2 * (n % 8)]) : "v"(*(double*)&v[2 * ((n + 3) % 8)]));
}

template <int NV, int KIND, int NM, int CH>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  __shared__ char lds[4096];
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  f32x4 sink = {0, 0, 0, 0};
  const char* lp = lds + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < NM) acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CH], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NV; ++n) filler<KIND>(v, m * NV + n, lp, sink);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == DSR) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = sink[0];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND, int NM, int CH>
float run(float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND, NM, CH>), dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, NM, CH>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters / 8;
}

template <int KIND, int CH>
void sweep(float* out, const char* name) {
  printf("%-18s %d accumulator chain(s): ns per slot  [fillers only | mfma + fillers]\n", name, CH);
#define ROW(NV) printf("  NV=%2d : %6.2f | %6.2f\n", NV, run<NV, KIND, 0, CH>(out), run<NV, KIND, 8, CH>(out))
  ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(8); ROW(12);
#undef ROW
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  sweep<FMA, 1>(out, "v_fma_f32");
  sweep<FMA, 4>(out, "v_fma_f32");
  sweep<DEPFMA, 1>(out, "v_fma_f32 serial");
  sweep<EXP, 1>(out, "v_exp_f32");
  sweep<CVT, 1>(out, "v_cvt_pk_bf16_f32");
  sweep<PKADD, 1>(out, "v_pk_add_f32");
  sweep<NOP, 1>(out, "s_nop 0");
  sweep<SMOV, 1>(out, "s_mov_b32");
  sweep<DSR, 1>(out, "ds_read_b128");
  return 0;
}
