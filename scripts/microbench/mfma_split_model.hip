// Microbenchmark (gfx950), round 3: the exact-split arithmetic of the SDF kernel (v_cvt_pk_bf16_f32 / v_lshlrev / v_and
// with a literal / v_sub) between the MFMAs of a dependent chain, one instruction kind at a time and as the real sequence.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { K_FMA, K_CVT, K_LSHL, K_ANDLIT, K_ANDREG, K_SUB, K_SEQ, K_SEQ_REGMASK, K_EXPLOG, K_CVTBF, K_CVTBF_SDWA, K_FMAMIX, K_CVTPKF16, K_MED3, K_MUL, K_CNDMASK, K_AND_INPLACE, K_SUB_3OP, K_LSHL_INPLACE, K_FMA_3OP, K_CVT_INPLACE };
template <int KIND>
__device__ __forceinline__ void filler(float (&v)[16], unsigned (&u)[16], int n, unsigned mask) {
  const int i = n % 16, j = (n + 5) % 16;
  if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[j]));
  if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[j]));
  if (KIND == K_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(v[i]) : "v"(u[j]));
  if (KIND == K_ANDLIT) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v[i]) : "v"(u[j]));
  if (KIND == K_ANDREG) asm volatile("v_and_b32 %0, %2, %1" : "=v"(v[i]) : "v"(u[j]), "v"(mask));
  if (KIND == K_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[j]));
  if (KIND == K_AND_INPLACE) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(mask));
  if (KIND == K_LSHL_INPLACE) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
  if (KIND == K_SUB_3OP) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[j]), "v"(v[(n + 9) % 16]));
  if (KIND == K_FMA_3OP) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(v[i]) : "v"(v[j]), "v"(v[(n + 9) % 16]));
  if (KIND == K_CVT_INPLACE) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[j]));
  if (KIND == K_CVTBF) asm volatile("v_cvt_f32_bf16 %0, %1" : "=v"(v[i]) : "v"(u[j]));
  if (KIND == K_CVTBF_SDWA) asm volatile("v_cvt_f32_bf16_sdwa %0, %1 src0_sel:WORD_1" : "=v"(v[i]) : "v"(u[j]));
  if (KIND == K_FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(u[j]));
  if (KIND == K_CVTPKF16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[j]));
  if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(v[i]) : "v"(v[j]));
  if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[j]));
  if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[j]) : "vcc");
  if (KIND == K_SEQ || KIND == K_SEQ_REGMASK) {  // stage n % 3 of the split of pair (2i, 2i+1), as three mini-phases
    const int st = n % 3, p = (n / 3) % 4;
    float &a = v[2 * p], &b = v[2 * p + 1], &xa = v[8 + 2 * p], &xb = v[9 + 2 * p];
    if (st == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[p]) : "v"(a), "v"(b));
    if (st == 1) {
      asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(xa) : "v"(u[p]));
      if (KIND == K_SEQ) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(xb) : "v"(u[p]));
      else asm volatile("v_and_b32 %0, %2, %1" : "=v"(xb) : "v"(u[p]), "v"(mask));
    }
    if (st == 2) {
      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(xa));
      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(b) : "v"(xb));
    }
  }
  if (KIND == K_EXPLOG) {
    if (n & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    else asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));
  }
}

template <int NV, int KIND, bool MF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  u32x4 b[3], a[3];
  for (int p = 0; p < 3; ++p) { b[p] = u32x4{threadIdx.x + p, 2u, 3u, 4u}; a[p] = b[p]; }
  float v[16];
  unsigned u[16];
  for (int j = 0; j < 16; ++j) { v[j] = 1.0f + threadIdx.x * 1e-3f + j; u[j] = j; }
  unsigned mask = 0xffff0000u;
  asm volatile("" : "+v"(mask));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 48; ++g) {
      constexpr int X[6] = {2, 0, 1, 1, 0, 0}, Y[6] = {0, 2, 1, 0, 1, 0};
      if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[X[g % 6]]), __builtin_bit_cast(bf16x8, b[Y[g % 6]]), acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NV; ++n) filler<KIND>(v, u, g * NV + n, mask);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int j = 0; j < 16; ++j) s += v[j] + u[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int KIND, bool MF>
float run(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(256), dim3(256), 0, 0, out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters / 48;
}
template <int KIND>
void sweep(float* out, const char* name) {
  printf("%-28s ns per MFMA gap [fillers only | mfma + fillers]:", name);
#define ROW(NV) printf("  NV=%d %5.2f|%5.2f", NV, run<NV, KIND, false>(out), run<NV, KIND, true>(out))
  ROW(0); ROW(2); ROW(4); ROW(6);
#undef ROW
  printf("\n");
}
int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  sweep<K_AND_INPLACE>(out, "v_and_b32 in place");
  sweep<K_LSHL_INPLACE>(out, "v_lshlrev_b32 in place");
  sweep<K_CVT_INPLACE>(out, "v_cvt_pk_bf16 in place");
  sweep<K_SUB_3OP>(out, "v_sub_f32 dst != src");
  sweep<K_FMA_3OP>(out, "v_fma_f32 dst != src");
  sweep<K_FMA>(out, "v_fma_f32");
  sweep<K_CVT>(out, "v_cvt_pk_bf16_f32");
  sweep<K_LSHL>(out, "v_lshlrev_b32 16");
  sweep<K_ANDLIT>(out, "v_and_b32 literal");
  sweep<K_ANDREG>(out, "v_and_b32 register mask");
  sweep<K_SUB>(out, "v_sub_f32");
  sweep<K_SEQ>(out, "split mini-phases (literal)");
  sweep<K_SEQ_REGMASK>(out, "split mini-phases (reg mask)");
  sweep<K_EXPLOG>(out, "v_exp / v_add alternating");
  sweep<K_CVTBF>(out, "v_cvt_f32_bf16");
  sweep<K_CVTBF_SDWA>(out, "v_cvt_f32_bf16_sdwa WORD_1");
  sweep<K_FMAMIX>(out, "v_fma_mix_f32 (f16 hi)");
  sweep<K_CVTPKF16>(out, "v_cvt_pk_f16_f32");
  sweep<K_MED3>(out, "v_med3_f32");
  sweep<K_MUL>(out, "v_mul_f32");
  sweep<K_CNDMASK>(out, "v_cndmask_b32");
  return 0;
}
