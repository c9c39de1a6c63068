// Microbenchmark (gfx950), round 4: does a SIMD overlap one wavefront's VALU work with ANOTHER wavefront's MFMAs?
// The blend kernel's per-view chain alternates a VALU block (ELU + operand split of a tile, ~120 instructions) with a block
// of 12 dependent MFMAs, and relies on the second wavefront of the SIMD being in the other phase.  Its SQ counters say the two
// hardly overlap (VALU issue 77 % + matrix pipe 33 % = 110 % of SIMD time).  This program times the same shape three ways:
//   A  two wavefronts per SIMD, one chain each                       (what blend_split does)
//   B  ONE wavefront per SIMD carrying TWO independent chains, software-pipelined by half a stage in the SOURCE: the MFMAs of
//      chain 0 and the VALU block of chain 1 stand in one scheduling region, interleaved 1 MFMA : NV VALU with
//      __builtin_amdgcn_sched_group_barrier
//   C  two wavefronts per SIMD, two chains each
// ns per tile (one VALU block + 12 MFMAs), per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

struct Frags { u32x4 p[3][2]; };
// the VALU block: "ELU" (4 ops) + exact three-way split (11 ops a pair) of a tile
__device__ __forceinline__ void valu_block(const f32x16& acc, Frags& f) {
#pragma unroll
  for (int pr = 0; pr < 8; ++pr) {
    float x = acc[2 * pr], y = acc[2 * pr + 1];
    x = fmaf(__builtin_amdgcn_fmed3f(x, 0.f, 3e38f), 0.69f, __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.f, 1.f) - 1.0f);
    y = fmaf(__builtin_amdgcn_fmed3f(y, 0.f, 3e38f), 0.69f, __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y), 0.f, 1.f) - 1.0f);
    const uint32_t p0 = pack2(x, y);
    const float rx = x - lo(p0), ry = y - hi(p0);
    const uint32_t p1 = pack2(rx, ry);
    const uint32_t p2 = pack2(rx - lo(p1), ry - hi(p1));
    f.p[0][pr >> 2][pr & 3] = p0; f.p[1][pr >> 2][pr & 3] = p1; f.p[2][pr >> 2][pr & 3] = p2;
  }
}
template <int NOP = 0>
__device__ __forceinline__ void mfma_block(f32x16& acc, const Frags& f, const u32x4& w) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // NOP > 0: the wavefront steps back from the issue port for NOP cycles after every MFMA (s_nop is not a VALU instruction), so
    // that its NEXT MFMA does not sit at the port waiting for the matrix pipe while the other wavefront has VALU work
#define MF(pc)                                                                                                                              \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f.p[pc][s]), acc, 0, 0, 0); \
  if (NOP >= 8) __builtin_amdgcn_sched_barrier(0);                                                                                          \
  if (NOP >= 8) asm volatile("s_nop 7");                                                                                                  \
  if (NOP >= 16) asm volatile("s_nop 7");                                                                                                 \
  if (NOP >= 24) asm volatile("s_nop 7");                                                                                                 \
  if (NOP >= 8) __builtin_amdgcn_sched_barrier(0);
    MF(2) MF(2) MF(1) MF(1) MF(0) MF(0)
#undef MF
  }
}

// one pair of the VALU block
__device__ __forceinline__ void valu_pair(const f32x16& acc, Frags& f, int pr) {
  float x = acc[2 * pr], y = acc[2 * pr + 1];
  x = fmaf(__builtin_amdgcn_fmed3f(x, 0.f, 3e38f), 0.69f, __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.f, 1.f) - 1.0f);
  y = fmaf(__builtin_amdgcn_fmed3f(y, 0.f, 3e38f), 0.69f, __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y), 0.f, 1.f) - 1.0f);
  const uint32_t p0 = pack2(x, y);
  const float rx = x - lo(p0), ry = y - hi(p0);
  const uint32_t p1 = pack2(rx, ry);
  const uint32_t p2 = pack2(rx - lo(p1), ry - hi(p1));
  f.p[0][pr >> 2][pr & 3] = p0; f.p[1][pr >> 2][pr & 3] = p1; f.p[2][pr >> 2][pr & 3] = p2;
}
// MFMAs of `am` (chain m) with the VALU block of `av` (chain v) dealt out between them by hand: one scheduling barrier per gap,
// the accumulator pinned in its gap (the instruction selector would otherwise sink the MFMAs; compile with -mllvm -pre-RA-sched=source)
__device__ __forceinline__ void woven(f32x16& am, const Frags& fm, const u32x4& w, const f32x16& av, Frags& fv) {
#pragma unroll
  for (int g = 0; g < 12; ++g) {
    const int s = g / 6, q = g % 6, pc = q < 2 ? 2 : q < 4 ? 1 : 0;
    am = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, fm.p[pc][s]), am, 0, 0, 0);
    asm volatile("" : "+v"(am));
    __builtin_amdgcn_sched_barrier(0);
    if (g % 3 != 2) valu_pair(av, fv, (g / 3) * 2 + g % 3);   // 8 pairs over 12 gaps
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int STREAMS, int WAVES, int NV>
__global__ __launch_bounds__(256 * WAVES, 1) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[STREAMS];
  Frags fr[STREAMS];
  for (int s = 0; s < STREAMS; ++s)
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.001f * (lane + r + s);
  const u32x4 w = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // small weights: the chain stays finite
  if (STREAMS == 1) {
    for (int it = 0; it < iters; ++it) {
      // NV = 1: VALU phase at high priority; NV = 2: MFMA phase at high priority (s_setprio), two waves / SIMD
      if (NV == 1) __builtin_amdgcn_s_setprio(3);
      if (NV == 2) __builtin_amdgcn_s_setprio(0);
      valu_block(acc[0], fr[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (NV == 1) __builtin_amdgcn_s_setprio(0);
      if (NV == 2) __builtin_amdgcn_s_setprio(3);
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.01f * r;
      if (NV == 8) mfma_block<8>(acc[0], fr[0], w);
      else if (NV == 16) mfma_block<16>(acc[0], fr[0], w);
      else if (NV == 24) mfma_block<24>(acc[0], fr[0], w);
      else mfma_block(acc[0], fr[0], w);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (NV == 0) {
    valu_block(acc[0], fr[0]);
    for (int it = 0; it < iters; ++it) {
      f32x16 n0, n1;
      for (int r = 0; r < 16; ++r) { n0[r] = 0.01f * r; n1[r] = 0.01f * r; }
      woven(n0, fr[0], w, acc[1], fr[1]);
      acc[0] = n0;
      woven(n1, fr[1], w, acc[0], fr[0]);
      acc[1] = n1;
    }
  } else {
    valu_block(acc[0], fr[0]);
    for (int it = 0; it < iters; ++it) {
      // region 1: MFMAs of stream 0 | VALU block of stream 1
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.01f * r;
      mfma_block(acc[0], fr[0], w);
      valu_block(acc[1], fr[1]);
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // region 2: MFMAs of stream 1 | VALU block of stream 0
      for (int r = 0; r < 16; ++r) acc[1][r] = 0.01f * r;
      mfma_block(acc[1], fr[1], w);
      valu_block(acc[0], fr[0]);
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < STREAMS; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
}

template <int STREAMS, int WAVES, int NV>
void run(float* out, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<STREAMS, WAVES, NV>), dim3(256), dim3(256 * WAVES), 0, 0, out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<STREAMS, WAVES, NV>), dim3(256), dim3(256 * WAVES), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double tiles_per_simd = (double)iters * WAVES * (STREAMS == 1 ? 1 : 2);
  printf("%-58s %7.1f ns per tile per SIMD\n", name, ms * 1e6 / tiles_per_simd);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<1, 1, 0>(out, "one wave / SIMD, one chain (no overlap possible)");
  run<1, 2, 0>(out, "A  two waves / SIMD, one chain each");
  run<1, 2, 1>(out, "A  two waves / SIMD, VALU phase at s_setprio 3");
  run<1, 2, 2>(out, "A  two waves / SIMD, MFMA phase at s_setprio 3");
  run<1, 2, 8>(out, "A  two waves / SIMD, s_nop 8 cycles after every MFMA");
  run<1, 2, 16>(out, "A  two waves / SIMD, s_nop 16 cycles after every MFMA");
  run<1, 2, 24>(out, "A  two waves / SIMD, s_nop 24 cycles after every MFMA");
  run<1, 3, 16>(out, "A3 three waves / SIMD, s_nop 16 cycles after every MFMA");
  run<1, 3, 0>(out, "A3 three waves / SIMD, one chain each");
  run<1, 4, 0>(out, "A4 four waves / SIMD, one chain each");
  run<2, 1, 8>(out, "B  one wave / SIMD, two chains, 1 MFMA : 8 VALU");
  run<2, 1, 10>(out, "B  one wave / SIMD, two chains, 1 MFMA : 10 VALU");
  run<2, 1, 12>(out, "B  one wave / SIMD, two chains, 1 MFMA : 12 VALU");
  run<2, 2, 10>(out, "C  two waves / SIMD, two chains each, 1 MFMA : 10 VALU");
  run<2, 1, 0>(out, "D  one wave / SIMD, two chains, woven by hand");
  run<2, 2, 0>(out, "E  two waves / SIMD, two chains each, woven by hand");
  return 0;
}
