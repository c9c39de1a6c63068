// Shared by two of the fp32 training kernels of the SDF network (sdf_smooth.hip, sdf_bwd.hip; sdf_smooth_bwd.hip keeps the
// round-5 form: with its four streams the per-wavefront operand arrays leave room for ONE 4-wavefront workgroup per CU and the
// shared stream measured slower, 3.5 vs 3.1 ms).
//
// Round 6: the weight stream of a sweep is shared by the wavefronts of a workgroup through LDS.  Until round 5 every wavefront
// (4 samples) read the whole weight image - 0.9 MB for a forward + reverse pair - from L2 by itself: 15,000 wavefronts x 0.9 MB =
// 13.5 GB per launch at 60 k samples (sdf_smooth: 1.47 ms = 9.2 TB/s of L2 traffic).  Measured gain: small (sdf_smooth 1.47 ->
// 1.35 ms, sdf_bwd 1.81 -> 1.69): the loops are bound by the LDS return path of their BROADCAST operands (two 16-byte reads that
// return 2 KB per k-step to the wavefront) and by the FMA issue, both already packed (v_pk_fma_f32) by the compiler - 34 TFLOP/s
// = 0.22 of the packed-FMA / fp32-MFMA rate.  Also measured and not kept (scripts/experiments/sdf_bwd_mfma.hip): the same
// function on v_mfma_f32_32x32x2_f32, 16 samples x 2 streams per wavefront, weights through this same LDS pipeline - correct
// (the parity tests pass) but 1.93 vs 1.69 ms: the fp32 matrix pipe has the packed-FMA rate (MI355X_MICROARCH.md), and the
// softplus algebra of a 32-column tile cannot overlap its MFMAs at one wavefront per SIMD.  Now a workgroup of SURF_TRAIN_WAVES wavefronts (4 samples each, same lane ownership and arithmetic order
// as before: results are bit-identical) walks the rows of a layer's matrix in chunks of CH rows: all 256 threads copy chunk c + 1
// from L2 into registers while every wavefront runs its FMAs on chunk c out of LDS, then park it in the other buffer; one
// workgroup barrier per chunk.  L2 traffic per sample drops by the number of wavefronts per workgroup.
#pragma once
#include "common.h"

#ifndef SURF_TRAIN_WAVES
#define SURF_TRAIN_WAVES 4
#endif
#ifndef SURF_TRAIN_CH
#define SURF_TRAIN_CH 16          // rows per chunk (forward rows: 128 floats, reverse rows: 160 floats)
#endif

namespace surf_train {

// Four samples' accumulators as two packed pairs: acc += w x is two v_pk_fma_f32 (the weight broadcast through op_sel, the
// operand pairs straight out of the 16-byte LDS read) instead of four v_fma_f32.  Round 6: the k / neuron loops of these
// kernels are bound by their FMA issue (34 TFLOP/s = 44 % of the plain-FMA rate at 2.1 GHz, rocprofv3: VALU busy), not by the
// weight stream as assumed in round 5; hipcc's SLP vectoriser had packed a quarter of them.  Same products, same order of
// accumulation per sample: bit-identical results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct V4 {
  f32x2 a, b;
  __device__ __forceinline__ void zero() { a = f32x2{0.f, 0.f}; b = a; }
  __device__ __forceinline__ float operator[](int s) const { return s < 2 ? a[s] : b[s - 2]; }
  __device__ __forceinline__ void set(int s, float v) { if (s < 2) a[s] = v; else b[s - 2] = v; }
  __device__ __forceinline__ void fma(float w, const f32x4& x) {
    const f32x2 wv = {w, w};
    a = __builtin_elementwise_fma(wv, f32x2{x[0], x[1]}, a);
    b = __builtin_elementwise_fma(wv, f32x2{x[2], x[3]}, b);
  }
};

constexpr int NW = SURF_TRAIN_WAVES, NT = 64 * NW, CH = SURF_TRAIN_CH;
constexpr int WBUF_FLOATS = 2 * CH * 160;      // two chunk buffers of the longer (reverse) rows

// body(row index, pointer to the row's ROW floats in LDS) for rows 0 .. rows - 1 of the row-major matrix at src (16-byte aligned,
// ROW a multiple of 4).  Every thread of the workgroup must call it with the same arguments; the caller's own barrier after the
// sweep (every kernel has one: the LDS operand arrays are rewritten next) also frees the buffers for the next sweep.
template <int ROW, typename Body>
__device__ __forceinline__ void stream_rows(const float* __restrict__ src, int rows, float* __restrict__ wbuf, Body&& body) {
  constexpr int N4 = CH * ROW / 4, PER = (N4 + NT - 1) / NT;
  static_assert(ROW % 4 == 0 && 2 * CH * ROW <= WBUF_FLOATS, "chunk buffers");
  f32x4 r[PER];
  const int tid = threadIdx.x;
  const int nch = (rows + CH - 1) / CH;
  auto gload = [&](int c) {
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src + (int64_t)c * CH * ROW);
    const int lim = (min(CH, rows - c * CH) * ROW) / 4;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + NT * u;
      r[u] = idx < lim ? s4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int b) {
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(wbuf + b * CH * ROW);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + NT * u;
      if (idx < N4) d4[idx] = r[u];
    }
  };
  gload(0);
  lstore(0);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();                               // chunk c is in buffer c & 1; the other buffer is free
    if (c + 1 < nch) gload(c + 1);
    const float* __restrict__ wl = wbuf + (c & 1) * CH * ROW;
    const int nr = min(CH, rows - c * CH);
#pragma unroll 4
    for (int kk = 0; kk < nr; ++kk) body(c * CH + kk, wl + kk * ROW);
    if (c + 1 < nch) lstore((c + 1) & 1);
  }
}

}  // namespace surf_train
