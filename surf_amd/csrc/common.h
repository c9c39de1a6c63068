// Shared device helpers for the SuRF hot-path kernels (gfx950 / wave64 only).
// The whole library is compiled with -ffp-contract=off: the reference computes coordinates with
// separate fp32 multiplies and adds, and floor()/rint()/comparisons on them decide which voxel or
// texel a sample lands in.  FMAs are written explicitly (fmaf) where they are wanted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/surf_hip.h"

#define SURF_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- exact three-way bf16 split of a pair of fp32 values (a = a1 + a2 + a3, 8 + 8 + 8 significant bits, RNE residuals) ----
// used by every kernel that runs fp32 contractions on the bf16 matrix pipe (sdf_mlp_split, blend_split, spconv_mfma).
// p[k] packs piece k of (a, b) as (lo, hi) halves; the residual a - float(piece) is a shift / mask + a subtraction (11 VALU
// per pair in all).  Tried in round 3 and NOT used (SURF_SPLIT_DOT2 = 1): the residual as ONE v_dot2c_f32_bf16
// (acc + piece.lo * -1 + piece.hi * 0; 7 VALU per pair).  Two findings (scripts/microbench/dot2_residual.hip, same-box A/B of
// bench.py): the instruction is not full rate - blend 43.1 vs 41.6 ms, SDF 129.9 vs 128.9 ms, 53 vs 45 ns per dependent
// split - and hipcc encodes the packed constant (-1, 0) as the inline constant -1.0, which the hardware expands to
// 0xBF800000 = (0, -1): the constant must be kept out of the inline-constant path (an opaque register).
#ifndef SURF_SPLIT_DOT2
#define SURF_SPLIT_DOT2 0
#endif
typedef __bf16 surf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t surf_pack2_bf16(float a, float b) {
  surf_bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  uint32_t u = __builtin_bit_cast(uint32_t, v);
#ifndef SURF_PACK_PIN
#define SURF_PACK_PIN 1
#endif
#if SURF_PACK_PIN
  asm volatile("" : "+v"(u));  // keep the packed value: the residuals below come from its two halves
#endif
  return u;
}
__device__ __forceinline__ void surf_residual_bf16(uint32_t packed, float a, float b, float& ra, float& rb) {
#if SURF_SPLIT_DOT2
  uint32_t lo = 0x0000bf80u, hi = 0xbf800000u;          // (-1, 0) and (0, -1) as packed bf16 pairs
  asm volatile("" : "+v"(lo), "+v"(hi));                // not an inline constant: see above
  ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(surf_bf16x2, packed), __builtin_bit_cast(surf_bf16x2, lo), a, false);
  rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(surf_bf16x2, packed), __builtin_bit_cast(surf_bf16x2, hi), b, false);
#else
  ra = a - __builtin_bit_cast(float, packed << 16);
  rb = b - __builtin_bit_cast(float, packed & 0xffff0000u);
#endif
}
__device__ __forceinline__ void surf_split3_bf16(float a, float b, uint32_t (&p)[3]) {
  float ra, rb, ra2, rb2;
  p[0] = surf_pack2_bf16(a, b);
  surf_residual_bf16(p[0], a, b, ra, rb);
  p[1] = surf_pack2_bf16(ra, rb);
  surf_residual_bf16(p[1], ra, rb, ra2, rb2);
  p[2] = surf_pack2_bf16(ra2, rb2);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

// The three z-neighbours (bz - 1, bz, bz + 1) of a column (x, y) of a z-fastest D^3 index table as ONE 12-byte load (round 6: the
// sparse-convolution kernels' submanifold and stride-2 windows walk 9 such columns instead of 27 single entries; the texture path
// serves distinct cache lines one at a time, and the table was half of those kernels' line requests).  The caller guarantees
// bz - 1 >= 0 and bz + 1 < D; a column outside the lattice yields (-1, -1, -1).
#ifndef SURF_SPCONV_TRIPLE
#define SURF_SPCONV_TRIPLE 1
#endif
struct __attribute__((packed, aligned(4))) I3u { int a, b, c; };      // a 12-byte load from a 4-byte aligned address
__device__ __forceinline__ I3u surf_table_column3(const int32_t* __restrict__ table, int D, int x, int y, int bz) {
  const bool ok = x >= 0 && x < D && y >= 0 && y < D;
  const I3u t = *reinterpret_cast<const I3u*>(table + (ok ? ((int64_t)x * D + y) * D + (bz - 1) : 0));
  return ok ? t : I3u{-1, -1, -1};
}

// grid_sample's normalised->index rule for align_corners=False applied to a world coordinate in
// [-1,1]: ((p + 1) * D - 1) / 2   (projector.py:406,415 use the default align_corners=False).
__device__ __forceinline__ float unnorm_acf(float p, int D) { return ((p + 1.0f) * (float)D - 1.0f) / 2.0f; }

// Trilinear fetch with zero padding from a dense [x][y][z] volume (F.grid_sample, zeros).
__device__ __forceinline__ float trilinear_zeros(const float* __restrict__ vol, int D, float gx, float gy, float gz) {
  float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  float tx = gx - fx, ty = gy - fy, tz = gz - fz;
  int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  float acc = 0.0f;
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) {
    int xi = x0 + dx;
    float wx = dx ? tx : 1.0f - tx;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      int yi = y0 + dy;
      float wy = dy ? ty : 1.0f - ty;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz) {
        int zi = z0 + dz;
        float wz = dz ? tz : 1.0f - tz;
        bool ok = (xi >= 0) & (xi < D) & (yi >= 0) & (yi < D) & (zi >= 0) & (zi < D);
        float v = 0.0f;
        if (ok) v = vol[((int64_t)xi * D + yi) * D + zi];
        acc += v * (wx * wy * wz);
      }
    }
  }
  return acc;
}

// Nearest lookup (nearbyint = half-to-even, zeros outside) of "is this voxel occupied" through the
// int32 index table: the reference's mask volume is 1 exactly where the table is >= 0
// (volume.py:112-116 vs :125-130).
__device__ __forceinline__ bool occupied_nearest(const int32_t* __restrict__ table, int D, float px, float py, float pz) {
  int xi = (int)rintf(unnorm_acf(px, D));
  int yi = (int)rintf(unnorm_acf(py, D));
  int zi = (int)rintf(unnorm_acf(pz, D));
  bool ok = (xi >= 0) & (xi < D) & (yi >= 0) & (yi < D) & (zi >= 0) & (zi < D);
  if (!ok) return false;
  return table[((int64_t)xi * D + yi) * D + zi] >= 0;
}

// Bilinear fetch of a texel4 map (H,W,4) with zero padding per tap.
__device__ __forceinline__ f32x4 bilinear_texel4(const float* __restrict__ map, int H, int W, float x, float y) {
  float fx = floorf(x), fy = floorf(y);
  float tx = x - fx, ty = y - fy;
  int x0 = (int)fx, y0 = (int)fy;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    int yi = y0 + dy;
    float wy = dy ? ty : 1.0f - ty;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      int xi = x0 + dx;
      float wx = dx ? tx : 1.0f - tx;
      bool ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
      if (ok) {
        f32x4 v = *reinterpret_cast<const f32x4*>(map + ((int64_t)yi * W + xi) * 4);
        float w = wx * wy;
        acc += v * w;
      }
    }
  }
  return acc;
}

static inline int surf_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
