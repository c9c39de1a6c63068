// K5m: the wide layers (C_in, C_out >= 16) of the sparse cost-regularisation U-Net as per-offset gather-GEMMs on the
// matrix cores.  Same function as spconv_kernel (spconv.hip; SparseCostRegNet, reg_network.py:38-88: 3^3 submanifold /
// stride-2 / transposed stride-2 sparse convolution + BatchNorm(eval) + ReLU (+ skip)); same PARITY-UNPINNED caveat.
//
// Output-stationary: one wavefront owns 32 output voxels and all C_out channels.  For each of the 27 kernel offsets
//   D[c_out][voxel] += W_k^T[c_out][c_in] * X[c_in][voxel]
// with X gathered through the dense int32 index table (missing neighbours = zero columns; a wavefront whose 32 voxels all
// miss an offset skips its MFMAs) and W_k staged in LDS for the four wavefronts of the workgroup.  The arithmetic is the
// bf16x3 scheme of the SDF / blend kernels: both operands split exactly into three bf16 pieces, six
// v_mfma_f32_32x32x16_bf16 products accumulated in fp32 - fp32-equivalent results on the bf16 pipe (the fp32 MFMA would
// run at 1/16 of its rate).  The weights are split once per model (surf_spconv_pack_weights, a device kernel).
//
// MFMA operand layout (32x32x16): A lane l: row l%32, k = 8(l/32)..+8;  B lane l: column l%32, k = 8(l/32)..+8;
// D register r of lane l: row (r&3) + 8(r>>2) + 4(l/32), column l%32 - so a lane ends with 4 x 4 consecutive output
// channels of its own voxel and stores them with 16-byte writes.
#include <stdlib.h>

#include "common.h"

namespace {

enum { MODE_SUBM = 0, MODE_DOWN = 1, MODE_UP = 2 };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct SpArgs {
  const float* in;
  const int32_t* in_table;
  int Din;
  const int32_t* out_coords;
  int64_t n_out;
  int mode;
  const u32x4* packed;  // [27][KS][MT][3][64] 16-byte fragments
  const float* scale;
  const float* shift;
  const float* skip;
  float* out;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  asm volatile("" : "+v"(u));  // keep the packed value: the residuals come from its two halves
  return u;
}
// exact three-way split of a pair: a = p0.lo + p1.lo + p2.lo (+ < 2^-24 |a|), same for b in the high halves
__device__ __forceinline__ void split3(float a, float b, uint32_t (&p)[3]) { surf_split3_bf16(a, b, p); }

template <int CIN, int COUT>
struct Shape {
  static constexpr int KS = (CIN + 15) / 16, MT = (COUT + 31) / 32;   // C_in = 8: one k-step whose upper half (lanes 32..63) is zero
  static constexpr int FRAGS = KS * MT * 3;       // 1 KB fragments per kernel offset
};

// weight (27, CIN, COUT) fp32 -> packed[k][ks][mt][piece][lane] (one thread per (k, ks, mt, lane))
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ out) {
  typedef Shape<CIN, COUT> S;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 27 * S::KS * S::MT * 64) return;
  const int lane = t & 63, mt = (t >> 6) % S::MT, ks = (t >> 6) / S::MT % S::KS, k = (t >> 6) / (S::MT * S::KS);
  const int co = 32 * mt + (lane & 31), ci0 = 16 * ks + 8 * (lane >> 5);
  u32x4 p[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool in = co < COUT && ci0 + 2 * j + 1 < CIN;
    const float a = in ? w[((int64_t)k * CIN + ci0 + 2 * j) * COUT + co] : 0.f;
    const float b = in ? w[((int64_t)k * CIN + ci0 + 2 * j + 1) * COUT + co] : 0.f;
    uint32_t q[3];
    split3(a, b, q);
    p[0][j] = q[0]; p[1][j] = q[1]; p[2][j] = q[2];
  }
  const int64_t base = ((int64_t)(k * S::KS + ks) * S::MT + mt) * 3 * 64 + lane;
#pragma unroll
  for (int q = 0; q < 3; ++q) out[base + q * 64] = p[q];
}

// NPROD = 6: the exact three-way split, fp32-equivalent (inference, and training under train_precision = fp32).
// NPROD = 1: BASELINE configs[3]'s bf16 policy (train_precision = bf16) - both operands rounded to bf16 (the first piece of the
// same weight image, the first piece of the gathered rows), ONE product per k-step, fp32 accumulation: what autocast(bf16) gives a
// convolution.  Used for the wide layers' forward and input gradients of a training step only.
template <int CIN, int COUT, int NPROD>
__global__ __launch_bounds__(256) void spconv_mfma_kernel(SpArgs a) {
  typedef Shape<CIN, COUT> S;
  constexpr int KS = S::KS, MT = S::MT, FRAGS = S::FRAGS;
  __shared__ u32x4 wlds[2][FRAGS * 64];                       // double-buffered W_k pieces (<= 2 x 24 KB)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool live = i < a.n_out;
  const int64_t ic = live ? i : a.n_out - 1;
  const int cx = a.out_coords[ic * 3 + 0], cy = a.out_coords[ic * 3 + 1], cz = a.out_coords[ic * 3 + 2];
  const int D = a.Din;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  auto stage = [&](int k, int buf) {                          // this thread's share of W_k -> LDS
    const u32x4* __restrict__ src = a.packed + (int64_t)k * FRAGS * 64;
#pragma unroll
    for (int u = threadIdx.x; u < FRAGS * 64; u += 256) wlds[buf][u] = src[u];
  };
  auto neighbour = [&](int k) -> int {                        // input row feeding this voxel through offset k, or -1
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    int x, y, z;
    bool ok = live;
    if (a.mode == MODE_SUBM) {
      x = cx + ox; y = cy + oy; z = cz + oz;
    } else if (a.mode == MODE_DOWN) {
      x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz;
    } else {                                                  // MODE_UP: coarse site q with 2 q + o == c
      const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
      ok = ok && ((tx | ty | tz) & 1) == 0;
      x = tx >> 1; y = ty >> 1; z = tz >> 1;
    }
    ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
    return ok ? a.in_table[((int64_t)x * D + y) * D + z] : -1;
  };

  stage(0, 0);
  int row = neighbour(0);
  for (int k = 0; k < 27; ++k) {
    // gather this offset's input columns (two 16-byte loads per 16-channel k-step), look the next offset's row up
    f32x4 xv[KS][2];
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.in + (int64_t)(row < 0 ? 0 : row) * CIN) + 2 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (row >= 0 && 16 * ks + 8 * h < CIN) { xv[ks][0] = src[4 * ks]; xv[ks][1] = src[4 * ks + 1]; }
      else { xv[ks][0] = f32x4{0.f, 0.f, 0.f, 0.f}; xv[ks][1] = xv[ks][0]; }
    }
    const bool any = __ballot(row >= 0) != 0ull;
    const int next_row = k + 1 < 27 ? neighbour(k + 1) : -1;
    __syncthreads();                                          // W_k landed in wlds[k & 1]; buffer (k + 1) & 1 is free
    if (k + 1 < 27) stage(k + 1, (k + 1) & 1);
    if (any) {
      const u32x4* __restrict__ wl = wlds[k & 1] + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 b[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (NPROD == 1) {
            b[0][j] = pack2(xv[ks][j >> 1][2 * (j & 1)], xv[ks][j >> 1][2 * (j & 1) + 1]);
          } else {
            uint32_t q[3];
            split3(xv[ks][j >> 1][2 * (j & 1)], xv[ks][j >> 1][2 * (j & 1) + 1], q);
            b[0][j] = q[0]; b[1][j] = q[1]; b[2][j] = q[2];
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const u32x4* __restrict__ wf = wl + (ks * MT + m) * 3 * 64;
#define SURF_MF(x, y) \
  acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[m], 0, 0, 0)
          if constexpr (NPROD == 1) {
            const u32x4 a0 = wf[0];
            SURF_MF(a0, b[0]);
          } else {
            const u32x4 a0 = wf[0], a1 = wf[64], a2 = wf[128];
            SURF_MF(a2, b[0]);  // smallest terms first
            SURF_MF(a0, b[2]);
            SURF_MF(a1, b[1]);
            SURF_MF(a1, b[0]);
            SURF_MF(a0, b[1]);
            SURF_MF(a0, b[0]);
          }
#undef SURF_MF
        }
      }
    }
    row = next_row;
  }
  if (!live) return;
  // epilogue: BatchNorm(eval) + ReLU (+ skip); register 4 g + q of tile m = channel 32 m + 8 g + 4 h + q
  float* __restrict__ dst = a.out + i * COUT;
  const float* __restrict__ sk = a.skip ? a.skip + i * COUT : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = 32 * m + 8 * g + 4 * h;
      if (co >= COUT) continue;
      f32x4 v = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
      if (a.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + co), sh = *reinterpret_cast<const f32x4*>(a.shift + co);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q] * sc[q] + sh[q], 0.f);
      }
      if (sk) v += *reinterpret_cast<const f32x4*>(sk + co);
      *reinterpret_cast<f32x4*>(dst + co) = v;
    }
  }
}

// ---- thin pairs (C_in, C_out <= 16), round 6 -----------------------------------------------------------------------------------
// The kernel above walks the 27 offsets as a chain of dependent round trips (table entry -> row -> MFMAs) with one workgroup
// barrier and one weight-slice stage per offset: fine for the wide layers, whose six-product k-steps hide it, but slower than the
// per-voxel FMA kernel for the thin pairs of the finest lattices (measured: <8,16> 2.28 vs 1.63 ms per training step).  The thin
// pairs get the structure of spconv_pipe_kernel (spconv.hip) around the matrix cores instead:
//   * the whole weight image of a thin pair (27 offsets x 1 or 3 one-KB fragments) is staged into LDS ONCE per workgroup, which
//     then walks its share of the 256-site tiles - no barrier inside the offset loop;
//   * all 27 table entries of a site are fetched first (27 independent loads), the row of offset k + 1 is in flight while offset
//     k is converted and multiplied;
//   * a wavefront whose 32 sites all miss an offset skips its conversions and MFMAs.
// Per site the matrix pipe needs 27 x 32 / 32 = 27 cycles for one bf16 product (162 for the exact six) against 216 cycles of
// FMA issue in the per-voxel kernel - but neither form is bound by its arithmetic: spconv_dgrad<8,16> 1.63 ms per training step
// on the FMA kernel, 1.88 here with one product, 2.70 with six (profiles/r06_train_experiments.txt).
template <int CIN, int COUT, int MODE, int NPROD>
__global__ __launch_bounds__(512) void spconv_thin_mfma_kernel(SpArgs a) {
  static_assert(CIN <= 16 && COUT <= 16, "one k-step, one row tile");
  constexpr int NP = NPROD == 1 ? 1 : 3;
  __shared__ u32x4 wl[27 * NP * 64];                          // 27 KB (one product) / 81 KB (three pieces)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  for (int u = threadIdx.x; u < 27 * NP * 64; u += 512) {      // packed: [27][piece 0..2][64]; one product: piece 0 only
    const int k = u / (NP * 64), e = (u / 64) % NP, l = u % 64;
    wl[u] = a.packed[((int64_t)k * 3 + e) * 64 + l];
  }
  __syncthreads();
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + 255) / 256;
  const bool loads = 8 * h < CIN;                              // C_in = 8: the upper k-half is zero
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t i = (tile * 8 + wave) * 32 + (lane & 31);
    const bool live = i < a.n_out;
    const int64_t ic = live ? i : a.n_out - 1;
    const int cx = a.out_coords[ic * 3 + 0], cy = a.out_coords[ic * 3 + 1], cz = a.out_coords[ic * 3 + 2];
    int rows[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
      int x, y, z;
      bool ok = live;
      if (MODE == MODE_SUBM) {
        x = cx + ox; y = cy + oy; z = cz + oz;
      } else if (MODE == MODE_DOWN) {
        x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz;
      } else {
        const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
        ok = ok && ((tx | ty | tz) & 1) == 0;
        x = tx >> 1; y = ty >> 1; z = tz >> 1;
      }
      ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
      const int r = a.in_table[ok ? ((int64_t)x * D + y) * D + z : 0];
      rows[k] = ok ? r : -1;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 cur[2], nxt[2];
    auto fetch = [&](int k, f32x4 (&v)[2]) {                  // this lane's eight channels of the neighbour row (row 0 when absent)
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.in + (int64_t)max(rows[k], 0) * CIN) + 2 * h;
      if (loads) { v[0] = src[0]; v[1] = src[1]; }
      else { v[0] = f32x4{0.f, 0.f, 0.f, 0.f}; v[1] = v[0]; }
    };
    fetch(0, cur);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if (k + 1 < 27) fetch(k + 1, nxt);
      const bool have = rows[k] >= 0;
      if (__ballot(have) != 0ull) {
        u32x4 b[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = have ? cur[j >> 1][2 * (j & 1)] : 0.f, x1 = have ? cur[j >> 1][2 * (j & 1) + 1] : 0.f;
          if constexpr (NPROD == 1) {
            b[0][j] = pack2(x0, x1);
          } else {
            uint32_t q[3];
            split3(x0, x1, q);
            b[0][j] = q[0]; b[1][j] = q[1]; b[2][j] = q[2];
          }
        }
        const u32x4* __restrict__ wf = wl + k * NP * 64 + lane;
#define SURF_MF(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)
        if constexpr (NPROD == 1) {
          const u32x4 a0 = wf[0];
          SURF_MF(a0, b[0]);
        } else {
          const u32x4 a0 = wf[0], a1 = wf[64], a2 = wf[128];
          SURF_MF(a2, b[0]);  // smallest terms first
          SURF_MF(a0, b[2]);
          SURF_MF(a1, b[1]);
          SURF_MF(a1, b[0]);
          SURF_MF(a0, b[1]);
          SURF_MF(a0, b[0]);
        }
#undef SURF_MF
      }
      cur[0] = nxt[0]; cur[1] = nxt[1];
    }
    if (!live) continue;
    // epilogue: BatchNorm(eval) + ReLU (+ skip); register 4 g + q = channel 8 g + 4 h + q
    float* __restrict__ dst = a.out + i * COUT;
    const float* __restrict__ sk = a.skip ? a.skip + i * COUT : nullptr;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = 8 * g + 4 * h;
      if (co >= COUT) continue;
      f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      if (a.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + co), sh = *reinterpret_cast<const f32x4*>(a.shift + co);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q] * sc[q] + sh[q], 0.f);
      }
      if (sk) v += *reinterpret_cast<const f32x4*>(sk + co);
      *reinterpret_cast<f32x4*>(dst + co) = v;
    }
  }
}

template <int CIN, int COUT>
int64_t packed_bytes() { return (int64_t)27 * Shape<CIN, COUT>::FRAGS * 64 * 16; }

}  // namespace

// The wide pairs; (16,16) on spconv_thin_mfma_kernel (0.40 -> 0.35 ms per training step); the pairs with 8 channels are
// instantiated for the A/B switch SURF_THIN_MFMA of surf_amd/ops.py only (measured slower than the per-voxel FMA kernels in both
// precisions: the host does not pack their weights by default).
#define SPM_CASES(X) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 32) X(32, 16) X(8, 8) X(16, 8) X(8, 16)

// 0 when the channel pair has no matrix-core kernel
extern "C" int64_t surf_spconv_packed_bytes(int cin, int cout) {
#define X(CI, CO) if (cin == CI && cout == CO) return packed_bytes<CI, CO>();
  SPM_CASES(X)
#undef X
  return 0;
}

extern "C" int surf_spconv_pack_weights(const float* weight, int cin, int cout, void* packed, void* stream) {
  if (!weight || !packed) return SURF_E_ARG;
#define X(CI, CO)                                                                                              \
  if (cin == CI && cout == CO) {                                                                               \
    const int n = 27 * Shape<CI, CO>::KS * Shape<CI, CO>::MT * 64;                                             \
    hipLaunchKernelGGL((spconv_pack_kernel<CI, CO>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, \
                       weight, (u32x4*)packed);                                                                \
    return surf_check_launch();                                                                                \
  }
  SPM_CASES(X)
#undef X
  return SURF_E_LIMIT;
}

extern "C" int surf_spconv_mfma(const float* in, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                                int64_t n_out, int mode, const void* packed, int cout, const float* bn_scale,
                                const float* bn_shift, const float* skip, float* out, int bf16_operands, void* stream) {
  if (!in || !in_table || !out_coords || !packed || !out || n_out <= 0 || D_in < 1) return SURF_E_ARG;
  if (mode < 0 || mode > 2 || ((bn_scale == nullptr) != (bn_shift == nullptr))) return SURF_E_ARG;
  SpArgs a;
  a.in = in; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode;
  a.packed = (const u32x4*)packed; a.scale = bn_scale; a.shift = bn_shift; a.skip = skip; a.out = out;
  const int64_t blocks = (n_out + 127) / 128;
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  if (cin <= 16 && cout <= 16 && !getenv("SURF_SPCONV_THIN_OLD")) {        // thin pairs: the pipelined kernel, persistent workgroups
    const int64_t tiles = (n_out + 255) / 256;
    // LDS: 27 KB (one product) lets four 512-thread workgroups share a CU, 81 KB (exact split) one
    const int64_t cap = bf16_operands ? 1024 : 256;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
#define T3(CI, CO, NPR)                                                                                                       \
    if (mode == MODE_SUBM) hipLaunchKernelGGL((spconv_thin_mfma_kernel<CI, CO, MODE_SUBM, NPR>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); \
    else if (mode == MODE_DOWN) hipLaunchKernelGGL((spconv_thin_mfma_kernel<CI, CO, MODE_DOWN, NPR>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); \
    else hipLaunchKernelGGL((spconv_thin_mfma_kernel<CI, CO, MODE_UP, NPR>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
#define TX(CI, CO)                                    \
    if (cin == CI && cout == CO) {                    \
      if (bf16_operands) { T3(CI, CO, 1) }            \
      else { T3(CI, CO, 6) }                          \
      return surf_check_launch();                     \
    }
    TX(8, 8) TX(16, 8) TX(8, 16) TX(16, 16)
#undef TX
#undef T3
  }
#define X(CI, CO)                                                                                                  \
  if (cin == CI && cout == CO) {                                                                                   \
    if (bf16_operands)                                                                                             \
      hipLaunchKernelGGL((spconv_mfma_kernel<CI, CO, 1>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a); \
    else                                                                                                           \
      hipLaunchKernelGGL((spconv_mfma_kernel<CI, CO, 6>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a); \
    return surf_check_launch();                                                                                    \
  }
  SPM_CASES(X)
#undef X
  return SURF_E_LIMIT;
}
