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
  static constexpr int KS = CIN / 16, MT = (COUT + 31) / 32;
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
    const float a = co < COUT ? w[((int64_t)k * CIN + ci0 + 2 * j) * COUT + co] : 0.f;
    const float b = co < COUT ? w[((int64_t)k * CIN + ci0 + 2 * j + 1) * COUT + co] : 0.f;
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
      if (row >= 0) { xv[ks][0] = src[4 * ks]; xv[ks][1] = src[4 * ks + 1]; }
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

template <int CIN, int COUT>
int64_t packed_bytes() { return (int64_t)27 * Shape<CIN, COUT>::FRAGS * 64 * 16; }

}  // namespace

#define SPM_CASES(X) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 32) X(32, 16)

// 0 when the channel pair has no matrix-core kernel (C_in or C_out < 16: those layers are gather bound, spconv.hip)
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
