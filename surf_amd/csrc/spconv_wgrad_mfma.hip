// K5w (round 5): weight gradient of the sparse 3^3 convolutions on the matrix cores, for the layers with C_in, C_out <= 32
// (reg_network.py:38-88 under loss.backward(), runner.py:163).  Same function as spconv_wgrad_kernel / spconv_wgrad_thin_kernel
// (spconv_bwd.hip):
//     dW[k][ci][co] = sum over output sites i whose offset-k neighbour j exists of  x[j][ci] * dy[i][co]
// For one offset this is the GEMM  dW_k (C_in x C_out) = X_k^T (C_in x sites) . dY (sites x C_out)  whose contraction runs over
// the SITES; X_k = the neighbour rows gathered through the index table (absent neighbours = zero rows).  The per-voxel kernels
// do it with LDS-fed FMAs at ~17 TFLOP/s (7.8 ms per training step for the <16,8> layers alone: the largest sparse-conv kernel
// of the step); here a workgroup takes 64 sites at a time, looks their 27 x 64 neighbours up ONCE (LDS), and each of its four
// wavefronts owns seven of the 27 offsets: per offset and 16-site k-step
//   A (32 rows = input channels x 16 sites): lane (ci, site half) gathers x[nbr][ci] for its 8 sites - the 32 lanes of a half
//     read one 4 C_in-byte row segment per load instruction (coalesced); rows re-read by other offsets / sites come from L2
//   B (16 sites x 32 columns = output channels): dy of the k-step, split ONCE per tile and reused by all 27 offsets
//   v_mfma_f32_32x32x16_bf16 into the offset's accumulator tile, which stays in registers over all tiles of the workgroup
// and is added to dW with float atomics at the end (<= 27 x C_in x C_out per workgroup, as the per-voxel kernels do).
// Both operands are split exactly into three bf16 pieces, six products per k-step (fp32-equivalent, whatever the training
// policy: a one-product bf16 form measured SLOWER - 2.3 vs 1.3 ms on <8,16> - the kernel is bound by its gathers, and without the
// six MFMAs nothing hides their latency).
// Measured per training step against the per-voxel kernels (profiles/r05_wgrad_mfma_ab.txt): <8,16> 1.86 -> 1.32 ms, <16,32>
// 1.30 -> 0.41, <32,16> 1.48 -> 0.91, <32,32> 1.19 -> 0.50: those four pairs use it.  NOT the layers on the finest lattices:
// <16,8> 7.8 -> 14.0 ms, <16,16> 1.27 -> 1.25, <8,8> 0.23 -> 0.29 - there spconv_wgrad_thin_kernel's LDS cache of the tile's
// ~450 distinct neighbour rows beats 27 x 64 row gathers from L2, and with 16 x 8 real entries of a 32 x 32 tile the six-product
// MFMA form needs as many matrix cycles as the FMA form needs VALU cycles.
#include "common.h"

namespace {

enum { MODE_SUBM = 0, MODE_DOWN = 1, MODE_UP = 2 };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct WgmArgs {
  const float* x;            // (n_in, CIN)
  const int32_t* in_table;   // (Din^3)
  int Din;
  const int32_t* out_coords; // (n_out, 3)
  int64_t n_out;
  int mode;
  const float* dy;           // (n_out, COUT)
  float* dW;                 // (27, CIN, COUT), accumulated
};

// 8 fp32 values (k-slots 0..7 of a lane) -> NP packed fragments
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&f)[3]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint32_t q[3];
    surf_split3_bf16(v[2 * p], v[2 * p + 1], q);
    f[0][p] = q[0]; f[1][p] = q[1]; f[2][p] = q[2];
  }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_mfma_kernel(WgmArgs a) {
  static_assert(CIN <= 32 && COUT <= 32, "one 32 x 32 accumulator tile per offset");
  constexpr int TS = 64, NG = TS / 16, NP = 3, KPW = 7;   // 4 wavefronts x 7 offsets >= 27
  __shared__ __attribute__((aligned(16))) int nbr[27 * TS];               // neighbour row of (offset, site) or -1
  __shared__ __attribute__((aligned(16))) float dyt[TS * 32];              // dy of the tile, columns padded to 32 with zeros
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kg = lane >> 5;
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  f32x16 acc[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                                              // the previous tile's LDS is consumed
    // ---- neighbour rows of the tile's 27 x 64 (offset, site) references, one table lookup each
    // (submanifold / stride-2 windows: the three z-neighbours of a column as one 12-byte load, common.h surf_table_column3; sites
    // on the lattice's z border and the transposed mode take the single lookups below)
    const bool columns = SURF_SPCONV_TRIPLE && a.mode != MODE_UP;
    if (columns)
      for (int q = threadIdx.x; q < 9 * TS; q += 256) {
        const int j = q / TS, st = q % TS;
        const int64_t si = tile * TS + st;
        I3u t3 = {-1, -1, -1};
        if (si < a.n_out) {
          const int f = a.mode == MODE_DOWN ? 2 : 1;
          const int bx = f * a.out_coords[si * 3 + 0], by = f * a.out_coords[si * 3 + 1], bz = f * a.out_coords[si * 3 + 2];
          if (bz - 1 >= 0 && bz + 1 < D) t3 = surf_table_column3(a.in_table, D, bx + j % 3 - 1, by + j / 3 - 1, bz);
          else t3 = I3u{-2, -2, -2};                             // -2: looked up entry by entry below
        }
        nbr[j * TS + st] = t3.a;
        nbr[(9 + j) * TS + st] = t3.b;
        nbr[(18 + j) * TS + st] = t3.c;
      }
    if (columns) __syncthreads();
    for (int q = threadIdx.x; q < 27 * TS; q += 256) {
      const int k = q / TS, st = q % TS;
      if (columns && nbr[q] != -2) continue;
      const int64_t si = tile * TS + st;
      int row = -1;
      if (si < a.n_out) {
        const int cx = a.out_coords[si * 3 + 0], cy = a.out_coords[si * 3 + 1], cz = a.out_coords[si * 3 + 2];
        const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
        int x, y, z;
        bool ok = true;
        if (a.mode == MODE_SUBM) { x = cx + ox; y = cy + oy; z = cz + oz; }
        else if (a.mode == MODE_DOWN) { x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz; }
        else {
          const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
          ok = ((tx | ty | tz) & 1) == 0;
          x = tx >> 1; y = ty >> 1; z = tz >> 1;
        }
        ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
        if (ok) row = a.in_table[((int64_t)x * D + y) * D + z];
      }
      nbr[q] = row;
    }
    // ---- dy of the tile (zero rows past the end, zero columns past C_out)
    for (int e = threadIdx.x; e < TS * 32; e += 256) {
      const int st = e >> 5, c = e & 31;
      const int64_t si = tile * TS + st;
      dyt[e] = (c < COUT && si < a.n_out) ? a.dy[si * COUT + c] : 0.f;
    }
    __syncthreads();
    // ---- B fragments of the four k-steps: lane (column co, site half kg) holds dy[16 g + 8 kg + j][co], j = 0..7
    u32x4 bf[NG][NP];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = dyt[(16 * g + 8 * kg + j) * 32 + col];
      split8(v, bf[g]);
    }
    // ---- this wavefront's offsets: KPW x NG steps (offset k = wave + 4 i, k-step g), software-pipelined by hand - the gathers
    // of step s + 1 are issued before step s is multiplied
    auto fetch = [&](int s, float (&v)[8]) __attribute__((always_inline)) -> bool {
      const int k = wave + 4 * (s / NG), g = s % NG;
      bool any = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (k < 27) {
        const int4 r0 = *reinterpret_cast<const int4*>(&nbr[k * TS + 16 * g + 8 * kg]);
        const int4 r1 = *reinterpret_cast<const int4*>(&nbr[k * TS + 16 * g + 8 * kg + 4]);
        const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          any = any || rows[j] >= 0;
          if (rows[j] >= 0 && col < CIN) v[j] = a.x[(int64_t)rows[j] * CIN + col];
        }
      }
      return __ballot(any) != 0ull;                             // wave-uniform: some neighbour exists in this k-step
    };
    float va[8], vb[8];
    bool live_a = fetch(0, va), live_b = false;
#pragma unroll
    for (int s = 0; s < KPW * NG; ++s) {
      float (&cur)[8] = (s & 1) ? vb : va;
      float (&nxt)[8] = (s & 1) ? va : vb;
      const bool live = (s & 1) ? live_b : live_a;
      if (s + 1 < KPW * NG) {
        const bool l = fetch(s + 1, nxt);
        if (s & 1) live_a = l; else live_b = l;
      }
      if (!live) continue;
      const int i = s / NG, g = s % NG;
      u32x4 af[NP];
      split8(cur, af);
#define SURF_MF(xp, yp) \
  acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xp), __builtin_bit_cast(bf16x8, yp), acc[i], 0, 0, 0)
      SURF_MF(af[2], bf[g][0]);  // smallest terms first
      SURF_MF(af[0], bf[g][2]);
      SURF_MF(af[1], bf[g][1]);
      SURF_MF(af[1], bf[g][0]);
      SURF_MF(af[0], bf[g][1]);
      SURF_MF(af[0], bf[g][0]);
#undef SURF_MF
    }
  }
  // ---- accumulator register r of lane (col, kg): row ci = (r & 3) + 8 (r >> 2) + 4 kg, column co = col
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int k = wave + 4 * i;
    if (k >= 27) break;
    float* dst = a.dW + (int64_t)k * CIN * COUT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (ci < CIN && col < COUT && acc[i][r] != 0.f) atomicAdd(dst + ci * COUT + col, acc[i][r]);
    }
  }
}

}  // namespace

#define WGM_CASES(X) X(8, 16) X(16, 32) X(32, 16) X(32, 32)

extern "C" int surf_spconv_wgrad_mfma_supported(int cin, int cout) {
#define X(CI, CO) if (cin == CI && cout == CO) return 1;
  WGM_CASES(X)
#undef X
  return 0;
}

extern "C" int surf_spconv_wgrad_mfma(const float* x, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                                      int64_t n_out, int mode, const float* dy, int cout, float* dW, void* stream) {
  if (!x || !in_table || !out_coords || !dy || !dW || n_out <= 0 || D_in < 1 || mode < 0 || mode > 2) return SURF_E_ARG;
  WgmArgs a;
  a.x = x; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode; a.dy = dy; a.dW = dW;
  const int64_t tiles = (n_out + 63) / 64;
  const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
#define X(CI, CO)                                                                                                       \
  if (cin == CI && cout == CO) {                                                                                        \
    hipLaunchKernelGGL((spconv_wgrad_mfma_kernel<CI, CO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);            \
    return surf_check_launch();                                                                                         \
  }
  WGM_CASES(X)
#undef X
  return SURF_E_LIMIT;
}
