// K5b  weight gradient of a sparse 3^3 convolution (first backward kernel of the volume-build side, row f2 / K12):
//     dW[k][ci][co] = sum over output sites i whose offset-k neighbour j exists of  x[j][ci] * dy[i][co]
// (the input gradient of a sparse convolution is itself a sparse convolution - submanifold with mirrored offsets, stride-2
// down <-> transposed up - with the transposed kernel slices, so it runs on spconv.hip's kernels: ops.spconv_backward).
// One workgroup sweeps tiles of 64 (wide layers) or 256 (thin layers) output sites; per kernel offset the gathered input rows and
// the dy rows sit in LDS and every thread owns 4 x 2 register blocks of that offset's slice (thin layers: one block on a
// 1/G share of the tile's sites; 0.75 LDS words per FMA instead of 2), accumulated in registers over the workgroup's tiles and added to dW with one float atomic per
// entry and workgroup.
#include "common.h"

namespace {

enum { MODE_SUBM = 0, MODE_DOWN = 1, MODE_UP = 2 };

struct WgArgs {
  const float* x;            // (n_in, CIN)
  const int32_t* in_table;   // (Din^3)
  int Din;
  const int32_t* out_coords; // (n_out, 3)
  int64_t n_out;
  int mode;
  const float* dy;           // (n_out, COUT)
  float* dW;                 // (27, CIN, COUT), accumulated
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(WgArgs a) {
  constexpr int TS = (CIN + COUT <= 48) ? 256 : 64;   // output sites per tile (LDS: TS (CIN + COUT + 6) floats)
  constexpr int NB = (CIN / 4) * (COUT / 2);          // 4 x 2 register blocks of the offset's (CIN, COUT) slice
  constexpr int PERB = NB >= 256 ? NB / 256 : 1;      // blocks per thread
  constexpr int G = NB >= 256 ? 1 : 256 / NB;         // site groups sharing one block (thin layers: every thread has work)
  constexpr int XS = CIN + 4, DS = COUT + 2;          // padded row strides (16-B / 8-B aligned vector reads)
  static_assert(NB >= 256 ? NB % 256 == 0 : 256 % NB == 0, "(CIN / 4) (COUT / 2) must divide or be a multiple of 256");
  __shared__ __attribute__((aligned(16))) float xs[TS * XS];
  __shared__ __attribute__((aligned(16))) float ds[TS * DS];
  __shared__ int rows[TS];
  __shared__ float red[NB >= 256 ? 1 : 256 * 8];
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  const int g = NB >= 256 ? 0 : threadIdx.x / NB;
  const int b0 = NB >= 256 ? threadIdx.x : threadIdx.x % NB;
  for (int k = 0; k < 27; ++k) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    float acc[PERB][4][2];
#pragma unroll
    for (int e = 0; e < PERB; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[e][q][0] = acc[e][q][1] = 0.f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      __syncthreads();
      for (int t = threadIdx.x; t < TS; t += 256) {
        const int64_t i = tile * TS + t;
        int row = -1;
        if (i < a.n_out) {
          const int cx = a.out_coords[i * 3 + 0], cy = a.out_coords[i * 3 + 1], cz = a.out_coords[i * 3 + 2];
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
        rows[t] = row;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < TS * (CIN / 4); e += 256) {       // 16-B gathers of the neighbour rows
        const int s = e / (CIN / 4), c4 = e % (CIN / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rows[s] >= 0) v = *reinterpret_cast<const f32x4*>(a.x + (int64_t)rows[s] * CIN + c4 * 4);
        *reinterpret_cast<f32x4*>(xs + s * XS + c4 * 4) = v;
      }
      for (int e = threadIdx.x; e < TS * COUT; e += 256) {
        const int s = e / COUT, c = e % COUT;
        const int64_t i = tile * TS + s;
        ds[s * DS + c] = (i < a.n_out && rows[s] >= 0) ? a.dy[i * COUT + c] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < PERB; ++e) {
        const int blk = b0 + 256 * e;
        const int ci0 = (blk / (COUT / 2)) * 4, co0 = (blk % (COUT / 2)) * 2;
#pragma unroll 4
        for (int t = g; t < TS; t += G) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + t * XS + ci0);
          const float d0 = ds[t * DS + co0], d1 = ds[t * DS + co0 + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[e][q][0] = fmaf(xv[q], d0, acc[e][q][0]);
            acc[e][q][1] = fmaf(xv[q], d1, acc[e][q][1]);
          }
        }
      }
    }
    float* dst = a.dW + (int64_t)k * CIN * COUT;
    if (NB >= 256) {
#pragma unroll
      for (int e = 0; e < PERB; ++e) {
        const int blk = b0 + 256 * e;
        const int ci0 = (blk / (COUT / 2)) * 4, co0 = (blk % (COUT / 2)) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (acc[e][q][r] != 0.f) atomicAdd(dst + (ci0 + q) * COUT + co0 + r, acc[e][q][r]);
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[(q * 2 + 0) * 256 + threadIdx.x] = acc[0][q][0];
        red[(q * 2 + 1) * 256 + threadIdx.x] = acc[0][q][1];
      }
      __syncthreads();
      for (int o = threadIdx.x; o < NB * 8; o += 256) {                  // o = entry (0..7) x block
        const int ent = o / NB, blk = o % NB;
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += red[ent * 256 + q * NB + blk];
        const int ci = (blk / (COUT / 2)) * 4 + ent / 2, co = (blk % (COUT / 2)) * 2 + (ent & 1);
        if (t != 0.f) atomicAdd(dst + ci * COUT + co, t);
      }
    }
  }
}

}  // namespace

#define WG_CASES(X) X(8, 8) X(16, 8) X(8, 16) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 32) X(32, 16)

extern "C" int surf_spconv_wgrad(const float* x, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                                 int64_t n_out, int mode, const float* dy, int cout, float* dW, void* stream) {
  if (!x || !in_table || !out_coords || !dy || !dW || n_out <= 0 || D_in < 1 || mode < 0 || mode > 2) return SURF_E_ARG;
  WgArgs a;
  a.x = x; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode; a.dy = dy; a.dW = dW;
  const int ts = (cin + cout <= 48) ? 256 : 64;
  const int64_t tiles = (n_out + ts - 1) / ts;
  const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
#define X(CI, CO)                                                                                              \
  if (cin == CI && cout == CO) {                                                                               \
    hipLaunchKernelGGL((spconv_wgrad_kernel<CI, CO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);        \
    return surf_check_launch();                                                                                \
  }
  WG_CASES(X)
#undef X
  return SURF_E_LIMIT;
}
