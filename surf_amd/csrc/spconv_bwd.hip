// K5b  weight gradient of a sparse 3^3 convolution (first backward kernel of the volume-build side, row f2 / K12):
//     dW[k][ci][co] = sum over output sites i whose offset-k neighbour j exists of  x[j][ci] * dy[i][co]
// (the input gradient of a sparse convolution is itself a sparse convolution - submanifold with mirrored offsets, stride-2
// down <-> transposed up - with the transposed kernel slices, so it runs on spconv.hip's kernels: ops.spconv_backward).
// One workgroup sweeps tiles of 64 output sites; per kernel offset the 64 gathered input rows and the 64 dy rows sit in LDS and
// every thread owns C_in C_out / 256 entries of that offset's slice, accumulated in registers over the workgroup's tiles and
// added to dW with one float atomic per entry and workgroup.
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
  constexpr int TS = 64;                         // output sites per tile
  constexpr int PER = (CIN * COUT + 255) / 256;  // dW entries per thread
  __shared__ float xs[TS][CIN + 1], ds[TS][COUT + 1];
  __shared__ int rows[TS];
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  for (int k = 0; k < 27; ++k) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    float acc[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] = 0.f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      __syncthreads();
      if (threadIdx.x < TS) {
        const int64_t i = tile * TS + threadIdx.x;
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
        rows[threadIdx.x] = row;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < TS * CIN; e += 256) {
        const int s = e / CIN, c = e % CIN;
        xs[s][c] = rows[s] >= 0 ? a.x[(int64_t)rows[s] * CIN + c] : 0.f;
      }
      for (int e = threadIdx.x; e < TS * COUT; e += 256) {
        const int s = e / COUT, c = e % COUT;
        const int64_t i = tile * TS + s;
        ds[s][c] = (i < a.n_out && rows[s] >= 0) ? a.dy[i * COUT + c] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int idx = threadIdx.x + 256 * e;
        if (idx < CIN * COUT) {
          const int ci = idx / COUT, co = idx % COUT;
          float s = acc[e];
#pragma unroll 8
          for (int t = 0; t < TS; ++t) s = fmaf(xs[t][ci], ds[t][co], s);
          acc[e] = s;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = threadIdx.x + 256 * e;
      if (idx < CIN * COUT && acc[e] != 0.f) atomicAdd(a.dW + (int64_t)k * CIN * COUT + idx, acc[e]);
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
  const int64_t tiles = (n_out + 63) / 64;
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
#define X(CI, CO)                                                                                              \
  if (cin == CI && cout == CO) {                                                                               \
    hipLaunchKernelGGL((spconv_wgrad_kernel<CI, CO>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);        \
    return surf_check_launch();                                                                                \
  }
  WG_CASES(X)
#undef X
  return SURF_E_LIMIT;
}
