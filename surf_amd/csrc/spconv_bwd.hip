// K5b  weight gradient of a sparse 3^3 convolution (first backward kernel of the volume-build side, row f2 / K12):
//     dW[k][ci][co] = sum over output sites i whose offset-k neighbour j exists of  x[j][ci] * dy[i][co]
// (the input gradient of a sparse convolution is itself a sparse convolution - submanifold with mirrored offsets, stride-2
// down <-> transposed up - with the transposed kernel slices, so it runs on spconv.hip's kernels: ops.spconv_backward).
// One workgroup sweeps tiles of 64 (wide layers) or 256 (thin layers) output sites; per kernel offset the gathered input rows and
// the dy rows sit in LDS and every thread owns C_in C_out / 256 entries of that offset's slice (thin layers: one entry on a
// 1/G share of the tile's sites), accumulated in registers over the workgroup's tiles and added to dW with one float atomic per
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
  constexpr int O = CIN * COUT;
  constexpr int TS = (CIN + COUT <= 48) ? 256 : 64;   // output sites per tile (LDS: TS (CIN + COUT + 2) floats)
  constexpr int PER = O >= 256 ? O / 256 : 1;         // dW entries per thread
  constexpr int G = O >= 256 ? 1 : 256 / O;           // site groups sharing one entry (thin layers: every thread has work)
  static_assert(O >= 256 ? O % 256 == 0 : 256 % O == 0, "CIN * COUT must divide or be a multiple of 256");
  __shared__ float xs[TS][CIN + 1], ds[TS][COUT + 1];
  __shared__ int rows[TS];
  __shared__ float red[256];
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  const int g = O >= 256 ? 0 : threadIdx.x / O;
  const int e0 = O >= 256 ? threadIdx.x : threadIdx.x % O;
  for (int k = 0; k < 27; ++k) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    float acc[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] = 0.f;
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
        const int idx = e0 + 256 * e;
        const int ci = idx / COUT, co = idx % COUT;
        float s = acc[e];
#pragma unroll 8
        for (int t = g; t < TS; t += G) s = fmaf(xs[t][ci], ds[t][co], s);
        acc[e] = s;
      }
    }
    if (O >= 256) {
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (acc[e] != 0.f) atomicAdd(a.dW + (int64_t)k * O + e0 + 256 * e, acc[e]);
    } else {
      __syncthreads();
      red[threadIdx.x] = acc[0];
      __syncthreads();
      if (threadIdx.x < O) {
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += red[q * O + threadIdx.x];
        if (t != 0.f) atomicAdd(a.dW + (int64_t)k * O + threadIdx.x, t);
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
