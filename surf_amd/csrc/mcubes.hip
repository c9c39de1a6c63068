// K13: iso-surface extraction (marching cubes) of the SDF lattice of extract_geometry.
//
// Replaces mcubes.marching_cubes(u, threshold) (models/modules/implicit_surface.py:353; PyMCubes 0.1.4, third party):
// the classic 256-case table (mc_tables.h), `u <= isovalue` inside test, one vertex per sign-changing lattice edge placed
// by linear interpolation in double precision, vertices in lattice-index units.  Vertex order: by (owner lattice point,
// axis); triangle order: by cell (x outermost, as PyMCubes iterates) then table order.
//
// Everything is HBM-bound integer / byte work on a (nx, ny, nz) lattice (512^3 = 134 M points in validate):
//   classify : one thread per lattice point, 4 coalesced float loads per point (self + three neighbours) for the edge
//              bits and 8 for the cell case -> one flag byte: bits 0-2 = sign change along x / y / z from this point,
//              bits 3-5 = number of triangles of the cell whose origin it is
//   (surf_compact on the flag bytes gives the active points, ~res^2 of the res^3)
//   count    : per-1024-entry block sums of vertex / triangle counts + single-block scan of the block sums
//   emit     : block-local scans give every active point its vertex / triangle offsets; vertices are written and the
//              first vertex id of every point is recorded in a dense int32 lattice (vbase) so that the triangle pass
//              can name the vertex of any cell edge as vbase[owner point] + rank(axis)
#include <math.h>

#define MC_TABLE_QUAL __constant__
#include "common.h"
#include "mc_tables.h"

namespace {

constexpr int MC_BLOCK = 1024;  // active entries per workgroup in count / emit

struct McDims { int nx, ny, nz; };

__device__ __forceinline__ int64_t lin(const McDims& d, int x, int y, int z) { return ((int64_t)x * d.ny + y) * d.nz + z; }

__global__ __launch_bounds__(256) void mc_classify_kernel(const float* __restrict__ u, McDims d, double iso,
                                                          uint8_t* __restrict__ flags) {
  const int64_t n = (int64_t)d.nx * d.ny * d.nz;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i % d.nz), y = (int)((i / d.nz) % d.ny), x = (int)(i / ((int64_t)d.nz * d.ny));
  const bool in0 = (double)u[i] <= iso;
  unsigned f = 0;
  const bool hx = x + 1 < d.nx, hy = y + 1 < d.ny, hz = z + 1 < d.nz;
  bool c[8];
  c[0] = in0;
  c[1] = hx ? ((double)u[lin(d, x + 1, y, z)] <= iso) : in0;
  c[3] = hy ? ((double)u[lin(d, x, y + 1, z)] <= iso) : in0;
  c[4] = hz ? ((double)u[lin(d, x, y, z + 1)] <= iso) : in0;
  if (hx && c[1] != in0) f |= 1u;
  if (hy && c[3] != in0) f |= 2u;
  if (hz && c[4] != in0) f |= 4u;
  if (hx && hy && hz) {
    c[2] = (double)u[lin(d, x + 1, y + 1, z)] <= iso;
    c[5] = (double)u[lin(d, x + 1, y, z + 1)] <= iso;
    c[6] = (double)u[lin(d, x + 1, y + 1, z + 1)] <= iso;
    c[7] = (double)u[lin(d, x, y + 1, z + 1)] <= iso;
    unsigned cs = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cs |= (c[k] ? 1u : 0u) << k;
    f |= (unsigned)MC_NTRI[cs] << 3;
  }
  flags[i] = (uint8_t)f;
}

__device__ __forceinline__ int nvert_of(unsigned f) { return __popc(f & 7u); }
__device__ __forceinline__ int ntri_of(unsigned f) { return (int)((f >> 3) & 7u); }

// block sums of (vertex count, triangle count) over MC_BLOCK active entries: ws[b], ws[nb + b]
__global__ __launch_bounds__(256) void mc_blocksum_kernel(const uint8_t* __restrict__ flags, const int32_t* __restrict__ active,
                                                          int64_t m, int nb, int32_t* __restrict__ ws) {
  __shared__ int s_v[4], s_t[4];
  int v = 0, t = 0;
  for (int k = 0; k < MC_BLOCK / 256; ++k) {
    const int64_t e = (int64_t)blockIdx.x * MC_BLOCK + k * 256 + threadIdx.x;
    if (e < m) {
      const unsigned f = flags[active[e]];
      v += nvert_of(f);
      t += ntri_of(f);
    }
  }
  v = (int)wave_sum((float)v);
  t = (int)wave_sum((float)t);
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = v; s_t[threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = s_v[0] + s_v[1] + s_v[2] + s_v[3];
    ws[nb + blockIdx.x] = s_t[0] + s_t[1] + s_t[2] + s_t[3];
  }
}

// exclusive scans of the two block-sum arrays in place by one workgroup; totals[0] = vertices, totals[1] = triangles
__global__ __launch_bounds__(1024) void mc_scan_kernel(int32_t* __restrict__ ws, int nb, int32_t* __restrict__ totals) {
  __shared__ int s_buf[1024];
  __shared__ int s_carry;
  for (int arr = 0; arr < 2; ++arr) {
    int32_t* a = ws + (int64_t)arr * nb;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nb; c0 += 1024) {
      const int i = c0 + threadIdx.x;
      const int v = i < nb ? a[i] : 0;
      s_buf[threadIdx.x] = v;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1) {
        const int t = threadIdx.x >= o ? s_buf[threadIdx.x - o] : 0;
        __syncthreads();
        s_buf[threadIdx.x] += t;
        __syncthreads();
      }
      const int incl = s_buf[threadIdx.x];
      const int carry = s_carry;
      if (i < nb) a[i] = carry + incl - v;
      __syncthreads();
      if (threadIdx.x == 1023) s_carry = carry + incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) totals[arr] = s_carry;
    __syncthreads();
  }
}

// exclusive scan of one value per thread over a 1024-thread workgroup
__device__ __forceinline__ int block_excl_scan_1024(int c, int* s_part /*16*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += s_part[w];
  __syncthreads();
  return off + incl - c;
}

__global__ __launch_bounds__(1024) void mc_vertex_kernel(const float* __restrict__ u, McDims d, double iso,
                                                         const uint8_t* __restrict__ flags, const int32_t* __restrict__ active,
                                                         int64_t m, const int32_t* __restrict__ ws, int32_t* __restrict__ vbase,
                                                         double* __restrict__ vertices) {
  __shared__ int s_part[16];
  const int64_t e = (int64_t)blockIdx.x * MC_BLOCK + threadIdx.x;
  const int32_t p = e < m ? active[e] : 0;
  const unsigned f = e < m ? flags[p] : 0u;
  const int nv = nvert_of(f);
  int off = ws[blockIdx.x] + block_excl_scan_1024(nv, s_part);
  if (e >= m) return;
  vbase[p] = off;
  if (nv == 0) return;
  const int z = p % d.nz, y = (p / d.nz) % d.ny, x = p / (d.nz * d.ny);
  const double f1 = (double)u[p];
#pragma unroll
  for (int axis = 0; axis < 3; ++axis) {
    if (!(f & (1u << axis))) continue;
    const int64_t q = axis == 0 ? lin(d, x + 1, y, z) : (axis == 1 ? lin(d, x, y + 1, z) : lin(d, x, y, z + 1));
    const double f2 = (double)u[q];
    // PyMCubes mc_isovalue_interpolation: (x2 - x1) (isovalue - f1) / (f2 - f1) + x1 with x2 - x1 = 1 lattice step
    const double t = f2 == f1 ? 0.5 : (1.0 * (iso - f1)) / (f2 - f1);
    double vx = (double)x, vy = (double)y, vz = (double)z;
    if (axis == 0) vx = t + vx;
    else if (axis == 1) vy = t + vy;
    else vz = t + vz;
    vertices[(int64_t)off * 3 + 0] = vx;
    vertices[(int64_t)off * 3 + 1] = vy;
    vertices[(int64_t)off * 3 + 2] = vz;
    ++off;
  }
}

__global__ __launch_bounds__(1024) void mc_triangle_kernel(const float* __restrict__ u, McDims d, double iso,
                                                           const uint8_t* __restrict__ flags, const int32_t* __restrict__ active,
                                                           int64_t m, int nb, const int32_t* __restrict__ ws,
                                                           const int32_t* __restrict__ vbase, int32_t* __restrict__ triangles) {
  __shared__ int s_part[16];
  const int64_t e = (int64_t)blockIdx.x * MC_BLOCK + threadIdx.x;
  const int32_t p = e < m ? active[e] : 0;
  const unsigned f = e < m ? flags[p] : 0u;
  const int nt = ntri_of(f);
  int off = ws[nb + blockIdx.x] + block_excl_scan_1024(nt, s_part);
  if (e >= m || nt == 0) return;
  const int z = p % d.nz, y = (p / d.nz) % d.ny, x = p / (d.nz * d.ny);
  // corners in table order and their lattice indices
  const int cx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, cy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  unsigned cs = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) cs |= ((double)u[lin(d, x + cx[k], y + cy[k], z + cz[k])] <= iso ? 1u : 0u) << k;
  // owner corner and axis of the 12 cell edges
  const int eo[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3}, ea[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int edge = MC_TRI[cs][3 * t + k];
      const int oc = eo[edge], axis = ea[edge];
      const int64_t q = lin(d, x + cx[oc], y + cy[oc], z + cz[oc]);
      const unsigned fq = flags[q];
      triangles[(int64_t)off * 3 + k] = vbase[q] + __popc(fq & ((1u << axis) - 1u));
    }
    ++off;
  }
}

}  // namespace

extern "C" int surf_mc_classify(const float* u, int nx, int ny, int nz, double iso, uint8_t* flags, void* stream) {
  if (!u || !flags) return SURF_E_ARG;
  if (nx < 1 || ny < 1 || nz < 1) return SURF_E_ARG;
  const int64_t n = (int64_t)nx * ny * nz;
  if (n >= ((int64_t)1 << 31)) return SURF_E_LIMIT;  // int32 lattice indices (surf_compact)
  McDims d{nx, ny, nz};
  hipLaunchKernelGGL(mc_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, d, iso, flags);
  return surf_check_launch();
}

extern "C" int64_t surf_mc_workspace_ints(int64_t n_active) {
  const int64_t nb = (n_active + MC_BLOCK - 1) / MC_BLOCK;
  return 2 * (nb > 0 ? nb : 1);
}

extern "C" int surf_mc_count(const uint8_t* flags, const int32_t* active, int64_t n_active, int32_t* workspace, int32_t* totals,
                             void* stream) {
  if (!flags || !active || !workspace || !totals || n_active <= 0) return SURF_E_ARG;
  const int nb = (int)((n_active + MC_BLOCK - 1) / MC_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mc_blocksum_kernel, dim3(nb), dim3(256), 0, st, flags, active, n_active, nb, workspace);
  hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, st, workspace, nb, totals);
  return surf_check_launch();
}

extern "C" int surf_mc_emit(const float* u, int nx, int ny, int nz, double iso, const uint8_t* flags, const int32_t* active,
                            int64_t n_active, const int32_t* workspace, int32_t* vbase, double* vertices, int32_t* triangles,
                            void* stream) {
  if (!u || !flags || !active || !workspace || !vbase || n_active <= 0) return SURF_E_ARG;
  const int nb = (int)((n_active + MC_BLOCK - 1) / MC_BLOCK);
  McDims d{nx, ny, nz};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mc_vertex_kernel, dim3(nb), dim3(1024), 0, st, u, d, iso, flags, active, n_active, workspace, vbase, vertices);
  hipLaunchKernelGGL(mc_triangle_kernel, dim3(nb), dim3(1024), 0, st, u, d, iso, flags, active, n_active, nb, workspace, vbase,
                     triangles);
  return surf_check_launch();
}
