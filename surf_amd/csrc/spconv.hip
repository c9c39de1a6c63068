// K5: sparse 3D convolutions of the cost-regularisation U-Net (3^3 kernels; submanifold, stride-2 down,
// transposed stride-2 up) with BatchNorm(eval) + ReLU (+ skip add) fused in the epilogue.
//
// Restates SparseCostRegNet  reg_network.py:38-88  over torchsparse 2.1.0 `spnn.Conv3d / BatchNorm / ReLU`
// (third-party, source absent: PARITY UNPINNED -- semantics documented in oracle/surf_oracle.py sparse_unet).
//
// Output-stationary gather form, no atomics: one thread owns one output voxel and all C_out channels; for each
// of the 27 offsets it looks the contributing input row up in a dense int32 index table, reads it with 16-byte
// loads and applies the (C_in x C_out) slice of the kernel, whose entries are wave-uniform (scalar loads).
// Kernel-offset enumeration: x fastest, z slowest (k = (oz+1)*9 + (oy+1)*3 + (ox+1)), torchsparse's order for
// odd kernel volumes.
#include "common.h"

namespace {

enum { MODE_SUBM = 0, MODE_DOWN = 1, MODE_UP = 2 };

struct SpconvArgs {
  const float* in;          // (n_in, CIN)
  const uint16_t* in16;     // (n_in, CIN) the same rows rounded to bf16 (spconv_pipe_kernel<.., true>), or null
  const int32_t* in_table;  // (Din^3) row of the input site or -1
  int Din;
  const int32_t* out_coords;  // (n_out, 3)
  int64_t n_out;
  int mode;
  const float* weight;  // (27, CIN, COUT)
  const float* scale;   // (COUT) gamma / sqrt(var + eps)      -- or null: no BN/ReLU
  const float* shift;   // (COUT) beta - mean * scale
  const float* skip;    // (n_out, COUT) added after the ReLU, or null
  float* out;           // (n_out, COUT)
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_kernel(SpconvArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_out) return;
  const int cx = a.out_coords[i * 3 + 0], cy = a.out_coords[i * 3 + 1], cz = a.out_coords[i * 3 + 2];
  const int D = a.Din;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  const float* __restrict__ W = a.weight;

  for (int k = 0; k < 27; ++k) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    int x, y, z;
    bool ok = true;
    if (a.mode == MODE_SUBM) {
      x = cx + ox; y = cy + oy; z = cz + oz;
    } else if (a.mode == MODE_DOWN) {
      x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz;
    } else {  // MODE_UP: coarse site q with 2 q + o == c
      const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
      ok = ((tx | ty | tz) & 1) == 0;
      x = tx >> 1; y = ty >> 1; z = tz >> 1;
    }
    ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
    int row = -1;
    if (ok) row = a.in_table[((int64_t)x * D + y) * D + z];
    if (row < 0) continue;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.in + (int64_t)row * CIN);
    const float* __restrict__ Wk = W + (int64_t)k * CIN * COUT;
    for (int c4 = 0; c4 < CIN / 4; ++c4) {
      const f32x4 xv = src[c4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xq = xv[q];
        const float* __restrict__ Wr = Wk + (c4 * 4 + q) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xq, Wr[co], acc[co]);
      }
    }
  }
  float* __restrict__ dst = a.out + i * COUT;
  const float* __restrict__ sk = a.skip ? a.skip + i * COUT : nullptr;
#pragma unroll
  for (int c4 = 0; c4 < COUT / 4; ++c4) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = c4 * 4 + q;
      float y = acc[co];
      if (a.scale) y = fmaxf(y * a.scale[co] + a.shift[co], 0.f);
      if (sk) y += sk[co];
      v[q] = y;
    }
    reinterpret_cast<f32x4*>(dst)[c4] = v;
  }
}

// Round 5: the same arithmetic (same order of the 27 offsets, same FMA order: bit-identical results) for the THIN channel pairs,
// software-pipelined.  The loop above is a chain of dependent round trips per offset - table entry, then the row it names, then
// the FMAs - with nothing of the next offset in flight, and the thin layers sit on the finest lattices (5 M sites: 8.3 ms of a
// training step, 2.7 ms of a volume build).  Here (a) all 27 table entries are fetched first, unconditionally (an offset outside
// the lattice or of the wrong parity reads entry 0 and is discarded): 27 independent loads; (b) the row of offset k + 1 is
// fetched - again unconditionally, row 0 for an absent neighbour - before the FMAs of offset k are issued.  MODE is a template
// parameter so that the fully unrolled body carries one coordinate rule.
#ifndef SURF_SPCONV_PIPE
#define SURF_SPCONV_PIPE 1
#endif
// R16 (round 6, the bf16 training policy): the gathered rows are read from a bf16 copy (half the bytes per neighbour row) and
// widened in registers; weights, accumulation and output stay fp32.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int CIN, int COUT, int MODE, bool R16 = false>
__global__ __launch_bounds__(256) void spconv_pipe_kernel(SpconvArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_out) return;
  const int cx = a.out_coords[i * 3 + 0], cy = a.out_coords[i * 3 + 1], cz = a.out_coords[i * 3 + 2];
  const int D = a.Din;
  int rows[27];
  // Round 6: the three z-neighbours of an (x, y) column are adjacent entries of the z-fastest table: for the submanifold and
  // stride-2 windows (z - 1, z, z + 1 around bz) they come as ONE 12-byte load per column - 9 requests instead of 27 for the
  // table (the texture path serves distinct cache lines one at a time: the table was half of this kernel's line requests).
  // Same entries, same order of the offsets below: bit-identical results.  (Sites on the lattice's z border take the single loads.)
  // Measured (same box): sparse U-Net of a volume build 7.82 -> 7.57 ms; training step -0.3 ms (ABBA x 2), -0.5 ms with in-order launches.
  bool triple = false;
  if constexpr ((MODE == MODE_SUBM || MODE == MODE_DOWN) && SURF_SPCONV_TRIPLE) {
    const int bx = MODE == MODE_SUBM ? cx : 2 * cx, by = MODE == MODE_SUBM ? cy : 2 * cy, bz = MODE == MODE_SUBM ? cz : 2 * cz;
    triple = bz - 1 >= 0 && bz + 1 < D;
    if (triple) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const I3u t3 = surf_table_column3(a.in_table, D, bx + j % 3 - 1, by + j / 3 - 1, bz);
        rows[j] = t3.a;
        rows[9 + j] = t3.b;
        rows[18 + j] = t3.c;
      }
    }
  }
  if (!triple)
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    int x, y, z;
    bool ok = true;
    if (MODE == MODE_SUBM) {
      x = cx + ox; y = cy + oy; z = cz + oz;
    } else if (MODE == MODE_DOWN) {
      x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz;
    } else {
      const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
      ok = ((tx | ty | tz) & 1) == 0;
      x = tx >> 1; y = ty >> 1; z = tz >> 1;
    }
    ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
    const int r = a.in_table[ok ? ((int64_t)x * D + y) * D + z : 0];
    rows[k] = ok ? r : -1;
  }
  // (measured and dropped: skipping the fetches of an offset that no lane of the wavefront has - a ballot per offset, uniform
  // branches around the loads - cost registers and scalar work: 6.9 instead of 6.2 ms per training step over all thin calls)
  constexpr int V = CIN / 4;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  f32x4 cur[V], nxt[V];
  auto fetch = [&](int row, f32x4 (&v)[V]) {
    if constexpr (R16) {
      const u32x4_t* __restrict__ src = reinterpret_cast<const u32x4_t*>(a.in16 + (int64_t)max(row, 0) * CIN);
#pragma unroll
      for (int c8 = 0; c8 < CIN / 8; ++c8) {
        const u32x4_t u = src[c8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * c8 + (j >> 1)][2 * (j & 1)] = __builtin_bit_cast(float, u[j] << 16);
          v[2 * c8 + (j >> 1)][2 * (j & 1) + 1] = __builtin_bit_cast(float, u[j] & 0xffff0000u);
        }
      }
    } else {
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.in + (int64_t)max(row, 0) * CIN);
#pragma unroll
      for (int c4 = 0; c4 < V; ++c4) v[c4] = src[c4];
    }
  };
  fetch(rows[0], cur);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k + 1 < 27) fetch(rows[k + 1], nxt);
    if (rows[k] >= 0) {
      const float* __restrict__ Wk = a.weight + (int64_t)k * CIN * COUT;
#pragma unroll
      for (int c4 = 0; c4 < V; ++c4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xq = cur[c4][q];
          const float* __restrict__ Wr = Wk + (c4 * 4 + q) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xq, Wr[co], acc[co]);
        }
    }
#pragma unroll
    for (int c4 = 0; c4 < V; ++c4) cur[c4] = nxt[c4];
  }
  float* __restrict__ dst = a.out + i * COUT;
  const float* __restrict__ sk = a.skip ? a.skip + i * COUT : nullptr;
#pragma unroll
  for (int c4 = 0; c4 < COUT / 4; ++c4) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = c4 * 4 + q;
      float y = acc[co];
      if (a.scale) y = fmaxf(y * a.scale[co] + a.shift[co], 0.f);
      if (sk) y += sk[co];
      v[q] = y;
    }
    reinterpret_cast<f32x4*>(dst)[c4] = v;
  }
}

// marks[q] = 1 for every coarse site q such that 2q is within the 3^3 window of an input voxel and inside the
// bounding box of the input coordinates (the output-site rule of a k3/s2 sparse conv, "dilate" in the oracle)
// 'floor' rule: the output sites of a k3/s2 conv are unique(floor(c / 2)) (MinkowskiEngine-style)
__global__ __launch_bounds__(256) void mark_down_sites_floor_kernel(const int32_t* __restrict__ coords, int64_t n, int D2,
                                                                    uint8_t* __restrict__ marks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  marks[((int64_t)(coords[i * 3 + 0] >> 1) * D2 + (coords[i * 3 + 1] >> 1)) * D2 + (coords[i * 3 + 2] >> 1)] = 1;
}

__global__ __launch_bounds__(256) void mark_down_sites_kernel(const int32_t* __restrict__ coords, int64_t n, int D2,
                                                              const int32_t* __restrict__ bbox,
                                                              uint8_t* __restrict__ marks) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 27) return;
  const int lox = bbox[0], loy = bbox[1], loz = bbox[2], hix = bbox[3], hiy = bbox[4], hiz = bbox[5];
  const int64_t i = t / 27;
  const int k = (int)(t % 27);
  const int x = coords[i * 3 + 0] + (k % 3 - 1), y = coords[i * 3 + 1] + ((k / 3) % 3 - 1), z = coords[i * 3 + 2] + (k / 9 - 1);
  if (((x | y | z) & 1) != 0) return;
  if (x < lox || x > hix || y < loy || y > hiy || z < loz || z > hiz) return;
  marks[((int64_t)(x >> 1) * D2 + (y >> 1)) * D2 + (z >> 1)] = 1;
}

// bounding box of integer coordinates: bbox = [min x, y, z, max x, y, z] (wave reduce, then six atomics per wave)
__global__ void bbox_init_kernel(int32_t* __restrict__ bbox) {
  if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
  else if (threadIdx.x < 6) bbox[threadIdx.x] = (int32_t)0x80000000;
}
__global__ __launch_bounds__(256) void bbox_kernel(const int32_t* __restrict__ coords, int64_t n, int32_t* __restrict__ bbox) {
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int v = coords[i * 3 + a];
      lo[a] = min(lo[a], v);
      hi[a] = max(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min_i(lo[a]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hi[a] = max(hi[a], __shfl_xor(hi[a], o));
  }
  // one atomic per workgroup and bound: 24 calls of the per-wavefront form cost 3.7 ms per build in same-address atomics
  __shared__ int part[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      part[threadIdx.x >> 6][a] = lo[a];
      part[threadIdx.x >> 6][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = part[0][a];
#pragma unroll
    for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, part[w][a]) : max(v, part[w][a]);
    if (a < 3) atomicMin(&bbox[a], v);
    else atomicMax(&bbox[a], v);
  }
}

// keys (ascending site numbers of a D^3 lattice) -> coords (n,3) and table[key] = rank
__global__ __launch_bounds__(256) void sites_from_keys_kernel(const int32_t* __restrict__ keys, int64_t n, int D,
                                                              int32_t* __restrict__ coords, int32_t* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int key = keys[i];
  coords[i * 3 + 0] = key / (D * D);
  coords[i * 3 + 1] = (key / D) % D;
  coords[i * 3 + 2] = key % D;
  table[key] = (int32_t)i;
}

__global__ __launch_bounds__(256) void table_from_coords_kernel(const int32_t* __restrict__ coords, int64_t n, int D,
                                                                int32_t* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  table[((int64_t)coords[i * 3 + 0] * D + coords[i * 3 + 1]) * D + coords[i * 3 + 2]] = (int32_t)i;
}

// out[i, :] = in[i, :C] @ W^T   (nn.Linear without bias, reg_network.py:67,86)
template <int C>
__global__ __launch_bounds__(256) void row_linear_kernel(const float* __restrict__ in, const float* __restrict__ W, int64_t n,
                                                         float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = in[i * C + c];
#pragma unroll
  for (int o = 0; o < C; ++o) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) acc = fmaf(x[c], W[o * C + c], acc);
    out[i * C + o] = acc;
  }
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

#define SPCONV_CASE(CI, CO)                                                                                  \
  if (cin == CI && cout == CO) {                                                                             \
    hipLaunchKernelGGL((spconv_kernel<CI, CO>), grid1d(n_out, 256), dim3(256), 0, (hipStream_t)stream, a);   \
    return surf_check_launch();                                                                              \
  }
#define SPCONV_PIPE_MODE(CI, CO, M)                                                                                      \
  hipLaunchKernelGGL((spconv_pipe_kernel<CI, CO, M>), grid1d(n_out, 256), dim3(256), 0, (hipStream_t)stream, a)
#define SPCONV_PIPE_CASE(CI, CO)                                                                             \
  if (SURF_SPCONV_PIPE && cin == CI && cout == CO) {                                                         \
    if (mode == MODE_SUBM) SPCONV_PIPE_MODE(CI, CO, MODE_SUBM);                                              \
    else if (mode == MODE_DOWN) SPCONV_PIPE_MODE(CI, CO, MODE_DOWN);                                         \
    else SPCONV_PIPE_MODE(CI, CO, MODE_UP);                                                                  \
    return surf_check_launch();                                                                              \
  }

extern "C" int surf_spconv(const float* in, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                           int64_t n_out, int mode, const float* weight, int cout, const float* bn_scale,
                           const float* bn_shift, const float* skip, float* out, void* stream) {
  if (!in || !in_table || !out_coords || !weight || !out || n_out <= 0 || D_in < 1) return SURF_E_ARG;
  if (mode < 0 || mode > 2 || ((bn_scale == nullptr) != (bn_shift == nullptr))) return SURF_E_ARG;
  SpconvArgs a;
  a.in = in; a.in16 = nullptr; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode;
  a.weight = weight; a.scale = bn_scale; a.shift = bn_shift; a.skip = skip; a.out = out;
  SPCONV_PIPE_CASE(8, 8) SPCONV_PIPE_CASE(16, 8) SPCONV_PIPE_CASE(8, 16) SPCONV_PIPE_CASE(16, 16)     // the thin pairs, pipelined
  SPCONV_CASE(8, 8) SPCONV_CASE(16, 8) SPCONV_CASE(8, 16) SPCONV_CASE(16, 16) SPCONV_CASE(16, 32) SPCONV_CASE(32, 32)
  SPCONV_CASE(32, 64) SPCONV_CASE(64, 64) SPCONV_CASE(64, 32) SPCONV_CASE(32, 16)
  return SURF_E_LIMIT;  // channel pair not instantiated (reg_network.py uses d_base = 8 only)
}

// The thin pair whose input rows are 16 channels wide, with the gathered rows read from a bf16 copy (the bf16 training policy,
// round 6): 64 -> 32 bytes per neighbour row, fp32 weights / accumulation / output.  Measured on the bench lattices
// (scripts/time_spconv_rows16.py): <16,8> 0.84 -> 0.51, 1.41 -> 0.77, 0.86 -> 0.58 ms; the pairs with 8 input channels gain 5 %
// (<8,16>) and (16,16) 7 % (it runs on the matrix cores anyway): only (16,8) is dispatched.
extern "C" int surf_spconv_rows16(const uint16_t* in16, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                                  int64_t n_out, int mode, const float* weight, int cout, float* out, void* stream) {
  if (!in16 || !in_table || !out_coords || !weight || !out || n_out <= 0 || D_in < 1 || mode < 0 || mode > 2) return SURF_E_ARG;
  if (cin != 16 || cout != 8) return SURF_E_LIMIT;
  SpconvArgs a;
  a.in = nullptr; a.in16 = in16; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode;
  a.weight = weight; a.scale = nullptr; a.shift = nullptr; a.skip = nullptr; a.out = out;
  if (mode == MODE_SUBM)
    hipLaunchKernelGGL((spconv_pipe_kernel<16, 8, MODE_SUBM, true>), grid1d(n_out, 256), dim3(256), 0, (hipStream_t)stream, a);
  else if (mode == MODE_DOWN)
    hipLaunchKernelGGL((spconv_pipe_kernel<16, 8, MODE_DOWN, true>), grid1d(n_out, 256), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((spconv_pipe_kernel<16, 8, MODE_UP, true>), grid1d(n_out, 256), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int surf_coords_bbox(const int32_t* coords, int64_t n, int32_t* bbox, void* stream) {
  if (!coords || !bbox || n <= 0) return SURF_E_ARG;
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bbox);
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, (hipStream_t)stream, coords, n, bbox);
  return surf_check_launch();
}

extern "C" int surf_mark_down_sites(const int32_t* coords, int64_t n, int D, const int32_t* bbox, uint8_t* marks, int rule,
                                    void* stream) {
  if (!coords || !marks || n <= 0 || D < 2) return SURF_E_ARG;
  if (rule != SURF_DOWN_DILATE && rule != SURF_DOWN_FLOOR) return SURF_E_ARG;
  const int D2 = D / 2 + 1;
  if (rule == SURF_DOWN_FLOOR) {
    hipLaunchKernelGGL(mark_down_sites_floor_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, coords, n, D2, marks);
  } else {
    if (!bbox) return SURF_E_ARG;
    hipLaunchKernelGGL(mark_down_sites_kernel, grid1d(n * 27, 256), dim3(256), 0, (hipStream_t)stream, coords, n, D2, bbox, marks);
  }
  return surf_check_launch();
}

extern "C" int surf_sites_from_keys(const int32_t* keys, int64_t n, int D, int32_t* coords, int32_t* table, void* stream) {
  if (!keys || !coords || !table || n <= 0 || D < 1) return SURF_E_ARG;
  hipLaunchKernelGGL(sites_from_keys_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, keys, n, D, coords, table);
  return surf_check_launch();
}

extern "C" int surf_table_from_coords(const int32_t* coords, int64_t n, int D, int32_t* table, void* stream) {
  if (!coords || !table || n <= 0 || D < 1) return SURF_E_ARG;
  hipLaunchKernelGGL(table_from_coords_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, coords, n, D, table);
  return surf_check_launch();
}

extern "C" int surf_row_linear8(const float* in, const float* weight, int64_t n, float* out, void* stream) {
  if (!in || !weight || !out || n <= 0) return SURF_E_ARG;
  hipLaunchKernelGGL(row_linear_kernel<8>, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, in, weight, n, out);
  return surf_check_launch();
}
