// Volume-build kernels: K2 cost volume, K3/K4 upsample + depth-band filter, stream compaction,
// row gathers, K6 densify.
//
// Restates Volume.up_sample / depth_filtering / back_proj_multiscale / sparse2dense / get_index
//          volume.py:35-52, 134-168, 54-97, 99-121, 123-132  and the row selections of surf.py:104-109.
#include <stdlib.h>

#include "common.h"

namespace {

struct ViewSet {
  int nv;
  float w2c[SURF_MAX_VIEWS][12];  // rows 0..2 of inverse(c2w)
  float K[SURF_MAX_VIEWS][12];    // rows 0..2 of the 4x4 intrinsics
};

__constant__ int kChildOff[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};

// voxel -> normalised image coordinates of view v (volume.py:64-77); returns in-frustum flag
__device__ __forceinline__ bool project_voxel(const ViewSet& vs, int v, float wx, float wy, float wz, float half_w,
                                              float half_h, float& nx, float& ny, float& qz) {
  const float* M = vs.w2c[v];
  float X = M[0] * wx + M[1] * wy + M[2] * wz + M[3];
  float Y = M[4] * wx + M[5] * wy + M[6] * wz + M[7];
  float Z = M[8] * wx + M[9] * wy + M[10] * wz + M[11];
  const float* K = vs.K[v];
  float qx = K[0] * X + K[1] * Y + K[2] * Z + K[3];
  float qy = K[4] * X + K[5] * Y + K[6] * Z + K[7];
  qz = K[8] * X + K[9] * Y + K[10] * Z + K[11];
  float x = qx / qz, y = qy / qz;
  nx = x / half_w - 1.0f;
  ny = y / half_h - 1.0f;
  return (fabsf(nx) <= 1.0f) && (fabsf(ny) <= 1.0f) && (qz > 0.0f);
}

// align_corners=True unnormalisation (volume.py:83,159)
__device__ __forceinline__ float unnorm_act(float g, int size) { return ((g + 1.0f) / 2.0f) * (float)(size - 1); }

__device__ __forceinline__ float bilinear_scalar(const float* __restrict__ map, int H, int W, float x, float y) {
  float fx = floorf(x), fy = floorf(y);
  float tx = x - fx, ty = y - fy;
  int x0 = (int)fx, y0 = (int)fy;
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    int yi = y0 + dy;
    float wy = dy ? ty : 1.0f - ty;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      int xi = x0 + dx;
      float wx = dx ? tx : 1.0f - tx;
      if ((xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)) acc += map[(int64_t)yi * W + xi] * (wx * wy);
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------------
// K3/K4: children of every parent voxel tested against the previous stage's depth maps.
// flags[8 p + o] = 1 iff child o of parent p is depth-consistent in more than one view.
// ---------------------------------------------------------------------------------------------------------
struct FilterArgs {
  const int32_t* parents;  // (n_par,3) coords on the D/2 lattice
  int64_t n_par;
  int D;                   // child lattice side
  float voxel_size;
  const float* depths;     // (nv,H,W)
  int H, W;
  float depth_range;
  ViewSet vs;
  uint8_t* flags;
};

__global__ __launch_bounds__(256) void upsample_filter_kernel(FilterArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_par * 8) return;
  const int64_t p = i >> 3;
  const int o = (int)(i & 7);
  const float cx = (float)(2 * a.parents[p * 3 + 0] + kChildOff[o][0]);
  const float cy = (float)(2 * a.parents[p * 3 + 1] + kChildOff[o][1]);
  const float cz = (float)(2 * a.parents[p * 3 + 2] + kChildOff[o][2]);
  const float wx = cx * a.voxel_size + (-1.0f), wy = cy * a.voxel_size + (-1.0f), wz = cz * a.voxel_size + (-1.0f);
  const float half_w = (float)(a.W - 1) / 2.0f, half_h = (float)(a.H - 1) / 2.0f;
  int cnt = 0;
  for (int v = 0; v < a.vs.nv; ++v) {
    float nx, ny, qz;
    bool m = project_voxel(a.vs, v, wx, wy, wz, half_w, half_h, nx, ny, qz);
    float d = bilinear_scalar(a.depths + (int64_t)v * a.H * a.W, a.H, a.W, unnorm_act(nx, a.W), unnorm_act(ny, a.H));
    cnt += ((fabsf(d - qz) < a.depth_range) && m) ? 1 : 0;
  }
  a.flags[i] = cnt > 1 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// K2: homography warp + softmax-weighted mean / variance cost volume for a list of voxels.
// Voxel i is either (mode 0) the i-th site of the full D^3 lattice, x slowest (volume.py:21-33), or
// (mode 1) child (idx[i] & 7) of parent (idx[i] >> 3).  Writes coords (n,3), feat (n,8), keep (n).
// ---------------------------------------------------------------------------------------------------------
struct CostVolArgs {
  const int32_t* parents;
  const int32_t* idx;
  int64_t n;
  int D;
  float voxel_size;
  const float* feats[4];  // texel4 pyramids coarse -> fine
  int hw[8];
  int stage;              // levels stage..3 are summed (volume.py:82)
  ViewSet vs;
  float w1[32], b1[8], w2[8], b2;  // agg_mlp: Linear(4,8), ELU, Linear(8,1)
  int32_t* coords;
  float* feat;
  uint8_t* keep;
};

__global__ __launch_bounds__(256) void costvol_kernel(CostVolArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int cx, cy, cz;
  if (a.idx) {
    const int64_t j = a.idx[i];
    const int64_t p = j >> 3;
    const int o = (int)(j & 7);
    cx = 2 * a.parents[p * 3 + 0] + kChildOff[o][0];
    cy = 2 * a.parents[p * 3 + 1] + kChildOff[o][1];
    cz = 2 * a.parents[p * 3 + 2] + kChildOff[o][2];
  } else {
    cz = (int)(i % a.D);
    cy = (int)((i / a.D) % a.D);
    cx = (int)(i / ((int64_t)a.D * a.D));
  }
  a.coords[i * 3 + 0] = cx; a.coords[i * 3 + 1] = cy; a.coords[i * 3 + 2] = cz;
  const float wx = (float)cx * a.voxel_size + (-1.0f), wy = (float)cy * a.voxel_size + (-1.0f),
              wz = (float)cz * a.voxel_size + (-1.0f);
  const int Hf = a.hw[6], Wf = a.hw[7];  // finest level sets the normalisation (volume.py:62,72-73)
  const float half_w = (float)(Wf - 1) / 2.0f, half_h = (float)(Hf - 1) / 2.0f;

  float f[SURF_MAX_VIEWS][4];
  float logit[SURF_MAX_VIEWS];
  int cnt = 0;
  float mx = -INFINITY;
#pragma unroll
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    f[v][0] = f[v][1] = f[v][2] = f[v][3] = 0.f;
    logit[v] = -INFINITY;
    if (v < a.vs.nv) {
      float nx, ny, qz;
      bool m = project_voxel(a.vs, v, wx, wy, wz, half_w, half_h, nx, ny, qz);
      cnt += m ? 1 : 0;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int l = a.stage; l < 4; ++l) {
        const int H = a.hw[2 * l], W = a.hw[2 * l + 1];
        acc += bilinear_texel4(a.feats[l] + (int64_t)v * H * W * 4, H, W, unnorm_act(nx, W), unnorm_act(ny, H));
      }
      f[v][0] = acc[0]; f[v][1] = acc[1]; f[v][2] = acc[2]; f[v][3] = acc[3];
      float s = a.b2;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float h = a.b1[o];
#pragma unroll
        for (int c = 0; c < 4; ++c) h += a.w1[o * 4 + c] * f[v][c];
        h = h > 0.f ? h : expm1f(h);
        s += a.w2[o] * h;
      }
      logit[v] = m ? s : -1e9f;
      mx = fmaxf(mx, logit[v]);
    }
  }
  float den = 0.f;
  float e[SURF_MAX_VIEWS];
#pragma unroll
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    e[v] = (v < a.vs.nv) ? expf(logit[v] - mx) : 0.f;
    den += e[v];
  }
  float mean[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    if (v < a.vs.nv) {
      const float w = e[v] / den;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float wf = f[v][c] * w;
        mean[c] += wf;
        sq[c] += wf * wf;
      }
    }
  }
  f32x4 o0 = {mean[0], mean[1], mean[2], mean[3]};
  f32x4 o1 = {sq[0] - mean[0] * mean[0], sq[1] - mean[1] * mean[1], sq[2] - mean[2] * mean[2], sq[3] - mean[3] * mean[3]};
  *reinterpret_cast<f32x4*>(a.feat + i * 8) = o0;
  *reinterpret_cast<f32x4*>(a.feat + i * 8 + 4) = o1;
  a.keep[i] = cnt > 1 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// Stable stream compaction of a byte flag array: idx_out = ascending list of i with flags[i] != 0.
// Three launches: per-block counts, single-block scan of the counts, scatter.
// ---------------------------------------------------------------------------------------------------------
constexpr int CP_ITEMS = 16, CP_THREADS = 256, CP_BLOCK = CP_ITEMS * CP_THREADS;

// The 16 flags of a thread as a 16-bit mask.  Round 6: one 16-byte load where the thread's slice lies inside the array (the flag
// arrays are whole allocations: 16-byte aligned, checked at launch) instead of 16 byte loads with 16 bound checks - the dense mark
// arrays of the stride-2 site lists are up to 44 MB and were read at a fraction of the memory rate (1.8 ms of a 16 ms volume build).
static_assert(CP_ITEMS == 16, "one uint4 per thread");
__device__ __forceinline__ unsigned cp_mask16(const uint8_t* __restrict__ flags, int64_t base, int64_t n, bool aligned) {
  unsigned bits = 0;
  if (aligned && base + CP_ITEMS <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(flags + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) bits |= (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) ? 1u : 0u) << k;
  } else {
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) bits |= ((base + k < n && flags[base + k]) ? 1u : 0u) << k;
  }
  return bits;
}

__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(const uint8_t* __restrict__ flags, int64_t n,
                                                                   int32_t* __restrict__ block_counts, bool aligned) {
  __shared__ int s_part[CP_THREADS / 64];
  const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
  int c = __popc(cp_mask16(flags, base, n, aligned));
  c = (int)wave_sum((float)c);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(int32_t* __restrict__ block_counts, int nb,
                                                            int32_t* __restrict__ total) {
  // exclusive scan in place by one block, chunks of 1024
  __shared__ int s_buf[1024];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nb; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int v = i < nb ? block_counts[i] : 0;
    s_buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = threadIdx.x >= o ? s_buf[threadIdx.x - o] : 0;
      __syncthreads();
      s_buf[threadIdx.x] += t;
      __syncthreads();
    }
    const int incl = s_buf[threadIdx.x];
    const int carry = s_carry;
    if (i < nb) block_counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}

__global__ __launch_bounds__(CP_THREADS) void compact_scatter_kernel(const uint8_t* __restrict__ flags, int64_t n,
                                                                     const int32_t* __restrict__ block_offsets,
                                                                     int32_t* __restrict__ idx_out, bool aligned) {
  __shared__ int s_part[CP_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
  const unsigned bits = cp_mask16(flags, base, n, aligned);
  const int c = __popc(bits);
  // exclusive scan of c across the wave, then across the 4 waves
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += s_part[w];
  int pos = block_offsets[blockIdx.x] + wave_off + incl - c;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k)
    if (bits & (1u << k)) idx_out[pos++] = (int32_t)(base + k);
}

// dst[i, 0:w] = src[idx[i] >> shift, 0:w]   (rows of w 32-bit words; shift = 3 selects the parent row)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                          int64_t n, int w, int shift, int dst_stride, int dst_off,
                                                          uint32_t* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * w) return;
  const int64_t i = t / w;
  const int c = (int)(t % w);
  dst[i * dst_stride + dst_off + c] = src[(int64_t)(idx[i] >> shift) * w + c];
}

// idx_out[i] = a[b[i]]  (composition of two index lists)
__global__ __launch_bounds__(256) void compose_index_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                            int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[b[i]];
}

// ---------------------------------------------------------------------------------------------------------
// K6 densify: background = x2 trilinear upsample of the previous matching volume (align_corners=False) or 0,
// index table = -1; then scatter the stage's logits and row numbers.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int i, int Dp, int& i0, int& i1, float& l1) {
  float src = ((float)i + 0.5f) * 0.5f - 0.5f;  // area_pixel_compute_source_index, scale 0.5
  if (src < 0.f) src = 0.f;
  i0 = (int)floorf(src);
  i1 = min(i0 + 1, Dp - 1);
  l1 = src - (float)i0;
}

// VEC voxels along z per vector (VEC = 4 when D % 4 == 0: 16-byte stores; 704^3 is 1.4 GB of logits + 1.4 GB of table), NV vectors
// per thread at a stride of the workgroup (round 5: 4 - fewer, longer workgroups for the 2.8 GB of stores)
template <int VEC, int NV>
__global__ __launch_bounds__(256) void dense_init_kernel(const float* __restrict__ prev, int D, float* __restrict__ dense,
                                                         int32_t* __restrict__ table) {
  const int64_t total = (int64_t)D * D * D;
  const int64_t t0 = (int64_t)blockIdx.x * (256 * NV) + threadIdx.x;
  // one 32-bit decomposition per vector (64-bit div / mod per voxel cost more than the whole upsample)
  const unsigned zc = (unsigned)D / VEC;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int64_t t = t0 + 256 * k, i0 = t * VEC;
    if (i0 >= total) return;
    const unsigned tu = (unsigned)t;
    const int zb = (int)(tu % zc) * VEC, y = (int)((tu / zc) % (unsigned)D), x = (int)(tu / (zc * (unsigned)D));
    float out[VEC];
    if (VEC == 4 && prev) {
      // the four voxels z = zb .. zb + 3 (zb a multiple of 4) read the coarse planes c0 .. c0 + 3, c0 = zb / 2 - 1, as the pairs
      // (0,1) (1,2) (1,2) (2,3): 16 loads for the vector instead of 32 (same expression, same order: bit-identical; at the two
      // ends the clamped plane carries weight 0 or coincides with up2_src's own clamp)
      const int Dp = D / 2;
      int x0, x1, y0, y1;
      float lx, ly;
      up2_src(x, Dp, x0, x1, lx);
      up2_src(y, Dp, y0, y1, ly);
      const float hx = 1.0f - lx, hy = 1.0f - ly;
      const int c0 = zb / 2 - 1;
      float P[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int zc = min(max(c0 + t, 0), Dp - 1);
        P[0][t] = prev[((int64_t)x0 * Dp + y0) * Dp + zc];
        P[1][t] = prev[((int64_t)x0 * Dp + y1) * Dp + zc];
        P[2][t] = prev[((int64_t)x1 * Dp + y0) * Dp + zc];
        P[3][t] = prev[((int64_t)x1 * Dp + y1) * Dp + zc];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int z0, z1;
        float lz;
        up2_src(zb + q, Dp, z0, z1, lz);
        const float hz = 1.0f - lz;
        const int a_ = q == 0 ? 0 : (q == 3 ? 2 : 1);
        out[q] = hx * (hy * (hz * P[0][a_] + lz * P[0][a_ + 1]) + ly * (hz * P[1][a_] + lz * P[1][a_ + 1])) +
                 lx * (hy * (hz * P[2][a_] + lz * P[2][a_ + 1]) + ly * (hz * P[3][a_] + lz * P[3][a_ + 1]));
      }
    } else
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float v = 0.f;
      if (prev) {
        const int Dp = D / 2;
        const int z = zb + q;
        int x0, x1, y0, y1, z0, z1;
        float lx, ly, lz;
        up2_src(x, Dp, x0, x1, lx);
        up2_src(y, Dp, y0, y1, ly);
        up2_src(z, Dp, z0, z1, lz);
        const float hx = 1.0f - lx, hy = 1.0f - ly, hz = 1.0f - lz;
        auto P = [&](int xi, int yi, int zi) { return prev[((int64_t)xi * Dp + yi) * Dp + zi]; };
        // ATen upsample_trilinear3d: w_d (w_h (w_w a + w_w b) + ...) with d = our x, h = y, w = z
        v = hx * (hy * (hz * P(x0, y0, z0) + lz * P(x0, y0, z1)) + ly * (hz * P(x0, y1, z0) + lz * P(x0, y1, z1))) +
            lx * (hy * (hz * P(x1, y0, z0) + lz * P(x1, y0, z1)) + ly * (hz * P(x1, y1, z0) + lz * P(x1, y1, z1)));
      }
      out[q] = v;
    }
    if (VEC == 4) {
      // streaming stores (2.8 GB at 704^3 that nothing re-reads from L2): 1.54 -> 1.47 ms
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f32x4{out[0], out[1 % VEC], out[2 % VEC], out[3 % VEC]}, reinterpret_cast<f32x4*>(dense + i0));
      __builtin_nontemporal_store(i32x4{-1, -1, -1, -1}, reinterpret_cast<i32x4*>(table + i0));
    } else {
      dense[i0] = out[0];
      table[i0] = -1;
    }
  }
}

__global__ __launch_bounds__(256) void dense_scatter_kernel(const int32_t* __restrict__ coords, const float* __restrict__ rows,
                                                            int row_stride, int64_t n, int D, float* __restrict__ dense,
                                                            int32_t* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t o = ((int64_t)coords[i * 3 + 0] * D + coords[i * 3 + 1]) * D + coords[i * 3 + 2];
  dense[o] = rows[i * row_stride];
  table[o] = (int32_t)i;
}


// ---------------------------------------------------------------------------------------------------------
// Backward kernels of the volume build (train mode; surf.py:80-131 under loss.backward(), runner.py:163).
// ---------------------------------------------------------------------------------------------------------
// K6 backward: the scattered sites take their gradient from the dense one, the background passes the rest to the previous
// stage's matching volume through the transposed x2 trilinear upsample (sites overwritten by the scatter pass nothing).
__global__ __launch_bounds__(256) void dense_rows_bwd_kernel(const int32_t* __restrict__ coords, const float* __restrict__ g_dense,
                                                             int64_t n, int D, int row_stride, float* __restrict__ g_rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t o = ((int64_t)coords[i * 3 + 0] * D + coords[i * 3 + 1]) * D + coords[i * 3 + 2];
  g_rows[i * row_stride] += g_dense[o];
}

// VEC voxels along z per thread (VEC = 4: 16-byte loads of the mostly-zero dense gradient first, the table only where something
// is non-zero - the 704^3 sweep is otherwise 2.8 GB of traffic and 1.4 M workgroups).
// Round 5: NV vectors per thread, all loaded before the first is looked at (a thread that loads one 16-byte vector and returns
// made the 704^3 sweep 1.4 M workgroups of a few hundred cycles each: 0.58 TB/s, bound by the workgroup dispatch rate).
template <int VEC, int NV>
__global__ __launch_bounds__(256) void dense_init_bwd_kernel(const float* __restrict__ g_dense, const int32_t* __restrict__ table,
                                                             int D, float* __restrict__ g_prev) {
  const int64_t total = (int64_t)D * D * D;
  const int64_t t0 = (int64_t)blockIdx.x * (256 * NV) + threadIdx.x;
  float gv[NV][VEC];
  bool any = false;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int64_t i0 = (t0 + 256 * k) * VEC;
#pragma unroll
    for (int q = 0; q < VEC; ++q) gv[k][q] = 0.f;
    if (i0 >= total) continue;
    if (VEC == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g_dense + i0);
      gv[k][0] = v[0]; gv[k][1 % VEC] = v[1]; gv[k][2 % VEC] = v[2]; gv[k][3 % VEC] = v[3];
    } else {
      gv[k][0] = g_dense[i0];
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) any = any || gv[k][q] != 0.f;
  }
  if (!any) return;
  const int Dp = D / 2;
  const unsigned zc = (unsigned)D / VEC;
#pragma unroll 1
  for (int k = 0; k < NV; ++k) {
    const int64_t t = t0 + 256 * k, i0 = t * VEC;
    if (i0 >= total) break;
    bool nz = false;
#pragma unroll
    for (int q = 0; q < VEC; ++q) nz = nz || gv[k][q] != 0.f;
    if (!nz) continue;
    const unsigned tu = (unsigned)t;
    const int zb = (int)(tu % zc) * VEC, y = (int)((tu / zc) % (unsigned)D), x = (int)(tu / (zc * (unsigned)D));
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const int64_t i = i0 + q;
      const float g = gv[k][q];
      if (g == 0.f || table[i] >= 0) continue;
      const int z = zb + q;
      int x0, x1, y0, y1, z0, z1;
      float lx, ly, lz;
      up2_src(x, Dp, x0, x1, lx);
      up2_src(y, Dp, y0, y1, ly);
      up2_src(z, Dp, z0, z1, lz);
      const float hx = 1.0f - lx, hy = 1.0f - ly, hz = 1.0f - lz;
      auto A = [&](int xi, int yi, int zi, float wgt) {
        if (wgt != 0.f) atomicAdd(g_prev + ((int64_t)xi * Dp + yi) * Dp + zi, g * wgt);
      };
      A(x0, y0, z0, hx * hy * hz); A(x0, y0, z1, hx * hy * lz); A(x0, y1, z0, hx * ly * hz); A(x0, y1, z1, hx * ly * lz);
      A(x1, y0, z0, lx * hy * hz); A(x1, y0, z1, lx * hy * lz); A(x1, y1, z0, lx * ly * hz); A(x1, y1, z1, lx * ly * lz);
    }
  }
}

// Round 5 (second pass): the same transposed x2 upsample in GATHER form, no atomics.  Along one axis the fine voxel i hands
// 0.75 / 0.25 of its gradient to the coarse cells (i - 1) / 2 and (i + 1) / 2 (align_corners=False, scale 0.5: src = i / 2 - 0.25),
// so coarse cell c collects from the fine voxels 2c - 1 .. 2c + 2 with the weights {0.25, 0.75, 0.75, 0.25}; at the two ends the
// clamped source index gives fine voxel 0 and fine voxel D - 1 entirely (weight 1) to cells 0 and Dp - 1.  A workgroup owns a
// 4 x 4 x 32 tile of coarse cells, stages the masked fine gradient of its (10 x 10 x 66)-voxel footprint in LDS (background
// voxels only: table < 0, looked up only where the gradient is non-zero), leaves if all of it is zero - the dense gradient is
// mostly zero - and otherwise every thread sums the 4 x 4 x 4 windows of its two cells.  The scatter form above paid eight
// float atomics per non-zero fine voxel (2.3 ms for the 704^3 stage).
constexpr int DG_TX = 4, DG_TY = 4, DG_TZ = 32;
constexpr int DG_FX = 2 * DG_TX + 2, DG_FY = 2 * DG_TY + 2, DG_FZ = 2 * DG_TZ + 2;

__global__ __launch_bounds__(256) void dense_init_bwd_gather_kernel(const float* __restrict__ g_dense, const int32_t* __restrict__ table,
                                                                    int D, float* __restrict__ g_prev) {
  __shared__ float tile[DG_FX * DG_FY * DG_FZ];
  const int Dp = D / 2;
  const int tz = (Dp + DG_TZ - 1) / DG_TZ, ty = (Dp + DG_TY - 1) / DG_TY;
  const int bz = blockIdx.x % tz, by = (blockIdx.x / tz) % ty, bx = blockIdx.x / (tz * ty);
  const int cx0 = bx * DG_TX, cy0 = by * DG_TY, cz0 = bz * DG_TZ;
  const int fx0 = 2 * cx0 - 1, fy0 = 2 * cy0 - 1, fz0 = 2 * cz0 - 1;
  bool nz = false;
  for (int e = threadIdx.x; e < DG_FX * DG_FY * DG_FZ; e += 256) {
    const int lz = e % DG_FZ, ly = (e / DG_FZ) % DG_FY, lx = e / (DG_FZ * DG_FY);
    const int x = fx0 + lx, y = fy0 + ly, z = fz0 + lz;
    float g = 0.f;
    if ((x >= 0) & (x < D) & (y >= 0) & (y < D) & (z >= 0) & (z < D)) {
      const int64_t i = ((int64_t)x * D + y) * D + z;
      g = g_dense[i];
      if (g != 0.f && table[i] >= 0) g = 0.f;       // a scattered site: its gradient went to the stage's rows
    }
    tile[e] = g;
    nz = nz || g != 0.f;
  }
  if (!__syncthreads_or(nz ? 1 : 0)) return;
#pragma unroll
  for (int h = 0; h < (DG_TX * DG_TY * DG_TZ) / 256; ++h) {
    const int v = threadIdx.x + 256 * h;
    const int lz = v % DG_TZ, ly = (v / DG_TZ) % DG_TY, lx = v / (DG_TZ * DG_TY);
    const int cx = cx0 + lx, cy = cy0 + ly, cz = cz0 + lz;
    if (cx >= Dp || cy >= Dp || cz >= Dp) continue;
    float wx[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wz[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (cx == 0) wx[1] = 1.0f;
    if (cx == Dp - 1) wx[2] = 1.0f;
    if (cy == 0) wy[1] = 1.0f;
    if (cy == Dp - 1) wy[2] = 1.0f;
    if (cz == 0) wz[1] = 1.0f;
    if (cz == Dp - 1) wz[2] = 1.0f;
    float acc = 0.f;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      float ax = 0.f;
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) {
        const float* row = tile + ((2 * lx + jx) * DG_FY + (2 * ly + jy)) * DG_FZ + 2 * lz;
        ax += wy[jy] * (wz[0] * row[0] + wz[1] * row[1] + wz[2] * row[2] + wz[3] * row[3]);
      }
      acc += wx[jx] * ax;
    }
    if (acc != 0.f) g_prev[((int64_t)cx * Dp + cy) * Dp + cz] += acc;      // this thread owns the cell: no atomic
  }
}

// backward of gather_rows: g_src[idx[i] >> shift, 0:w] += g_dst[i, off : off + w]
__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const float* __restrict__ g_dst, const int32_t* __restrict__ idx,
                                                               int64_t n, int w, int shift, int dst_stride, int dst_off,
                                                               float* __restrict__ g_src) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * w) return;
  const int64_t i = t / w;
  const int c = (int)(t % w);
  atomicAdd(g_src + (int64_t)(idx[i] >> shift) * w + c, g_dst[i * dst_stride + dst_off + c]);
}

// Octet-cooperative scatter of a texel4 gradient through the bilinear taps, called by ALL 64 lanes of a wavefront: in round r the eight lanes of an
// octet serve lane 8 o + r's request, lane j adding channel j & 3 of the tap column j >> 2 - the two x-taps of a row are
// adjacent texels, so one instruction writes 32 contiguous bytes per request instead of eight separate float atomics (the L2
// atomic rate is per memory transaction, not per float: this is what bounds the kernel).
__device__ __forceinline__ void bilinear_texel4_scatter_coop(float* __restrict__ map, int H, int W, float x, float y, const float g[4],
                                                             bool act) {
  const int lane = threadIdx.x & 63, j = lane & 7, c = j & 3, dx = j >> 2;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int src = (lane & ~7) | r;
    const float sx = __shfl(x, src), sy = __shfl(y, src);
    const bool on = __shfl(act ? 1 : 0, src) != 0;
    const float g0 = __shfl(g[0], src), g1 = __shfl(g[1], src), g2 = __shfl(g[2], src), g3 = __shfl(g[3], src);
    const float gc = c == 0 ? g0 : (c == 1 ? g1 : (c == 2 ? g2 : g3));
    if (on && gc != 0.f) {
      const float fx = floorf(sx), fy = floorf(sy);
      const float tx = sx - fx, ty = sy - fy;
      const int xi = (int)fx + dx, y0 = (int)fy;
      const float wxg = (dx ? tx : 1.0f - tx) * gc;
      if ((xi >= 0) & (xi < W)) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int yi = y0 + dy;
          const float v = wxg * (dy ? ty : 1.0f - ty);
#ifndef SURF_X_CV_NOATOMIC   // timing experiment only (wrong results): how much of costvol_bwd is the scatter
          if ((yi >= 0) & (yi < H) && v != 0.f) atomicAdd(map + ((int64_t)yi * W + xi) * 4 + c, v);
#else
          if ((yi >= 0) & (yi < H) && v == 12345.f) map[0] = v;
#endif
        }
      }
    }
  }
}

// Round 5, measured and removed - three ways of sending fewer texel adds to memory, all at 12.9-13.2 ms per step against 12.94 for
// the octet-cooperative scatter above (with the scatter compiled out the kernel takes 2.9 ms):
//  (i)   one view's adds summed in an LDS hash per 256-voxel workgroup (key (level, y, x), ds_add_f32, one 16-byte flush per
//        distinct texel): 13.15 ms - the voxels that share a texel sit in the same wavefront and serialise on their LDS bank;
//  (ii)  adjacent lanes of an octet that hit the same 2 x 2 block summed by a segmented shuffle scan, only run heads served: 12.96;
//  (iii) the same with the heads served by rank, so that the atomic wave-instructions drop with the merge: 12.94.
//  (iv)  round r serving the eight ADJACENT voxels 8 r .. 8 r + 7 (one per octet) instead of every octet its own r-th voxel, so
//        that the eight 32-byte tap rows of an instruction fall into few 64-byte segments: 12.90;
//  (v)   16 replicas of the gradient maps (workgroup b adds into replica b % 16), summed afterwards: 12.48 - it is not
//        cross-workgroup contention on the small coarse-level maps either.
// scripts/microbench/atomic_shapes.hip (profiles/r05_microbench_atomic_shapes.txt) prices a float atomic at ~12.2 ns per CU
// for every distinct 64-byte segment its lanes touch; 100 M (voxel, view, level) units x 2 tap rows at that price are 9.6 of
// the kernel's 12.9 ms, and none of (i)-(v) made the segments fewer in a way the memory side noticed (neighbouring voxels'
// footprints OVERLAP, and same-address adds inside one instruction serialise).  What did: sorting the adds by image tile - the
// "binned scatter" further down, which is what runs; this direct form remains for shapes the binned one does not take.
constexpr int CV_REPLICAS = 64;

struct CostVolBwdArgs {
  const int32_t* coords;
  const float* g;         // (n, 8) = [d mean | d var]
  int64_t n;
  float voxel_size;
  const float* feats[4];
  float* gfeats[4];
  int hw[8];
  int stage;
  ViewSet vs;
  float w1[32], b1[8], w2[8], b2;
  float* gagg;            // CV_REPLICAS x 64 floats (49 used per replica)
  float* units;           // binned form (below): (nv, n, 8) = [d feature (4) | nx, ny | active | 0] per (view, voxel), or null
  unsigned* gmax;         // binned form: bits of max |d feature| over the call (atomicMax on the bit patterns of non-negative floats)
};

// (round 5: compiled for three wavefronts per SIMD - 168 registers, 20 bytes spilled - instead of the 171 the allocator takes by
// itself, one over the three-wave limit: 3.23 -> 3.16 ms per launch)
#ifndef SURF_CVB_WAVES
#define SURF_CVB_WAVES 3
#endif
__global__ __launch_bounds__(256, SURF_CVB_WAVES) void costvol_bwd_kernel(CostVolBwdArgs a) {
  // voxels in lattice order: neighbouring threads hit neighbouring texels (a strided order that spreads the atomics over the
  // maps measured 30 % slower - the kernel is bound by the locality of its gathers and atomics, not by same-line contention)
  const int64_t i_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i_ < a.n;
  const int64_t i = live ? i_ : a.n - 1;        // every lane runs the whole body (the scatter below is octet-cooperative)
  float gacc[49];
  float dfmax = 0.f;
#pragma unroll
  for (int k = 0; k < 49; ++k) gacc[k] = 0.f;
  {
    const float wx = (float)a.coords[i * 3 + 0] * a.voxel_size + (-1.0f), wy = (float)a.coords[i * 3 + 1] * a.voxel_size + (-1.0f),
                wz = (float)a.coords[i * 3 + 2] * a.voxel_size + (-1.0f);
    const int Hf = a.hw[6], Wf = a.hw[7];
    const float half_w = (float)(Wf - 1) / 2.0f, half_h = (float)(Hf - 1) / 2.0f;
    float f[SURF_MAX_VIEWS][4], logit[SURF_MAX_VIEWS], nxv[SURF_MAX_VIEWS], nyv[SURF_MAX_VIEWS];
    bool inside[SURF_MAX_VIEWS];
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
      f[v][0] = f[v][1] = f[v][2] = f[v][3] = 0.f;
      logit[v] = -INFINITY;
      inside[v] = false;
      nxv[v] = nyv[v] = 0.f;
      if (v < a.vs.nv) {
        float qz;
        inside[v] = project_voxel(a.vs, v, wx, wy, wz, half_w, half_h, nxv[v], nyv[v], qz);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int l = a.stage; l < 4; ++l) {
          const int H = a.hw[2 * l], W = a.hw[2 * l + 1];
          acc += bilinear_texel4(a.feats[l] + (int64_t)v * H * W * 4, H, W, unnorm_act(nxv[v], W), unnorm_act(nyv[v], H));
        }
        f[v][0] = acc[0]; f[v][1] = acc[1]; f[v][2] = acc[2]; f[v][3] = acc[3];
        float s = a.b2;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          float h = a.b1[o];
#pragma unroll
          for (int c = 0; c < 4; ++c) h += a.w1[o * 4 + c] * f[v][c];
          h = h > 0.f ? h : expm1f(h);
          s += a.w2[o] * h;
        }
        logit[v] = inside[v] ? s : -1e9f;
        mx = fmaxf(mx, logit[v]);
      }
    }
    float den = 0.f, wv[SURF_MAX_VIEWS];
#pragma unroll
    for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
      wv[v] = (v < a.vs.nv) ? expf(logit[v] - mx) : 0.f;
      den += wv[v];
    }
    float mean[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
      wv[v] /= den;
#pragma unroll
      for (int c = 0; c < 4; ++c) mean[c] += f[v][c] * wv[v];
    }
    float gm[4], gv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { gm[c] = a.g[i * 8 + c]; gv[c] = a.g[i * 8 + 4 + c]; }
    float dw[SURF_MAX_VIEWS], dwf[SURF_MAX_VIEWS][4], dot = 0.f;
#pragma unroll
    for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
      dw[v] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dwf[v][c] = gm[c] + 2.0f * gv[c] * (f[v][c] * wv[v] - mean[c]);
        dw[v] += dwf[v][c] * f[v][c];
      }
      dot += wv[v] * dw[v];
    }
#pragma unroll
    for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
      if (v < a.vs.nv) {
        float df[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) df[c] = dwf[v][c] * wv[v];
        const float ds = (inside[v] && live) ? wv[v] * (dw[v] - dot) : 0.f;
        if (ds != 0.f) {
          gacc[48] += ds;
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            float h = a.b1[o];
#pragma unroll
            for (int c = 0; c < 4; ++c) h += a.w1[o * 4 + c] * f[v][c];
            const float act = h > 0.f ? h : expm1f(h);
            const float dh = ds * a.w2[o] * (h > 0.f ? 1.0f : expf(h));
            gacc[40 + o] += ds * act;
            gacc[32 + o] += dh;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              gacc[o * 4 + c] += dh * f[v][c];
              df[c] += dh * a.w1[o * 4 + c];
            }
          }
        }
        const bool act = live && !(df[0] == 0.f && df[1] == 0.f && df[2] == 0.f && df[3] == 0.f);   // views outside the frustum: 0
        if (a.units) {                      // binned form: the scatter happens per image tile in costvol_tile_kernel
          if (act) dfmax = fmaxf(dfmax, fmaxf(fmaxf(fabsf(df[0]), fabsf(df[1])), fmaxf(fabsf(df[2]), fabsf(df[3]))));
          if (live) {
            f32x4* u = reinterpret_cast<f32x4*>(a.units + ((int64_t)v * a.n + i) * 8);
            u[0] = f32x4{df[0], df[1], df[2], df[3]};
            u[1] = f32x4{nxv[v], nyv[v], act ? 1.f : 0.f, 0.f};
          }
        } else {
          for (int l = a.stage; l < 4; ++l) {
            const int H = a.hw[2 * l], W = a.hw[2 * l + 1];
            bilinear_texel4_scatter_coop(a.gfeats[l] + (int64_t)v * H * W * 4, H, W, unnorm_act(nxv[v], W), unnorm_act(nyv[v], H), df, act);
          }
        }
      }
    }
  }
  // agg_mlp gradients: wavefront sums -> LDS -> one set of 49 atomics per workgroup into one of CV_REPLICAS replicas (every
  // workgroup adding to the same 49 floats serialises at the memory side: ~12 ns per add and line, 10+ ms at 5 M voxels)
  if (a.gmax) {
    dfmax = wave_max(dfmax);
    if ((threadIdx.x & 63) == 0 && dfmax > 0.f) atomicMax(a.gmax, __float_as_uint(dfmax));
  }
  __shared__ float red[4][49];
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    const float t = wave_sum(gacc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 49) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (t != 0.f) atomicAdd(a.gagg + (blockIdx.x % CV_REPLICAS) * 64 + threadIdx.x, t);
  }
}

__global__ void costvol_bwd_finalize_kernel(const float* __restrict__ replicas, float* __restrict__ g_agg) {
  const int k = threadIdx.x;
  if (k >= 49) return;
  float t = 0.f;
  for (int r = 0; r < CV_REPLICAS; ++r) t += replicas[r * 64 + k];
  g_agg[k] += t;
}

// ---- binned scatter of costvol_bwd (round 5, third pass) ------------------------------------------------------------------------
// The float atomics of the direct scatter are served at the memory side at ~12 ns per CU and 64-byte segment, and no variant
// that merged requests inside a wavefront moved the kernel (see above): the voxels that share a texel are the ones along a
// viewing ray, far apart in lattice order.  So the adds are sorted COARSELY instead: costvol_bwd_kernel leaves one 32-byte
// record per (view, voxel) - the feature gradient and the normalised image position - a counting sort (LDS histograms, one
// reservation per workgroup and bucket) lists the records of every 16 x 16-pixel tile (CVT) of every view, and one workgroup per
// (view, tile) accumulates its records into LDS images of the tile at the pyramid levels the stage reads and sends each touched
// texel to memory ONCE.  Requests per step: 200 M 32-byte adds -> ~15 M coalesced ones.
// The LDS images are 64-bit FIXED POINT: ds_add_f32 costs 81 ns per wave-instruction and CU on gfx950, ds_add_u64 5.5
// (scripts/microbench/lds_atomic_rates.hip, profiles/r05_microbench_lds_atomic_rates.txt; with float LDS atomics this kernel
// took 8.3 ms of a training step, 7.2 of them the atomics).  The scale is a power of two chosen per workgroup from the largest
// |d feature| of the call (costvol_bwd_kernel leaves it in `gmax`) and the number of records the workgroup adds, so that the sum
// cannot overflow and one unit is <= 2^-40 of the largest term: finer than fp32 accumulation, and order-independent.
constexpr int CVT = 16;                                   // tile edge in finest-level pixels
constexpr int CVT_E[4] = {6, 8, 12, 20};                  // LDS image edge per level (coarse -> fine): (16 >> (3 - l)) + margins
constexpr int CVT_OFF[5] = {0, 36, 36 + 64, 36 + 64 + 144, 36 + 64 + 144 + 400};
constexpr int CVT_MAX_BUCKETS = 16384;

struct CvBinArgs {
  const float* units;     // (nv, n, 8)
  int64_t n;
  int nv;
  int H3, W3;             // finest level
  int ntx, nty;           // tiles per view
  int* counts;            // (nv ntx nty)
  int* offsets;           // (nv ntx nty + 1), exclusive prefix of counts
  int* cursor;            // (nv ntx nty)
  int* order;             // (nv n) record ids grouped by bucket
};

__device__ __forceinline__ int cv_bucket(const CvBinArgs& a, int v, float nx, float ny) {
  int x0 = (int)floorf(unnorm_act(nx, a.W3)), y0 = (int)floorf(unnorm_act(ny, a.H3));
  x0 = min(max(x0, 0), a.W3 - 1);
  y0 = min(max(y0, 0), a.H3 - 1);
  return (v * a.nty + y0 / CVT) * a.ntx + x0 / CVT;
}

// PASS 0: counts;  PASS 1: placement (cursor starts at the bucket offsets)
template <int PASS>
__global__ __launch_bounds__(1024) void costvol_bin_kernel(CvBinArgs a) {
  extern __shared__ int hist[];
  const int nb = a.nv * a.ntx * a.nty;
  for (int b = threadIdx.x; b < nb; b += 1024) hist[b] = 0;
  __syncthreads();
  const int64_t q = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int bucket = -1, local = 0;
  if (q < a.n * a.nv) {
    const f32x4 u = reinterpret_cast<const f32x4*>(a.units)[q * 2 + 1];
    if (u[2] != 0.f) {
      bucket = cv_bucket(a, (int)(q / a.n), u[0], u[1]);
      local = atomicAdd(&hist[bucket], 1);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 1024) {
    const int c = hist[b];
    if (c) {
      if (PASS == 0) atomicAdd(&a.counts[b], c);
      else hist[b] = atomicAdd(&a.cursor[b], c);          // this workgroup's range in the bucket
    }
  }
  if (PASS == 1) {
    __syncthreads();
    if (bucket >= 0) a.order[hist[bucket] + local] = (int)q;
  }
}

// exclusive prefix sum of <= CVT_MAX_BUCKETS counts with one workgroup; cursor = offsets
__global__ __launch_bounds__(1024) void costvol_scan_kernel(const int* __restrict__ counts, int nb, int* __restrict__ offsets,
                                                            int* __restrict__ cursor) {
  __shared__ int part[1024];
  constexpr int PER = CVT_MAX_BUCKETS / 1024;
  int v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int b = threadIdx.x * PER + k;
    v[k] = b < nb ? counts[b] : 0;
    s += v[k];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int b = threadIdx.x * PER + k;
    if (b < nb) { offsets[b] = run; cursor[b] = run; }
    run += v[k];
  }
  if (threadIdx.x == 1023) offsets[nb] = part[1023];
}

struct CvTileArgs {
  const float* units;
  const int* offsets;
  const int* order;
  int64_t n;
  int nv, ntx, nty, stage;
  int hw[8];
  float* gfeats[4];
  const unsigned* gmax;
};

// one workgroup per (view, tile); SPLIT workgroups share a bucket's list (blockIdx.y), each with its own LDS images
__global__ __launch_bounds__(256) void costvol_tile_kernel(CvTileArgs a) {
  __shared__ unsigned long long img[CVT_OFF[4] * 4];          // 64-bit fixed point, channel-planar per level
  const int bucket = blockIdx.x;
  const int beg = a.offsets[bucket], end = a.offsets[bucket + 1];
  const int share = (end - beg + (int)gridDim.y - 1) / (int)gridDim.y;
  const int p0 = beg + share * (int)blockIdx.y, p1 = min(p0 + share, end);
  if (p0 >= p1) return;
  const int v = bucket / (a.ntx * a.nty), tile = bucket % (a.ntx * a.nty);
  const int ty = tile / a.ntx, tx = tile % a.ntx;
  for (int e = threadIdx.x; e < CVT_OFF[4] * 4; e += 256) img[e] = 0ull;
  // one unit = 2^-k: largest term < 2^(ex + 1), at most 4 (p1 - p0) terms per word (every record, all four taps on one texel)
  const int ex = (int)((*a.gmax >> 23) & 0xff) - 127;
  const int k = 60 - ex - (32 - __clz(p1 - p0));
  // level-l origin of the LDS image: the tile's first finest-level pixel mapped to level l, one texel of margin
  int ox[4], oy[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int H = a.hw[2 * l], W = a.hw[2 * l + 1];
    const float rx = (float)(W - 1) / (float)(a.hw[7] - 1), ry = (float)(H - 1) / (float)(a.hw[6] - 1);
    ox[l] = max((int)floorf((float)(tx * CVT) * rx) - 1, 0);
    oy[l] = max((int)floorf((float)(ty * CVT) * ry) - 1, 0);
  }
  __syncthreads();
  // thread t walks its own contiguous share of the list: the records of a bucket arrive in runs of lattice order, and the 64
  // lanes of an instruction taking 64 CONSECUTIVE records (voxels a pixel or less apart) would serialise on the same LDS words
  const int chunk = (p1 - p0 + 255) / 256;
  const int q0 = p0 + (int)threadIdx.x * chunk, q1 = min(q0 + chunk, p1);
  for (int p = q0; p < q1; ++p) {
    const int q = a.order[p];
    const f32x4 g = reinterpret_cast<const f32x4*>(a.units)[(int64_t)q * 2];
    const f32x4 pos = reinterpret_cast<const f32x4*>(a.units)[(int64_t)q * 2 + 1];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l < a.stage) continue;
      const int H = a.hw[2 * l], W = a.hw[2 * l + 1], E = CVT_E[l];
      const float x = unnorm_act(pos[0], W), y = unnorm_act(pos[1], H);
      const float fx = floorf(x), fy = floorf(y);
      const float tx_ = x - fx, ty_ = y - fy;
      const int x0 = (int)fx, y0 = (int)fy;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yi = y0 + dy;
        if ((yi < 0) | (yi >= H)) continue;
        const float wy = dy ? ty_ : 1.0f - ty_;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int xi = x0 + dx;
          if ((xi < 0) | (xi >= W)) continue;
          const float w = (dx ? tx_ : 1.0f - tx_) * wy;
          const int lx = xi - ox[l], ly = yi - oy[l];
          if ((unsigned)lx < (unsigned)E && (unsigned)ly < (unsigned)E) {
            unsigned long long* d = img + CVT_OFF[l] * 4 + ly * E + lx;   // channel-planar: neighbouring texels, neighbouring banks
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float val = w * g[c];
              if (val != 0.f) atomicAdd(d + c * E * E, (unsigned long long)(long long)rintf(ldexpf(val, k)));   // exact: power-of-two scale, |.| < 2^62
            }
          } else {                                              // outside the LDS image (an irregular pyramid): straight to memory
            float* d = a.gfeats[l] + (((int64_t)v * H + yi) * W + xi) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float val = w * g[c];
              if (val != 0.f) atomicAdd(d + c, val);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // flush: every touched float once; the 64 lanes of an add cover 16 consecutive texels of an image row
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < a.stage) continue;
    const int H = a.hw[2 * l], W = a.hw[2 * l + 1], E = CVT_E[l];
    float* map = a.gfeats[l] + (int64_t)v * H * W * 4;
    for (int e = threadIdx.x; e < E * E * 4; e += 256) {
      const int c = e & 3, lx = (e >> 2) % E, ly = (e >> 2) / E;
      const long long acc = (long long)img[CVT_OFF[l] * 4 + c * E * E + ly * E + lx];
      if (acc == 0) continue;
      const float val = (float)ldexp((double)acc, -k);
      const int xi = ox[l] + lx, yi = oy[l] + ly;
      if (xi < W && yi < H) atomicAdd(map + ((int64_t)yi * W + xi) * 4 + c, val);
    }
  }
}

void fill_views(ViewSet& vs, int nv, const float* h_intrs, const float* h_w2c) {
  vs.nv = nv;
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int s = v < nv ? v : 0;
    for (int k = 0; k < 12; ++k) {
      vs.w2c[v][k] = h_w2c[s * 16 + k];
      vs.K[v][k] = h_intrs[s * 16 + k];
    }
  }
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

extern "C" int surf_upsample_filter(const int32_t* parents, int64_t n_parents, int D, const float* depths, int nv, int H,
                                    int W, const float* h_intrs, const float* h_w2c, float depth_range, uint8_t* flags,
                                    void* stream) {
  if (!parents || !depths || !h_intrs || !h_w2c || !flags || n_parents <= 0 || D < 2 || H < 2 || W < 2) return SURF_E_ARG;
  if (nv < 1 || nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  FilterArgs a;
  a.parents = parents; a.n_par = n_parents; a.D = D; a.voxel_size = (float)(2.0 / (double)(D - 1));
  a.depths = depths; a.H = H; a.W = W; a.depth_range = depth_range; a.flags = flags;
  fill_views(a.vs, nv, h_intrs, h_w2c);
  hipLaunchKernelGGL(upsample_filter_kernel, grid1d(n_parents * 8, 256), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int surf_costvol(const int32_t* parents, const int32_t* idx, int64_t n, int D, const float* const* h_feats,
                            const int* h_hw, int stage, int nv, const float* h_intrs, const float* h_w2c,
                            const float* h_agg /* w1(8x4) b1(8) w2(8) b2(1) */, int32_t* coords, float* feat,
                            uint8_t* keep, void* stream) {
  if (!h_feats || !h_hw || !h_intrs || !h_w2c || !h_agg || !coords || !feat || !keep || n <= 0 || D < 2) return SURF_E_ARG;
  if ((idx != nullptr) != (parents != nullptr)) return SURF_E_ARG;
  if (!idx && n != (int64_t)D * D * D) return SURF_E_ARG;
  if (nv < 1 || nv > SURF_MAX_VIEWS || stage < 0 || stage > 3) return SURF_E_LIMIT;
  CostVolArgs a;
  a.parents = parents; a.idx = idx; a.n = n; a.D = D; a.voxel_size = (float)(2.0 / (double)(D - 1)); a.stage = stage;
  for (int l = 0; l < 4; ++l) {
    if (!h_feats[l]) return SURF_E_ARG;
    a.feats[l] = h_feats[l];
    a.hw[2 * l] = h_hw[2 * l];
    a.hw[2 * l + 1] = h_hw[2 * l + 1];
  }
  fill_views(a.vs, nv, h_intrs, h_w2c);
  for (int k = 0; k < 32; ++k) a.w1[k] = h_agg[k];
  for (int k = 0; k < 8; ++k) { a.b1[k] = h_agg[32 + k]; a.w2[k] = h_agg[40 + k]; }
  a.b2 = h_agg[48];
  a.coords = coords; a.feat = feat; a.keep = keep;
  hipLaunchKernelGGL(costvol_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int64_t surf_compact_workspace_ints(int64_t n) { return (n + CP_BLOCK - 1) / CP_BLOCK + 1; }

extern "C" int surf_compact(const uint8_t* flags, int64_t n, int32_t* workspace, int32_t* idx_out, int32_t* total,
                            void* stream) {
  if (!flags || !workspace || !idx_out || !total || n <= 0) return SURF_E_ARG;
  if (n > 0x7fffffffLL) return SURF_E_LIMIT;
  const int nb = (int)((n + CP_BLOCK - 1) / CP_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ((uintptr_t)flags & 15u) == 0;
  hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(CP_THREADS), 0, st, flags, n, workspace, aligned);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, workspace, nb, total);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb), dim3(CP_THREADS), 0, st, flags, n, workspace, idx_out, aligned);
  return surf_check_launch();
}

extern "C" int surf_gather_rows(const void* src, const int32_t* idx, int64_t n, int row_words, int idx_shift,
                                int dst_stride_words, int dst_offset_words, void* dst, void* stream) {
  if (!src || !idx || !dst || n <= 0 || row_words <= 0 || idx_shift < 0 || dst_stride_words < row_words) return SURF_E_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, grid1d(n * row_words, 256), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)src, idx, n, row_words, idx_shift, dst_stride_words, dst_offset_words, (uint32_t*)dst);
  return surf_check_launch();
}

extern "C" int surf_compose_index(const int32_t* a, const int32_t* b, int64_t n, int32_t* out, void* stream) {
  if (!a || !b || !out || n <= 0) return SURF_E_ARG;
  hipLaunchKernelGGL(compose_index_kernel, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
  return surf_check_launch();
}

extern "C" int surf_densify(const int32_t* coords, const float* rows, int row_stride, int64_t n, int D, const float* prev,
                            float* dense, int32_t* table, void* stream) {
  if (!coords || !rows || !dense || !table || n <= 0 || D < 2 || row_stride < 1) return SURF_E_ARG;
  if (prev && (D & 1)) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (D % 4 == 0)
    hipLaunchKernelGGL((dense_init_kernel<4, 4>), grid1d((int64_t)D * D * D / 4, 256 * 4), dim3(256), 0, st, prev, D, dense, table);
  else
    hipLaunchKernelGGL((dense_init_kernel<1, 4>), grid1d((int64_t)D * D * D, 256 * 4), dim3(256), 0, st, prev, D, dense, table);
  hipLaunchKernelGGL(dense_scatter_kernel, grid1d(n, 256), dim3(256), 0, st, coords, rows, row_stride, n, D, dense, table);
  return surf_check_launch();
}

extern "C" int surf_densify_backward(const int32_t* coords, int64_t n, int D, const int32_t* table, const float* g_dense,
                                     int row_stride, float* g_rows, float* g_prev, void* stream) {
  if (!coords || !table || !g_dense || !g_rows || n <= 0 || D < 2 || row_stride < 1) return SURF_E_ARG;
  if (g_prev && (D & 1)) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dense_rows_bwd_kernel, grid1d(n, 256), dim3(256), 0, st, coords, g_dense, n, D, row_stride, g_rows);
  if (g_prev)
  {
    const int64_t total = (int64_t)D * D * D;
    const int Dp = D / 2;
    const int64_t tiles = (int64_t)((Dp + DG_TX - 1) / DG_TX) * ((Dp + DG_TY - 1) / DG_TY) * ((Dp + DG_TZ - 1) / DG_TZ);
#ifndef SURF_DENSE_BWD_SCATTER
    if (tiles <= 0x7fffffffLL && Dp >= 2)
      hipLaunchKernelGGL(dense_init_bwd_gather_kernel, dim3((unsigned)tiles), dim3(256), 0, st, g_dense, table, D, g_prev);
    else
#endif
    if (D % 4 == 0)
      hipLaunchKernelGGL((dense_init_bwd_kernel<4, 8>), grid1d(total / 4, 256 * 8), dim3(256), 0, st, g_dense, table, D, g_prev);
    else
      hipLaunchKernelGGL((dense_init_bwd_kernel<1, 8>), grid1d(total, 256 * 8), dim3(256), 0, st, g_dense, table, D, g_prev);
  }
  return surf_check_launch();
}

extern "C" int surf_scatter_rows_add(const float* g_dst, const int32_t* idx, int64_t n, int row_words, int idx_shift,
                                     int dst_stride_words, int dst_offset_words, float* g_src, void* stream) {
  if (!g_dst || !idx || !g_src || n <= 0 || row_words <= 0 || idx_shift < 0 || dst_stride_words < row_words) return SURF_E_ARG;
  hipLaunchKernelGGL(scatter_rows_add_kernel, grid1d(n * row_words, 256), dim3(256), 0, (hipStream_t)stream, g_dst, idx, n,
                     row_words, idx_shift, dst_stride_words, dst_offset_words, g_src);
  return surf_check_launch();
}

// the binned form needs a regular pyramid (level l = finest >> (3 - l): what the LDS image edges assume) and <= 16,384 (view, tile)
// buckets (5 views of 576 x 800: 9,000); otherwise - the Tanks&Temples shape - the direct scatter runs
#ifndef SURF_CVB_BINNED
#define SURF_CVB_BINNED 1
#endif
static bool cv_binned_ok(int64_t n, int nv, const int* hw) {
  if (!SURF_CVB_BINNED || n * nv > 0x7fffffffLL) return false;
  if (getenv("SURF_CVB_DIRECT")) return false;            // tests: keep the direct scatter (the fallback of the shapes below) exercised
  const int H3 = hw[6], W3 = hw[7];
  for (int l = 0; l < 4; ++l)
    if (hw[2 * l] != (H3 >> (3 - l)) || hw[2 * l + 1] != (W3 >> (3 - l)) || hw[2 * l] < 2 || hw[2 * l + 1] < 2) return false;
  const int64_t nb = (int64_t)nv * ((W3 + CVT - 1) / CVT) * ((H3 + CVT - 1) / CVT);
  return nb <= CVT_MAX_BUCKETS;
}

extern "C" int64_t surf_costvol_backward_workspace_floats(int64_t n, int nv, int H, int W) {
  // agg replicas | units (nv n 8) | order (nv n) | counts, offsets (+1), cursor
  const int64_t nb = (int64_t)nv * ((W + CVT - 1) / CVT) * ((H + CVT - 1) / CVT);
  return (int64_t)CV_REPLICAS * 64 + n * nv * 9 + 3 * nb + 8;
}

extern "C" int64_t surf_costvol_backward_workspace_floats_for(int64_t n, int nv, const int* h_hw) {
  // the exact need of THIS pyramid: the direct scatter (irregular pyramid / Tanks&Temples shape, > 16,384 buckets, SURF_CVB_DIRECT)
  // only uses the agg_mlp replicas - no 9 floats per (view, voxel) pair (1.1 - 1.5 GB at the finest DTU stage)
  if (!h_hw || n <= 0 || nv < 1) return 0;
  if (!cv_binned_ok(n, nv, h_hw)) return (int64_t)CV_REPLICAS * 64;
  return surf_costvol_backward_workspace_floats(n, nv, h_hw[6], h_hw[7]);
}

extern "C" int surf_costvol_backward(const int32_t* coords, const float* g, int64_t n, int D, const float* const* h_feats,
                                     float* const* h_gfeats, const int* h_hw, int stage, int nv, const float* h_intrs,
                                     const float* h_w2c, const float* h_agg, float* workspace, float* g_agg, void* stream) {
  if (!coords || !g || !h_feats || !h_gfeats || !h_hw || !h_intrs || !h_w2c || !h_agg || !workspace || !g_agg || n <= 0 || D < 2)
    return SURF_E_ARG;
  if (nv < 1 || nv > SURF_MAX_VIEWS || stage < 0 || stage > 3) return SURF_E_LIMIT;
  CostVolBwdArgs a;
  a.coords = coords; a.g = g; a.n = n; a.voxel_size = (float)(2.0 / (double)(D - 1)); a.stage = stage; a.gagg = workspace;
  for (int l = 0; l < 4; ++l) {
    if (!h_feats[l] || (l >= stage && !h_gfeats[l])) return SURF_E_ARG;
    a.feats[l] = h_feats[l];
    a.gfeats[l] = h_gfeats[l];
    a.hw[2 * l] = h_hw[2 * l];
    a.hw[2 * l + 1] = h_hw[2 * l + 1];
  }
  fill_views(a.vs, nv, h_intrs, h_w2c);
  for (int k = 0; k < 32; ++k) a.w1[k] = h_agg[k];
  for (int k = 0; k < 8; ++k) { a.b1[k] = h_agg[32 + k]; a.w2[k] = h_agg[40 + k]; }
  a.b2 = h_agg[48];
  hipStream_t st = (hipStream_t)stream;
  const bool binned = cv_binned_ok(n, nv, h_hw);
  const int H3 = h_hw[6], W3 = h_hw[7];
  const int ntx = (W3 + CVT - 1) / CVT, nty = (H3 + CVT - 1) / CVT, nb = nv * ntx * nty;
  float* units = workspace + CV_REPLICAS * 64;
  int* order = reinterpret_cast<int*>(units + n * nv * 8);
  int* counts = order + n * nv;
  int* offsets = counts + nb;
  int* cursor = offsets + nb + 1;
  unsigned* gmax = reinterpret_cast<unsigned*>(cursor + nb);
  a.units = binned ? units : nullptr;
  a.gmax = binned ? gmax : nullptr;
  hipError_t e = hipMemsetAsync(workspace, 0, CV_REPLICAS * 64 * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  if (binned) {
    e = hipMemsetAsync(counts, 0, (size_t)(3 * nb + 2) * sizeof(int), st);       // counts ... cursor, gmax
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(costvol_bwd_kernel, grid1d(n, 256), dim3(256), 0, st, a);
  if (binned) {
    CvBinArgs b;
    b.units = units; b.n = n; b.nv = nv; b.H3 = H3; b.W3 = W3; b.ntx = ntx; b.nty = nty;
    b.counts = counts; b.offsets = offsets; b.cursor = cursor; b.order = order;
    const dim3 bgrid = grid1d(n * nv, 1024);
    hipLaunchKernelGGL(costvol_bin_kernel<0>, bgrid, dim3(1024), (size_t)nb * sizeof(int), st, b);
    hipLaunchKernelGGL(costvol_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nb, offsets, cursor);
    hipLaunchKernelGGL(costvol_bin_kernel<1>, bgrid, dim3(1024), (size_t)nb * sizeof(int), st, b);
    CvTileArgs t;
    t.units = units; t.offsets = offsets; t.order = order; t.n = n; t.nv = nv; t.ntx = ntx; t.nty = nty; t.stage = stage;
    for (int l = 0; l < 4; ++l) { t.hw[2 * l] = h_hw[2 * l]; t.hw[2 * l + 1] = h_hw[2 * l + 1]; t.gfeats[l] = h_gfeats[l]; }
    t.gmax = gmax;
    // a bucket holds n nv / nb records on average and several times that at the image centre: SPLIT workgroups per bucket
    const int64_t avg = n * nv / nb;
    const int split = avg > 16384 ? 8 : (avg > 4096 ? 4 : (avg > 1024 ? 2 : 1));
    hipLaunchKernelGGL(costvol_tile_kernel, dim3(nb, split), dim3(256), 0, st, t);
  }
  hipLaunchKernelGGL(costvol_bwd_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, g_agg);
  return surf_check_launch();
}
