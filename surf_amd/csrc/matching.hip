// K7: matching field -- per-view expected depth from the dense matching volume.
// Restates MatchingField.forward / depth_render  matching_field.py:73-141, 18-71  (perturb = False).
//
// Two or eight lanes = one low-resolution pixel ray of one view (neighbouring rays walk neighbouring voxels,
// which keeps the trilinear gathers in L2); the softmax expectation over the n (or 2n) samples is
// accumulated online, so nothing is stored per sample and the reference's sort of the two bands is not
// needed (a softmax-weighted mean does not depend on sample order).
#include <stdlib.h>

#include "common.h"

namespace {

struct MatchArgs {
  const float* mvol;
  int D;
  int nv;
  float Kinv[SURF_MAX_VIEWS][9];  // inverse(intrinsics)[:3,:3]
  float R[SURF_MAX_VIEWS][9];     // c2w[:3,:3]
  float Rinv[SURF_MAX_VIEWS][9];  // inverse(c2w[:3,:3])
  float t[SURF_MAX_VIEWS][3];
  float nearv[SURF_MAX_VIEWS], farv[SURF_MAX_VIEWS];
  int H, W, h, w;
  const float* lin_x;  // torch.linspace(0, W-1, w)
  const float* lin_y;  // torch.linspace(0, H-1, h)
  const float* lin_n;  // torch.linspace(0, 1, n)
  int n;
  const float* pre;    // (nv,H,W) previous-stage depths or null
  float ratio_cur, ratio_prev;
  const float* jitter; // (nv, h*w, 2) train-mode `rand - 0.5` per ray and band, or null (matching_field.py:33-35)
  float* out;          // (nv,h,w)
  float* stats;        // (nv,h,w,4) or null: the forward writes (max logit, softmax denominator, expected z, 0) per ray, the
                       // backward reads them instead of walking the samples a first time (round 5)
};

__device__ __forceinline__ void band(float zc, float half, float n0, float f0, float& lo, float& hi) {
  lo = zc - half;
  hi = zc + half;
  if (hi > f0) lo = lo - (hi - f0);
  if (lo < n0) hi = hi + (n0 - lo);
  lo = fminf(fmaxf(lo, n0), f0);
  hi = fminf(fmaxf(hi, n0), f0);
}

// LPR lanes per ray (2 or 8): lane j fetches the corners c LPR + j (c < 8 / LPR) of every sample's trilinear cell, corner id =
// 4 dx + 2 dy + dz, so the two z-corners of a row are adjacent lanes AND adjacent floats (8 contiguous bytes per request); the
// partial sums meet by log2(LPR) xor-shuffles.  One lane per ray (64 scattered 4-byte requests per load instruction) ran at 53 %
// of the HBM roofline.  Eight lanes (one corner each) win on the fine stages (many rays, <= 64 samples: latency-bound), two
// lanes on the coarse ones (128 samples per ray: the per-sample arithmetic is replicated per lane): measured per stage
// 0.36 / 0.88 / 1.02 / 1.89 ms (one lane) vs 0.29 / 0.79 / 1.55 / 2.57 (two) vs 0.41 / 1.26 / 0.69 / 1.31 (eight).
template <int LPR>
__global__ __launch_bounds__(256) void matching_depth_kernel(MatchArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_view = (int64_t)a.h * a.w;
  const int j = (int)(t % LPR);
  const int64_t i = t / LPR;
  if (i >= per_view * a.nv) return;               // uniform over the ray's lanes
  const int v = (int)(i / per_view);
  const int p = (int)(i % per_view);
  const float px = a.lin_x[p % a.w], py = a.lin_y[p / a.w];
  const float* Ki = a.Kinv[v];
  float cx = Ki[0] * px + Ki[1] * py + Ki[2];
  float cy = Ki[3] * px + Ki[4] * py + Ki[5];
  float cz_ = Ki[6] * px + Ki[7] * py + Ki[8];
  const float nrm = sqrtf(cx * cx + cy * cy + cz_ * cz_);
  cx /= nrm; cy /= nrm; cz_ /= nrm;
  const float* R = a.R[v];
  const float dx = R[0] * cx + R[1] * cy + R[2] * cz_;
  const float dy = R[3] * cx + R[4] * cy + R[5] * cz_;
  const float dz = R[6] * cx + R[7] * cy + R[8] * cz_;
  const float* Ri = a.Rinv[v];
  const float cosz = Ri[6] * dx + Ri[7] * dy + Ri[8] * dz;  // z of the ray direction in the camera frame
  const float ox = a.t[v][0], oy = a.t[v][1], oz = a.t[v][2];
  const float n0 = a.nearv[v], f0 = a.farv[v];

  float lo[2], hi[2];
  int nb = 1;
  lo[0] = n0; hi[0] = f0; lo[1] = n0; hi[1] = f0;
  if (a.pre) {
    const float pre = a.pre[((int64_t)v * a.H + (int)py) * a.W + (int)px];
    const float zc = pre / cosz;
    band(zc, ((f0 - n0) * a.ratio_cur) / 2.0f, n0, f0, lo[0], hi[0]);
    band(zc, ((f0 - n0) * a.ratio_prev) / 2.0f, n0, f0, lo[1], hi[1]);
    nb = 2;
  }
  const int D = a.D;
  float m = -INFINITY, den = 0.f, num = 0.f;
  for (int b = 0; b < nb; ++b) {
    const float rng = hi[b] - lo[b];
    const float shift = a.jitter ? a.jitter[i * 2 + b] * rng / (float)a.n : 0.f;
    for (int k = 0; k < a.n; ++k) {
      float z = lo[b] + rng * a.lin_n[k];
      if (a.jitter) z = z + shift;
      const float qx = unnorm_acf(ox + dx * z, D), qy = unnorm_acf(oy + dy * z, D), qz = unnorm_acf(oz + dz * z, D);
      const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
      const float tx = qx - fx, ty = qy - fy, tz = qz - fz;
      const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
      float rho = 0.f;
#pragma unroll
      for (int c = 0; c < 8 / LPR; ++c) {
        const int corner = c * LPR + j;
        const int cdx = corner >> 2, cdy = (corner >> 1) & 1, cdz = corner & 1;
        const int xi = x0 + cdx, yi = y0 + cdy, zi = z0 + cdz;
        const float wgt = (cdx ? tx : 1.0f - tx) * (cdy ? ty : 1.0f - ty) * (cdz ? tz : 1.0f - tz);
        const bool ok = (xi >= 0) & (xi < D) & (yi >= 0) & (yi < D) & (zi >= 0) & (zi < D);
        if (ok) rho += a.mvol[((int64_t)xi * D + yi) * D + zi] * wgt;
      }
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) rho += __shfl_xor(rho, o);
      const float mn = fmaxf(m, rho);
      const float sc = expf(m - mn), e = expf(rho - mn);
      den = den * sc + e;
      num = num * sc + e * z;
      m = mn;
    }
  }
  if (j == 0) {
    a.out[i] = (num / den) * cosz;
    if (a.stats) reinterpret_cast<f32x4*>(a.stats)[i] = f32x4{m, den, num / den, 0.f};
  }
}

// Round 6: SAMPLE-per-lane form.  L = 32 or 64 lanes = one ray, lane l = samples l, l + L, ... of the ray's nb n samples: the position
// arithmetic of a sample is done once (the corner-per-lane forms above repeat it in every lane of the ray: they are bound by VALU
// issue, 240 instructions per ray-sample-octet), its eight corners are four 8-byte loads (the z-neighbours of a cell are adjacent
// floats), and the softmax expectation is a segmented wave reduction of per-lane (max, denominator, numerator) triples instead of
// a serial online chain over the samples.  Same sums in another order: results agree with the forms above to rounding.
// Measured per stage (bench scene, same box, three alternations): 0.196 / 0.70 / 0.60 / 1.20 ms (best of the forms above) ->
// 0.164 / 0.73 / 0.525 / 0.96 ms: -12 % in all - less than the instruction count promised, because what remains is the texture
// path's cache-line rate (a wave's 64 samples touch ~256 distinct lines of the z-fastest volume per step either way).
struct __attribute__((packed, aligned(4))) F2u { float a, b; };      // an 8-byte load from a 4-byte aligned address

template <int L>
__global__ __launch_bounds__(256) void matching_depth_spl_kernel(MatchArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_view = (int64_t)a.h * a.w;
  const int l = (int)(t % L);
  const int64_t ray = t / L;
  const bool live = ray < per_view * a.nv;        // uniform over the ray's lanes; dead rays walk ray 0 (the shuffles want every lane)
  const int64_t i = live ? ray : 0;
  const int v = (int)(i / per_view);
  const int p = (int)(i % per_view);
  const float px = a.lin_x[p % a.w], py = a.lin_y[p / a.w];
  const float* Ki = a.Kinv[v];
  float cx = Ki[0] * px + Ki[1] * py + Ki[2];
  float cy = Ki[3] * px + Ki[4] * py + Ki[5];
  float cz_ = Ki[6] * px + Ki[7] * py + Ki[8];
  const float nrm = sqrtf(cx * cx + cy * cy + cz_ * cz_);
  cx /= nrm; cy /= nrm; cz_ /= nrm;
  const float* R = a.R[v];
  const float dx = R[0] * cx + R[1] * cy + R[2] * cz_;
  const float dy = R[3] * cx + R[4] * cy + R[5] * cz_;
  const float dz = R[6] * cx + R[7] * cy + R[8] * cz_;
  const float* Ri = a.Rinv[v];
  const float cosz = Ri[6] * dx + Ri[7] * dy + Ri[8] * dz;
  const float ox = a.t[v][0], oy = a.t[v][1], oz = a.t[v][2];
  const float n0 = a.nearv[v], f0 = a.farv[v];
  float lo[2], hi[2];
  int nb = 1;
  lo[0] = n0; hi[0] = f0; lo[1] = n0; hi[1] = f0;
  if (a.pre) {
    const float pre = a.pre[((int64_t)v * a.H + (int)py) * a.W + (int)px];
    const float zc = pre / cosz;
    band(zc, ((f0 - n0) * a.ratio_cur) / 2.0f, n0, f0, lo[0], hi[0]);
    band(zc, ((f0 - n0) * a.ratio_prev) / 2.0f, n0, f0, lo[1], hi[1]);
    nb = 2;
  }
  const int D = a.D, S = nb * a.n;
  const float* __restrict__ mvol = a.mvol;
  float m = -INFINITY, den = 0.f, num = 0.f;
  for (int s = l; s < S; s += L) {
    const int b = s >= a.n ? 1 : 0, k = s - b * a.n;
    const float rng = hi[b] - lo[b];
    float z = lo[b] + rng * a.lin_n[k];
    if (a.jitter) z = z + a.jitter[i * 2 + b] * rng / (float)a.n;
    const float qx = unnorm_acf(ox + dx * z, D), qy = unnorm_acf(oy + dy * z, D), qz = unnorm_acf(oz + dz * z, D);
    const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
    const float tx = qx - fx, ty = qy - fy, tz = qz - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const bool zlo = (z0 >= 0) & (z0 < D), zhi = (z0 + 1 >= 0) & (z0 + 1 < D);
    float rho = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cdx = c >> 1, cdy = c & 1;
      const int xi = x0 + cdx, yi = y0 + cdy;
      if ((xi >= 0) & (xi < D) & (yi >= 0) & (yi < D)) {
        const float wxy = (cdx ? tx : 1.0f - tx) * (cdy ? ty : 1.0f - ty);
        const float* __restrict__ row = mvol + ((int64_t)xi * D + yi) * D;
        float v0 = 0.f, v1 = 0.f;
        if (zlo & zhi) {
          const F2u pr = *reinterpret_cast<const F2u*>(row + z0);
          v0 = pr.a; v1 = pr.b;
        } else {
          if (zlo) v0 = row[z0];
          if (zhi) v1 = row[z0 + 1];
        }
        // (corner weight = the product in the order of the forms above: x, y, z)
        if (zlo) rho += v0 * (wxy * (1.0f - tz));
        if (zhi) rho += v1 * (wxy * tz);
      }
    }
    const float mn = fmaxf(m, rho);
    const float sc = expf(m - mn), e = expf(rho - mn);
    den = den * sc + e;
    num = num * sc + e * z;
    m = mn;
  }
  // merge the L lanes' (max, denominator, numerator) triples; a lane without samples carries (-inf, 0, 0)
#pragma unroll
  for (int o = 1; o < L; o <<= 1) {
    const float m2 = __shfl_xor(m, o), den2 = __shfl_xor(den, o), num2 = __shfl_xor(num, o);
    const float mn = fmaxf(m, m2);
    const float s1 = m == -INFINITY ? 0.f : expf(m - mn), s2 = m2 == -INFINITY ? 0.f : expf(m2 - mn);
    den = den * s1 + den2 * s2;
    num = num * s1 + num2 * s2;
    m = mn;
  }
  if (l == 0 && live) {
    a.out[i] = (num / den) * cosz;
    if (a.stats) reinterpret_cast<f32x4*>(a.stats)[i] = f32x4{m, den, num / den, 0.f};
  }
}

// F.interpolate(size=(H,W), mode='bilinear', align_corners=False) of (nv,h,w) maps  (matching_field.py:137)
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ src, int nv, int h, int w, int H,
                                                                int W, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)H * W;
  if (i >= per * nv) return;
  const int v = (int)(i / per);
  const int y = (int)((i % per) / W), x = (int)(i % W);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
  if (fy < 0.f) fy = 0.f;
  if (fx < 0.f) fx = 0.f;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float* s = src + (int64_t)v * h * w;
  dst[i] = hy * (hx * s[y0 * w + x0] + lx * s[y0 * w + x1]) + ly * (hx * s[y1 * w + x0] + lx * s[y1 * w + x1]);
}

// ---- backward (train mode): d loss / d matching volume given d loss / d full-resolution depth maps --------------------
// Transpose of upsample_bilinear_kernel: one thread per full-resolution pixel, four atomics into the low-resolution map.
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ g_full, int nv, int h, int w, int H,
                                                                    int W, float* __restrict__ g_lr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)H * W;
  if (i >= per * nv) return;
  const float g = g_full[i];
  if (g == 0.f) return;
  const int v = (int)(i / per);
  const int y = (int)((i % per) / W), x = (int)(i % W);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
  if (fy < 0.f) fy = 0.f;
  if (fx < 0.f) fx = 0.f;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  float* d = g_lr + (int64_t)v * h * w;
  atomicAdd(d + y0 * w + x0, g * hy * hx);
  atomicAdd(d + y0 * w + x1, g * hy * lx);
  atomicAdd(d + y1 * w + x0, g * ly * hx);
  atomicAdd(d + y1 * w + x1, g * ly * lx);
}

__device__ __forceinline__ void trilinear_scatter(float* __restrict__ vol, int D, float gx, float gy, float gz, float g) {
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
  for (int dx = 0; dx < 2; ++dx)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dz = 0; dz < 2; ++dz) {
        const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
        const float wgt = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
        if ((xi >= 0) & (xi < D) & (yi >= 0) & (yi < D) & (zi >= 0) & (zi < D) && wgt != 0.f)
          atomicAdd(vol + ((int64_t)xi * D + yi) * D + zi, g * wgt);
      }
}

// depth = cosz sum_k z_k softmax(rho)_k with the z_k constants (the bands come from detached depths, matching_field.py:104):
//   d rho_k = g cosz w_k (z_k - E),  E = depth / cosz;  each d rho_k is scattered through the trilinear taps of its sample.
// Eight lanes per ray, lane j = corner (dx, dy, dz) = (j >> 2, (j >> 1) & 1, j & 1) of every sample: the corner values meet by
// three xor-shuffles (rho).  Pass 1 recomputes the softmax statistics (m, den), pass 2 the weights: nothing was stored per
// sample in the forward.  `views`: the views that carry gradient (0 and src_idx), one grid slice each.
//
// The scatter is what bounds this kernel: float atomics are served at the memory side at a per-REQUEST rate (~2-3 10^10 /s
// chip-wide, MI355X_MICROARCH.md "Global float atomics"), and a trilinear scatter issues one request per z-pair of corners
// (round 2: 324 M requests per training step = 12 of the kernel's 14 ms).  Round 3: a workgroup owns an 8 x 4 PATCH of
// low-resolution pixels (not a 32 x 1 strip) - neighbouring rays of a patch walk the same voxel cells, 0.35-0.7 voxels apart -
// and accumulates its scatter in an LDS hash table keyed by the z-pair of the cell (open addressing, linear probing; ds_add_f32
// for the values); the table is flushed once, two adjacent lanes per entry so that a request still carries the 8 contiguous
// bytes of a z-pair.  Entries that do not find a slot within MAX_PROBE steps go to global memory directly (never observed on
// the bench scene: a patch touches ~1 K pairs, the table holds 2 K).  Summation order changes (it was already atomic-ordered).
constexpr int MB_TX = 8, MB_TY = 4;            // rays per workgroup: 8 x 4 patch x 8 lanes = 256 threads
#ifndef SURF_MB_SLOT_BITS
#define SURF_MB_SLOT_BITS 11
#endif
constexpr int MB_SLOT_BITS = SURF_MB_SLOT_BITS, MB_SLOTS = 1 << MB_SLOT_BITS, MB_MAX_PROBE = 32;   // 24 KB of LDS: six workgroups per CU

__device__ __forceinline__ void mb_accumulate(int* __restrict__ keys, float* __restrict__ vals, int64_t off, float gv,
                                              float* __restrict__ dmvol) {
  const int key = (int)(off >> 1), half = (int)(off & 1);
  unsigned hsh = ((unsigned)key * 2654435761u) >> (32 - MB_SLOT_BITS);
#pragma unroll 1
  for (int probe = 0; probe < MB_MAX_PROBE; ++probe) {
    const int slot = (int)((hsh + probe) & (MB_SLOTS - 1));
    int cur = keys[slot];
    if (cur == -1) cur = atomicCAS(&keys[slot], -1, key), cur = cur == -1 ? key : cur;
    if (cur == key) {
      atomicAdd(&vals[2 * slot + half], gv);
      return;
    }
  }
  atomicAdd(dmvol + off, gv);                                    // table (locally) full
}

__global__ __launch_bounds__(256) void matching_depth_bwd_kernel(MatchArgs a, const float* __restrict__ g_lr, float* __restrict__ dmvol,
                                                                 int view0, int view1) {
  __shared__ int keys[MB_SLOTS];
  __shared__ float vals[2 * MB_SLOTS];
  for (int e = threadIdx.x; e < MB_SLOTS; e += 256) {
    keys[e] = -1;
    vals[2 * e] = 0.f;
    vals[2 * e + 1] = 0.f;
  }
  __syncthreads();
  const int tiles_x = (a.w + MB_TX - 1) / MB_TX, tiles_y = (a.h + MB_TY - 1) / MB_TY;
  const int tile = (int)(blockIdx.x % (unsigned)(tiles_x * tiles_y));
  const int v = blockIdx.x / (unsigned)(tiles_x * tiles_y) == 0 ? view0 : view1;
  const int j = (int)(threadIdx.x & 7);
  const int r = (int)(threadIdx.x >> 3);               // ray of the patch
  const int pxi = (tile % tiles_x) * MB_TX + (r & (MB_TX - 1)), pyi = (tile / tiles_x) * MB_TY + r / MB_TX;
  const int64_t per_view = (int64_t)a.h * a.w;
  bool live = v >= 0 && pxi < a.w && pyi < a.h;
  const int p = live ? pyi * a.w + pxi : 0;
  const int64_t i = (int64_t)(v < 0 ? 0 : v) * per_view + p;
  const float g = live ? g_lr[i] : 0.f;
  live = live && g != 0.f;                               // octet-uniform
  if (live) {
    const float px = a.lin_x[pxi], py = a.lin_y[pyi];
    const float* Ki = a.Kinv[v];
    float cx = Ki[0] * px + Ki[1] * py + Ki[2];
    float cy = Ki[3] * px + Ki[4] * py + Ki[5];
    float cz_ = Ki[6] * px + Ki[7] * py + Ki[8];
    const float nrm = sqrtf(cx * cx + cy * cy + cz_ * cz_);
    cx /= nrm; cy /= nrm; cz_ /= nrm;
    const float* R = a.R[v];
    const float dx = R[0] * cx + R[1] * cy + R[2] * cz_;
    const float dy = R[3] * cx + R[4] * cy + R[5] * cz_;
    const float dz = R[6] * cx + R[7] * cy + R[8] * cz_;
    const float* Ri = a.Rinv[v];
    const float cosz = Ri[6] * dx + Ri[7] * dy + Ri[8] * dz;
    const float ox = a.t[v][0], oy = a.t[v][1], oz = a.t[v][2];
    const float n0 = a.nearv[v], f0 = a.farv[v];
    float lo[2], hi[2];
    int nb = 1;
    lo[0] = n0; hi[0] = f0; lo[1] = n0; hi[1] = f0;
    if (a.pre) {
      const float pre = a.pre[((int64_t)v * a.H + (int)py) * a.W + (int)px];
      const float zc = pre / cosz;
      band(zc, ((f0 - n0) * a.ratio_cur) / 2.0f, n0, f0, lo[0], hi[0]);
      band(zc, ((f0 - n0) * a.ratio_prev) / 2.0f, n0, f0, lo[1], hi[1]);
      nb = 2;
    }
    const int cdx = j >> 2, cdy = (j >> 1) & 1, cdz = j & 1;
    const int D = a.D;
    float m = -INFINITY, den = 0.f, num = 0.f;
    float Es = 0.f;
    const bool have = a.stats != nullptr;                // the forward's statistics: one walk over the samples instead of two
    if (have) {
      const f32x4 st = reinterpret_cast<const f32x4*>(a.stats)[i];
      m = st[0]; den = st[1]; Es = st[2];
    }
    for (int pass = have ? 1 : 0; pass < 2; ++pass) {
      const float E = pass ? (have ? Es : num / den) : 0.f;
      for (int b = 0; b < nb; ++b) {
        const float rng = hi[b] - lo[b];
        const float shift = a.jitter ? a.jitter[i * 2 + b] * rng / (float)a.n : 0.f;
        for (int k = 0; k < a.n; ++k) {
          float z = lo[b] + rng * a.lin_n[k];
          if (a.jitter) z = z + shift;
          const float qx = unnorm_acf(ox + dx * z, D), qy = unnorm_acf(oy + dy * z, D), qz = unnorm_acf(oz + dz * z, D);
          const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
          const float tx = qx - fx, ty = qy - fy, tz = qz - fz;
          const int xi = (int)fx + cdx, yi = (int)fy + cdy, zi = (int)fz + cdz;
          const float wgt = (cdx ? tx : 1.0f - tx) * (cdy ? ty : 1.0f - ty) * (cdz ? tz : 1.0f - tz);
          const bool ok = (xi >= 0) & (xi < D) & (yi >= 0) & (yi < D) & (zi >= 0) & (zi < D);
          const int64_t off = ((int64_t)xi * D + yi) * D + zi;
          float rho = ok ? a.mvol[off] * wgt : 0.f;
          rho += __shfl_xor(rho, 1);
          rho += __shfl_xor(rho, 2);
          rho += __shfl_xor(rho, 4);
          if (pass == 0) {
            const float mn = fmaxf(m, rho);
            const float sc = expf(m - mn), e = expf(rho - mn);
            den = den * sc + e;
            num = num * sc + e * z;
            m = mn;
          } else {
            const float wk = expf(rho - m) / den;
            const float gv = g * cosz * wk * (z - E) * wgt;
            if (ok && gv != 0.f) mb_accumulate(keys, vals, off, gv, dmvol);
          }
        }
      }
    }
  }
  __syncthreads();
  // flush: lanes 2e, 2e + 1 own the two z-halves of entry e: adjacent lanes, adjacent floats
  for (int e = threadIdx.x; e < 2 * MB_SLOTS; e += 256) {
    const int key = keys[e >> 1];
    const float val = vals[e];
    if (key >= 0 && val != 0.f) atomicAdd(dmvol + 2 * (int64_t)key + (e & 1), val);
  }
}

}  // namespace

static void fill_match_args(MatchArgs& a, int nv, const float* h_kinv, const float* h_c2w, const float* h_rinv, const float* h_near_fars) {
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int s = v < nv ? v : 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        a.Kinv[v][r * 3 + c] = h_kinv[s * 9 + r * 3 + c];
        a.R[v][r * 3 + c] = h_c2w[s * 16 + r * 4 + c];
        a.Rinv[v][r * 3 + c] = h_rinv[s * 9 + r * 3 + c];
      }
    for (int r = 0; r < 3; ++r) a.t[v][r] = h_c2w[s * 16 + r * 4 + 3];
    a.nearv[v] = h_near_fars[s * 2 + 0];
    a.farv[v] = h_near_fars[s * 2 + 1];
  }
}

extern "C" int surf_matching_depth(const float* mvol, int D, int nv, const float* h_kinv, const float* h_c2w,
                                   const float* h_rinv, const float* h_near_fars, int H, int W, int h, int w,
                                   const float* lin_x, const float* lin_y, const float* lin_n, int n, const float* pre_depths,
                                   float ratio_cur, float ratio_prev, const float* jitter, float* depth_lr, float* depth_full,
                                   float* stats, void* stream) {
  if (!mvol || !h_kinv || !h_c2w || !h_rinv || !h_near_fars || !lin_x || !lin_y || !lin_n || !depth_lr || !depth_full)
    return SURF_E_ARG;
  if (D < 2 || H < 1 || W < 1 || h < 1 || w < 1 || n < 1) return SURF_E_ARG;
  if (nv < 1 || nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  MatchArgs a;
  a.mvol = mvol; a.D = D; a.nv = nv; a.H = H; a.W = W; a.h = h; a.w = w; a.lin_x = lin_x; a.lin_y = lin_y; a.lin_n = lin_n;
  a.n = n; a.pre = pre_depths; a.ratio_cur = ratio_cur; a.ratio_prev = ratio_prev; a.jitter = jitter; a.out = depth_lr;
  a.stats = stats;
  fill_match_args(a, nv, h_kinv, h_c2w, h_rinv, h_near_fars);
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_lr = (int64_t)nv * h * w, n_full = (int64_t)nv * H * W;
  const char* form_env = getenv("SURF_MD_FORM");                 // 0: the corner-per-lane forms (A/B, tests); read per call
  const int form = form_env ? atoi(form_env) : 1;
  const int S = (pre_depths ? 2 : 1) * n;
  if (form == 1 && S <= 32)
    hipLaunchKernelGGL(matching_depth_spl_kernel<32>, dim3((unsigned)((n_lr * 32 + 255) / 256)), dim3(256), 0, st, a);
  else if (form == 1)
    hipLaunchKernelGGL(matching_depth_spl_kernel<64>, dim3((unsigned)((n_lr * 64 + 255) / 256)), dim3(256), 0, st, a);
  else if (S <= 64)
    hipLaunchKernelGGL(matching_depth_kernel<8>, dim3((unsigned)((n_lr * 8 + 255) / 256)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(matching_depth_kernel<2>, dim3((unsigned)((n_lr * 2 + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, st, depth_lr, nv, h, w,
                     H, W, depth_full);
  return surf_check_launch();
}

extern "C" int surf_matching_depth_backward(const float* mvol, int D, int nv, const float* h_kinv, const float* h_c2w,
                                            const float* h_rinv, const float* h_near_fars, int H, int W, int h, int w,
                                            const float* lin_x, const float* lin_y, const float* lin_n, int n,
                                            const float* pre_depths, float ratio_cur, float ratio_prev, const float* jitter,
                                            const float* g_full, int view0, int view1, float* g_lr, float* dmvol,
                                            const float* stats, void* stream) {
  if (!mvol || !h_kinv || !h_c2w || !h_rinv || !h_near_fars || !lin_x || !lin_y || !lin_n || !g_full || !g_lr || !dmvol)
    return SURF_E_ARG;
  if (D < 2 || H < 1 || W < 1 || h < 1 || w < 1 || n < 1) return SURF_E_ARG;
  if (nv < 1 || nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  MatchArgs a;
  a.mvol = mvol; a.D = D; a.nv = nv; a.H = H; a.W = W; a.h = h; a.w = w; a.lin_x = lin_x; a.lin_y = lin_y; a.lin_n = lin_n;
  a.n = n; a.pre = pre_depths; a.ratio_cur = ratio_cur; a.ratio_prev = ratio_prev; a.jitter = jitter; a.out = nullptr;
  a.stats = const_cast<float*>(stats);
  fill_match_args(a, nv, h_kinv, h_c2w, h_rinv, h_near_fars);
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_lr = (int64_t)nv * h * w, n_full = (int64_t)nv * H * W;
  const hipError_t e = hipMemsetAsync(g_lr, 0, n_lr * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, st, g_full, nv, h, w, H,
                     W, g_lr);
  if (view0 < 0 || view0 >= nv || view1 >= nv) return SURF_E_ARG;
  if ((int64_t)D * D * D > 0x7fffffffLL) return SURF_E_LIMIT;    // 31-bit cell keys in the workgroup's LDS table (D <= 1290)
  const int64_t tiles = (int64_t)((w + MB_TX - 1) / MB_TX) * ((h + MB_TY - 1) / MB_TY);
  const bool two = view1 >= 0 && view1 != view0;
  hipLaunchKernelGGL(matching_depth_bwd_kernel, dim3((unsigned)(tiles * (two ? 2 : 1))), dim3(256), 0, st, a, g_lr, dmvol, view0,
                     two ? view1 : -1);
  return surf_check_launch();
}
