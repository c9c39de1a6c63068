// K16  per-stage photometric loss of the matching-field depth maps (training).
// Replaces compute_ptloss  models/losses/photometric_loss.py:54-125  (+ SSIM :6-33): every other view is warped into view
// `ref` through its depth map (bilinear, zeros, align_corners=True); per pixel and source view: smooth-L1, smooth-L1 of
// the horizontal / vertical image gradients, SSIM (3x3 means, reflection padding, masked); each reduced to the SUM of its
// `topk` smallest source views (torch.topk(largest=False)) and weighted by the reference mask.
//   ptloss_warp_kernel   one thread per (source, pixel): texel4 (rgb, valid) of the warped image
//   ptloss_terms_kernel  one thread per pixel: [l1 m, gx mx, gy my, ssim m | m, mx, my, m]; the caller sums the columns
// Byte-bound image-space work; sources <= SURF_MAX_VIEWS - 1.
#include "common.h"

// wavefronts per SIMD the forward term kernel is compiled for (round 5: 3 - 168 registers, 20 bytes spilled: 98 -> 83 us per
// launch; 4 spills 172 bytes and is slower, 132 us).  The backward term kernel stays at 2 (237 registers): at 3 it spills 352
// bytes and takes 473 instead of 332 us.
#ifndef SURF_PT_WAVES
#define SURF_PT_WAVES 3
#endif

namespace {

struct PtArgs {
  const float* imgs;    // (nv,H,W,4) texel4 rgb
  const float* depth;   // (H,W) depth of view `ref`
  const float* mask;    // (H,W) reference mask
  int nv, ref, H, W, ns, topk;
  float Kinv[9];                        // inverse(intrs[ref])[:3,:3]
  float c2w[12];                        // c2ws[ref][:3,:4]
  float w2c[SURF_MAX_VIEWS][12];        // inverse(c2ws[s])[:3,:4] per SOURCE slot
  float K[SURF_MAX_VIEWS][9];           // intrs[s][:3,:3]
  int view[SURF_MAX_VIEWS];             // source slot -> view index
  float* warp;          // (ns,H,W,4)
  float* terms;         // (H,W,8)
};

__device__ __forceinline__ float smooth_l1(float d) {
  const float a = fabsf(d);
  return a < 1.0f ? 0.5f * d * d : a - 0.5f;
}

__global__ __launch_bounds__(256) void ptloss_warp_kernel(PtArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)a.H * a.W;
  if (i >= per * a.ns) return;
  const int s = (int)(i / per), p = (int)(i % per), y = p / a.W, x = p % a.W;
  const float d = a.depth[p];
  const float X = (float)x * d, Y = (float)y * d, Z = d;
  const float cx = a.Kinv[0] * X + a.Kinv[1] * Y + a.Kinv[2] * Z;
  const float cy = a.Kinv[3] * X + a.Kinv[4] * Y + a.Kinv[5] * Z;
  const float cz = a.Kinv[6] * X + a.Kinv[7] * Y + a.Kinv[8] * Z;
  const float wx = a.c2w[0] * cx + a.c2w[1] * cy + a.c2w[2] * cz + a.c2w[3];
  const float wy = a.c2w[4] * cx + a.c2w[5] * cy + a.c2w[6] * cz + a.c2w[7];
  const float wz = a.c2w[8] * cx + a.c2w[9] * cy + a.c2w[10] * cz + a.c2w[11];
  const float* M = a.w2c[s];
  const float sx = M[0] * wx + M[1] * wy + M[2] * wz + M[3];
  const float sy = M[4] * wx + M[5] * wy + M[6] * wz + M[7];
  const float sz = M[8] * wx + M[9] * wy + M[10] * wz + M[11];
  const float* K = a.K[s];
  const float px = K[0] * sx + K[1] * sy + K[2] * sz;
  const float py = K[3] * sx + K[4] * sy + K[5] * sz;
  const float pz = K[6] * sx + K[7] * sy + K[8] * sz;
  const float u = px / (pz + 1e-8f), v = py / (pz + 1e-8f);
  const float nx = u / ((float)(a.W - 1) / 2.0f) - 1.0f, ny = v / ((float)(a.H - 1) / 2.0f) - 1.0f;
  const bool ok = fabsf(nx) <= 1.0f && fabsf(ny) <= 1.0f && pz > 0.0f;
  const float gx = ((nx + 1.0f) / 2.0f) * (float)(a.W - 1), gy = ((ny + 1.0f) / 2.0f) * (float)(a.H - 1);   // align_corners=True
  f32x4 t = bilinear_texel4(a.imgs + (int64_t)a.view[s] * per * 4, a.H, a.W, gx, gy);
  t[3] = ok ? 1.0f : 0.0f;
  reinterpret_cast<f32x4*>(a.warp)[i] = t;
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256, SURF_PT_WAVES) void ptloss_terms_kernel(PtArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)a.H * a.W;
  if (p >= per) return;
  const int y = (int)(p / a.W), x = (int)(p % a.W);
  const f32x4* __restrict__ ref = reinterpret_cast<const f32x4*>(a.imgs) + (int64_t)a.ref * per;
  const float mref = a.mask[p];
  const float mx = x + 1 < a.W ? mref * a.mask[p + 1] : 0.f;
  const float my = y + 1 < a.H ? mref * a.mask[p + a.W] : 0.f;
  // reference 3x3 statistics (shared by all sources)
  float r_mu[3] = {0, 0, 0}, r_sq[3] = {0, 0, 0};
  int yy[3], xx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { yy[k] = reflect(y + k - 1, a.H); xx[k] = reflect(x + k - 1, a.W); }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x4 r = ref[(int64_t)yy[j] * a.W + xx[k]];
#pragma unroll
      for (int c = 0; c < 3; ++c) { r_mu[c] += r[c]; r_sq[c] = fmaf(r[c], r[c], r_sq[c]); }
    }
  const f32x4 r0 = ref[p];
  const f32x4 rxn = x + 1 < a.W ? ref[p + 1] : r0, ryn = y + 1 < a.H ? ref[p + a.W] : r0;
  float best[4][SURF_MAX_VIEWS];      // per term: the values of all sources, kept sorted ascending
  for (int s = 0; s < a.ns; ++s) {
    const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(a.warp) + (int64_t)s * per;
    const f32x4 w0 = w[p];
    float l1 = 0.f, gx = 0.f, gy = 0.f;
    const f32x4 wxn = x + 1 < a.W ? w[p + 1] : w0, wyn = y + 1 < a.H ? w[p + a.W] : w0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      l1 += smooth_l1(w0[c] - r0[c]);
      gx += smooth_l1((w0[c] - wxn[c]) - (r0[c] - rxn[c]));
      gy += smooth_l1((w0[c] - wyn[c]) - (r0[c] - ryn[c]));
    }
    l1 /= 3.0f; gx /= 3.0f; gy /= 3.0f;
    float w_mu[3] = {0, 0, 0}, w_sq[3] = {0, 0, 0}, wr[3] = {0, 0, 0}, mpool = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int64_t q = (int64_t)yy[j] * a.W + xx[k];
        const f32x4 t = w[q], r = ref[q];
        mpool += (t[3] > 0.5f && a.mask[q] > 0.5f) ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { w_mu[c] += t[c]; w_sq[c] = fmaf(t[c], t[c], w_sq[c]); wr[c] = fmaf(t[c], r[c], wr[c]); }
      }
    mpool /= 9.0f;
    float ssim = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mux = w_mu[c] / 9.0f, muy = r_mu[c] / 9.0f;
      const float sgx = w_sq[c] / 9.0f - mux * mux, sgy = r_sq[c] / 9.0f - muy * muy, sgxy = wr[c] / 9.0f - mux * muy;
      const float n = (2.0f * mux * muy + 1e-4f) * (2.0f * sgxy + 9e-4f);
      const float d = (mux * mux + muy * muy + 1e-4f) * (sgx + sgy + 9e-4f);
      ssim += mpool * fminf(fmaxf((1.0f - n / d) / 2.0f, 0.0f), 1.0f);
    }
    ssim /= 3.0f;
    const float v4[4] = {l1, gx, gy, ssim};
#pragma unroll
    for (int t = 0; t < 4; ++t) {       // insertion into the ascending list
      int j = s;
      while (j > 0 && best[t][j - 1] > v4[t]) { best[t][j] = best[t][j - 1]; --j; }
      best[t][j] = v4[t];
    }
  }
  float sum[4] = {0, 0, 0, 0};
  for (int t = 0; t < 4; ++t)
    for (int k = 0; k < a.topk; ++k) sum[t] += best[t][k];
  f32x4* out = reinterpret_cast<f32x4*>(a.terms) + p * 2;
  out[0] = f32x4{sum[0] * mref, sum[1] * mx, sum[2] * my, sum[3] * mref};
  out[1] = f32x4{mref, mx, my, mref};     // column 7 = column 4: the SSIM term's denominator, so that rows 0-3 / rows 4-7 is one division
}

// ---- backward w.r.t. the depth map (train mode) --------------------------------------------------------------------
// loss = sum_t T_t / (M_t + 1e-8); coef[t] = upstream / (M_t + 1e-8) (device, 4 floats: l1, gx, gy, ssim).
//   ptloss_bwd_terms_kernel  one thread per pixel: recomputes the four per-source values, selects the topk sources of each
//                            term like the forward, and scatters d loss / d warped rgb into g_warp (ns, 4 planes, H W) (atomics: the
//                            gradient and SSIM windows overlap);
//   ptloss_bwd_depth_kernel  one thread per pixel: d warped / d depth = bilinear derivative x d(u,v)/d depth (the pixel's
//                            projection is affine in its depth: p(d) = d a + b), summed over the sources.
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f); }

struct PtBwd {
  const float* coef;
  float* g_warp;
  float* g_depth;
};

template <bool BWD>
__device__ __forceinline__ void pt_source(const PtArgs& a, const PtBwd& b, int s, int64_t p, int x, int y, const int yy[3],
                                          const int xx[3], const float r_mu[3], const float r_sq[3], float mref, float mx,
                                          float my, const unsigned sel /* bit t: this source is selected for term t */,
                                          float v4[4]) {
  const int64_t per = (int64_t)a.H * a.W;
  const f32x4* __restrict__ ref = reinterpret_cast<const f32x4*>(a.imgs) + (int64_t)a.ref * per;
  const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(a.warp) + (int64_t)s * per;
  // Round 5: g_warp is CHANNEL-PLANAR, (ns, 4, H W) - the 64 lanes of an add are 64 adjacent pixels of one channel: contiguous
  // dwords = four 64-byte segments per instruction instead of sixteen in the texel4 layout (a float atomic costs ~12.2 ns per
  // CU and distinct segment: scripts/microbench/atomic_shapes.hip)
  float* gw = BWD ? b.g_warp + (int64_t)s * per * 4 : nullptr;
#define GW(q, c) (gw + (int64_t)(c) * per + (q))
  const f32x4 r0 = ref[p], w0 = w[p];
  const bool hx = x + 1 < a.W, hy = y + 1 < a.H;
  const f32x4 rxn = hx ? ref[p + 1] : r0, ryn = hy ? ref[p + a.W] : r0;
  const f32x4 wxn = hx ? w[p + 1] : w0, wyn = hy ? w[p + a.W] : w0;
  float l1 = 0.f, gx = 0.f, gy = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float d0 = w0[c] - r0[c], d1 = (w0[c] - wxn[c]) - (r0[c] - rxn[c]), d2 = (w0[c] - wyn[c]) - (r0[c] - ryn[c]);
    l1 += smooth_l1(d0);
    gx += smooth_l1(d1);
    gy += smooth_l1(d2);
    if (BWD) {
      float g0 = 0.f;
      if (sel & 1u) g0 += b.coef[0] * mref * smooth_l1_grad(d0) / 3.0f;
      if ((sel & 2u) && hx) {
        const float g = b.coef[1] * mx * smooth_l1_grad(d1) / 3.0f;
        g0 += g;
        if (g != 0.f) atomicAdd(GW(p + 1, c), -g);
      }
      if ((sel & 4u) && hy) {
        const float g = b.coef[2] * my * smooth_l1_grad(d2) / 3.0f;
        g0 += g;
        if (g != 0.f) atomicAdd(GW(p + a.W, c), -g);
      }
      if (g0 != 0.f) atomicAdd(GW(p, c), g0);
    }
  }
  float w_mu[3] = {0, 0, 0}, w_sq[3] = {0, 0, 0}, wr[3] = {0, 0, 0}, mpool = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int64_t q = (int64_t)yy[j] * a.W + xx[k];
      const f32x4 t = w[q], r = ref[q];
      mpool += (t[3] > 0.5f && a.mask[q] > 0.5f) ? 1.0f : 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { w_mu[c] += t[c]; w_sq[c] = fmaf(t[c], t[c], w_sq[c]); wr[c] = fmaf(t[c], r[c], wr[c]); }
    }
  mpool /= 9.0f;
  float ssim = 0.f;
  float dmu[3], dsq[3], dwr[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mux = w_mu[c] / 9.0f, muy = r_mu[c] / 9.0f;
    const float sgx = w_sq[c] / 9.0f - mux * mux, sgy = r_sq[c] / 9.0f - muy * muy, sgxy = wr[c] / 9.0f - mux * muy;
    const float A = 2.0f * mux * muy + 1e-4f, B = 2.0f * sgxy + 9e-4f;
    const float Cc = mux * mux + muy * muy + 1e-4f, Dd = sgx + sgy + 9e-4f;
    const float n = A * B, d = Cc * Dd;
    const float f = (1.0f - n / d) / 2.0f;
    ssim += mpool * fminf(fmaxf(f, 0.0f), 1.0f);
    dmu[c] = dsq[c] = dwr[c] = 0.f;
    if (BWD && (sel & 8u) && f > 0.0f && f < 1.0f) {
      const float up = b.coef[3] * mref * mpool / 3.0f;                 // d loss / d f
      const float fn = -0.5f / d, fd = 0.5f * n / (d * d);              // d f / d n, d f / d d
      // in the window sums' variables: mux = w_mu / 9, q = w_sq / 9, z = wr / 9
      dmu[c] = up * (fn * (2.0f * muy * B - 2.0f * muy * A) + fd * (2.0f * mux * Dd - 2.0f * mux * Cc));
      dsq[c] = up * fd * Cc;
      dwr[c] = up * fn * 2.0f * A;
    }
  }
  if (BWD && (sel & 8u)) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int64_t q = (int64_t)yy[j] * a.W + xx[k];
        const f32x4 t = w[q], r = ref[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float g = (dmu[c] + 2.0f * t[c] * dsq[c] + r[c] * dwr[c]) / 9.0f;
          if (g != 0.f) atomicAdd(GW(q, c), g);
        }
      }
  }
  v4[0] = l1 / 3.0f; v4[1] = gx / 3.0f; v4[2] = gy / 3.0f; v4[3] = ssim / 3.0f;
#undef GW
}

__global__ __launch_bounds__(256, 2) void ptloss_bwd_terms_kernel(PtArgs a, PtBwd b) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)a.H * a.W;
  if (p >= per) return;
  const int y = (int)(p / a.W), x = (int)(p % a.W);
  const f32x4* __restrict__ ref = reinterpret_cast<const f32x4*>(a.imgs) + (int64_t)a.ref * per;
  const float mref = a.mask[p];
  const float mx = x + 1 < a.W ? mref * a.mask[p + 1] : 0.f;
  const float my = y + 1 < a.H ? mref * a.mask[p + a.W] : 0.f;
  if (mref == 0.f) return;                       // every term of this pixel carries a factor mref
  float r_mu[3] = {0, 0, 0}, r_sq[3] = {0, 0, 0};
  int yy[3], xx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { yy[k] = reflect(y + k - 1, a.H); xx[k] = reflect(x + k - 1, a.W); }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x4 r = ref[(int64_t)yy[j] * a.W + xx[k]];
#pragma unroll
      for (int c = 0; c < 3; ++c) { r_mu[c] += r[c]; r_sq[c] = fmaf(r[c], r[c], r_sq[c]); }
    }
  float best[4][SURF_MAX_VIEWS];
  int bidx[4][SURF_MAX_VIEWS];
  for (int s = 0; s < a.ns; ++s) {
    float v4[4];
    pt_source<false>(a, b, s, p, x, y, yy, xx, r_mu, r_sq, mref, mx, my, 0u, v4);
    for (int t = 0; t < 4; ++t) {
      int j = s;
      while (j > 0 && best[t][j - 1] > v4[t]) { best[t][j] = best[t][j - 1]; bidx[t][j] = bidx[t][j - 1]; --j; }
      best[t][j] = v4[t];
      bidx[t][j] = s;
    }
  }
  for (int s = 0; s < a.ns; ++s) {
    unsigned sel = 0;
    for (int t = 0; t < 4; ++t)
      for (int k = 0; k < a.topk; ++k)
        if (bidx[t][k] == s) sel |= 1u << t;
    if (!sel) continue;
    float v4[4];
    pt_source<true>(a, b, s, p, x, y, yy, xx, r_mu, r_sq, mref, mx, my, sel, v4);
  }
}

__global__ __launch_bounds__(256) void ptloss_bwd_depth_kernel(PtArgs a, PtBwd b) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)a.H * a.W;
  if (p >= per) return;
  const int y = (int)(p / a.W), x = (int)(p % a.W);
  const float d = a.depth[p];
  // direction of the pixel's ray in the reference camera and the world (the depth-linear part)
  const float rx = a.Kinv[0] * (float)x + a.Kinv[1] * (float)y + a.Kinv[2];
  const float ry = a.Kinv[3] * (float)x + a.Kinv[4] * (float)y + a.Kinv[5];
  const float rz = a.Kinv[6] * (float)x + a.Kinv[7] * (float)y + a.Kinv[8];
  const float ax = a.c2w[0] * rx + a.c2w[1] * ry + a.c2w[2] * rz;
  const float ay = a.c2w[4] * rx + a.c2w[5] * ry + a.c2w[6] * rz;
  const float az = a.c2w[8] * rx + a.c2w[9] * ry + a.c2w[10] * rz;
  const float wx = ax * d + a.c2w[3], wy = ay * d + a.c2w[7], wz = az * d + a.c2w[11];
  float acc = 0.f;
  for (int s = 0; s < a.ns; ++s) {
    const float* gs = b.g_warp + (int64_t)s * per * 4;                    // channel planes (see pt_source)
    const float g[3] = {gs[p], gs[per + p], gs[2 * per + p]};
    if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) continue;
    const float* M = a.w2c[s];
    const float* K = a.K[s];
    const float sx = M[0] * wx + M[1] * wy + M[2] * wz + M[3], sy = M[4] * wx + M[5] * wy + M[6] * wz + M[7],
                sz = M[8] * wx + M[9] * wy + M[10] * wz + M[11];
    const float tx = M[0] * ax + M[1] * ay + M[2] * az, ty = M[4] * ax + M[5] * ay + M[6] * az, tz = M[8] * ax + M[9] * ay + M[10] * az;
    const float px = K[0] * sx + K[1] * sy + K[2] * sz, py = K[3] * sx + K[4] * sy + K[5] * sz, pz = K[6] * sx + K[7] * sy + K[8] * sz;
    const float qx = K[0] * tx + K[1] * ty + K[2] * tz, qy = K[3] * tx + K[4] * ty + K[5] * tz, qz = K[6] * tx + K[7] * ty + K[8] * tz;
    const float den = pz + 1e-8f;
    const float u = px / den, v = py / den;
    const float du = (qx * den - px * qz) / (den * den), dv = (qy * den - py * qz) / (den * den);
    // the forward's sampling position: gx = ((u / ((W-1)/2) - 1 + 1) / 2) (W-1) = u up to rounding; same for v
    const float nx = u / ((float)(a.W - 1) / 2.0f) - 1.0f, ny = v / ((float)(a.H - 1) / 2.0f) - 1.0f;
    const float gxp = ((nx + 1.0f) / 2.0f) * (float)(a.W - 1), gyp = ((ny + 1.0f) / 2.0f) * (float)(a.H - 1);
    const float fx = floorf(gxp), fy = floorf(gyp);
    const float lx = gxp - fx, ly = gyp - fy;
    const int x0 = (int)fx, y0 = (int)fy;
    const float* img = a.imgs + (int64_t)a.view[s] * per * 4;
    float dgx = 0.f, dgy = 0.f;
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        const int xi = x0 + dxx, yi = y0 + dyy;
        if ((xi >= 0) & (xi < a.W) & (yi >= 0) & (yi < a.H)) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(img + ((int64_t)yi * a.W + xi) * 4);
          const float gt = g[0] * t[0] + g[1] * t[1] + g[2] * t[2];
          dgx += gt * (dxx ? 1.0f : -1.0f) * (dyy ? ly : 1.0f - ly);
          dgy += gt * (dyy ? 1.0f : -1.0f) * (dxx ? lx : 1.0f - lx);
        }
      }
    acc += dgx * du + dgy * dv;
  }
  b.g_depth[p] = acc;
}

}  // namespace

static int fill_pt_args(PtArgs& a, int nv, int ref_idx, const float* h_intrs, const float* h_c2w, const float* h_w2c);

extern "C" int surf_ptloss_terms(const float* imgs_t4, int nv, int H, int W, const float* depth, const float* mask, int ref_idx,
                                 int topk, const float* h_intrs, const float* h_c2w, const float* h_w2c, float* warp,
                                 float* terms, void* stream) {
  if (!imgs_t4 || !depth || !mask || !h_intrs || !h_c2w || !h_w2c || !warp || !terms) return SURF_E_ARG;
  if (nv < 2 || nv > SURF_MAX_VIEWS || ref_idx < 0 || ref_idx >= nv || H < 2 || W < 2) return SURF_E_ARG;
  if (topk < 1 || topk > nv - 1) return SURF_E_ARG;
  PtArgs a;
  a.imgs = imgs_t4; a.depth = depth; a.mask = mask; a.nv = nv; a.ref = ref_idx; a.H = H; a.W = W; a.ns = nv - 1; a.topk = topk;
  a.warp = warp; a.terms = terms;
  if (const int rc = fill_pt_args(a, nv, ref_idx, h_intrs, h_c2w, h_w2c)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)H * W;
  hipLaunchKernelGGL(ptloss_warp_kernel, dim3((unsigned)((per * a.ns + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ptloss_terms_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, a);
  return surf_check_launch();
}

static int fill_pt_args(PtArgs& a, int nv, int ref_idx, const float* h_intrs, const float* h_c2w, const float* h_w2c) {
  // inverse of the reference intrinsics' upper-left 3x3 (host, double)
  {
    const float* K = h_intrs + ref_idx * 16;
    const double m[9] = {K[0], K[1], K[2], K[4], K[5], K[6], K[8], K[9], K[10]};
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (det == 0.0) return SURF_E_ARG;
    const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                           (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                           (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
    for (int k = 0; k < 9; ++k) a.Kinv[k] = (float)inv[k];
  }
  for (int k = 0; k < 12; ++k) a.c2w[k] = h_c2w[ref_idx * 16 + k];
  int slot = 0;
  for (int v = 0; v < nv; ++v) {
    if (v == ref_idx) continue;
    a.view[slot] = v;
    for (int k = 0; k < 12; ++k) a.w2c[slot][k] = h_w2c[v * 16 + k];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) a.K[slot][r * 3 + c] = h_intrs[v * 16 + r * 4 + c];
    ++slot;
  }
  for (; slot < SURF_MAX_VIEWS; ++slot) {
    a.view[slot] = 0;
    for (int k = 0; k < 12; ++k) a.w2c[slot][k] = 0.f;
    for (int k = 0; k < 9; ++k) a.K[slot][k] = 0.f;
  }
  return 0;
}

extern "C" int surf_ptloss_backward(const float* imgs_t4, int nv, int H, int W, const float* depth, const float* mask, int ref_idx,
                                    int topk, const float* h_intrs, const float* h_c2w, const float* h_w2c, const float* warp,
                                    const float* coef, float* g_warp, float* g_depth, void* stream) {
  if (!imgs_t4 || !depth || !mask || !h_intrs || !h_c2w || !h_w2c || !warp || !coef || !g_warp || !g_depth) return SURF_E_ARG;
  if (nv < 2 || nv > SURF_MAX_VIEWS || ref_idx < 0 || ref_idx >= nv || H < 2 || W < 2) return SURF_E_ARG;
  if (topk < 1 || topk > nv - 1) return SURF_E_ARG;
  PtArgs a;
  a.imgs = imgs_t4; a.depth = depth; a.mask = mask; a.nv = nv; a.ref = ref_idx; a.H = H; a.W = W; a.ns = nv - 1; a.topk = topk;
  a.warp = const_cast<float*>(warp); a.terms = nullptr;
  if (const int rc = fill_pt_args(a, nv, ref_idx, h_intrs, h_c2w, h_w2c)) return rc;
  PtBwd b;
  b.coef = coef; b.g_warp = g_warp; b.g_depth = g_depth;
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)H * W;
  const hipError_t e = hipMemsetAsync(g_warp, 0, per * a.ns * 4 * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(ptloss_bwd_terms_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, a, b);
  hipLaunchKernelGGL(ptloss_bwd_depth_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, a, b);
  return surf_check_launch();
}
