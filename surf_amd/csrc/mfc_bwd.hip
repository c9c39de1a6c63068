// K15b  backward of the multi-view feature-consistency term (mfc_loss, losses/loss.py:43-45) w.r.t. the SDF.
// The patches of surface_patch_warp2 depend on the network only through the surface point p = o + d z0 of each ray (the
// normal and the feature maps are detached, implicit_surface.py:224-235), i.e. through ONE scalar per ray.  So
//     d ncc / d z0  =  the forward-mode tangent of (patch warp -> LNCC) along the ray direction:
//   patch_tangent_kernel  the patches again + d(patch value)/d z0: bilinear spatial gradient x d(sample position)/d z0, with
//                         the reference pixel and the plane-induced homography (its n . p term) differentiated in closed form
//   lncc_jvp_kernel       d ncc / d z0 from patches and tangents (the sums of compute_LNCC2 are linear in the patch values)
//   crossing_bwd_kernel   z0 = (s1 z2 - s2 z1) / (s1 - s2 + 1e-10) at the first sign change: adds g_z0 dz0/ds to d_sdf of the
//                         two bracketing samples (zero where the crossing is absent or z0 was clamped to 0)
// Restates the autograd of projector.py:560-645, losses/ncc.py:7-51 and implicit_surface.py:181-220 under loss.backward().
#include <math.h>

#include "common.h"

namespace {

struct TanArgs {
  const float* pts;    // (R,3)
  const float* dirs;   // (R,3) d pts / d z0
  const float* grads;  // (R,3)
  const float* maps[3];
  int R, nv, H, W, patch;
  float K[SURF_MAX_VIEWS][9], Kinv0[9], Rm[SURF_MAX_VIEWS][9], t[SURF_MAX_VIEWS][3];
  float* ref_out; float* src_out;   // values   (1,R,P,12), (nv-1,R,P,12)
  float* ref_tan; float* src_tan;   // tangents, same shapes
};

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

// bilinear fetch (zeros) with its gradient w.r.t. the pixel coordinates
__device__ __forceinline__ void bilinear_grad(const float* __restrict__ map, int H, int W, float x, float y, f32x4& v, f32x4& dx, f32x4& dy) {
  const float fx = floorf(x), fy = floorf(y);
  const float tx = x - fx, ty = y - fy;
  const int x0 = (int)fx, y0 = (int)fy;
  f32x4 c[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int xi = x0 + i, yi = y0 + j;
      const bool ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
      c[j][i] = ok ? *reinterpret_cast<const f32x4*>(map + ((int64_t)yi * W + xi) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  v = (c[0][0] * (1.0f - tx) + c[0][1] * tx) * (1.0f - ty) + (c[1][0] * (1.0f - tx) + c[1][1] * tx) * ty;
  dx = (c[0][1] - c[0][0]) * (1.0f - ty) + (c[1][1] - c[1][0]) * ty;
  dy = (c[1][0] - c[0][0]) * (1.0f - tx) + (c[1][1] - c[0][1]) * tx;
}

__global__ __launch_bounds__(256) void patch_tangent_kernel(TanArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)a.R * a.nv) return;
  const int ray = (int)(wid / a.nv), view = (int)(wid % a.nv);
  const int P = a.patch * a.patch, hp = a.patch / 2;
  const float px = a.pts[ray * 3 + 0], py = a.pts[ray * 3 + 1], pz = a.pts[ray * 3 + 2];
  const float vx = a.dirs[ray * 3 + 0], vy = a.dirs[ray * 3 + 1], vz = a.dirs[ray * 3 + 2];
  const float* R0 = a.Rm[0];
  float pr[3], prd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float rot = R0[0 * 3 + i] * px + R0[1 * 3 + i] * py + R0[2 * 3 + i] * pz;
    const float tt = -(R0[0 * 3 + i] * a.t[0][0] + R0[1 * 3 + i] * a.t[0][1] + R0[2 * 3 + i] * a.t[0][2]);
    pr[i] = rot + tt;
    prd[i] = R0[0 * 3 + i] * vx + R0[1 * 3 + i] * vy + R0[2 * 3 + i] * vz;
  }
  const float* K0 = a.K[0];
  const float qx = K0[0] * pr[0] + K0[1] * pr[1] + K0[2] * pr[2], qxd = K0[0] * prd[0] + K0[1] * prd[1] + K0[2] * prd[2];
  const float qy = K0[3] * pr[0] + K0[4] * pr[1] + K0[5] * pr[2], qyd = K0[3] * prd[0] + K0[4] * prd[1] + K0[5] * prd[2];
  const float qz = K0[6] * pr[0] + K0[7] * pr[1] + K0[8] * pr[2], qzd = K0[6] * prd[0] + K0[7] * prd[1] + K0[8] * prd[2];
  const float pix_x = qx / (qz + 1e-8f), pix_y = qy / (qz + 1e-8f);
  const float pix_xd = (qxd - pix_x * qzd) / (qz + 1e-8f), pix_yd = (qyd - pix_y * qzd) / (qz + 1e-8f);
  float Hm[9], Hd[9];
  if (view > 0) {
    float g[3] = {a.grads[ray * 3 + 0], a.grads[ray * 3 + 1], a.grads[ray * 3 + 2]};
    float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (gn <= 0.f) gn = 1e-8f;
    g[0] /= gn; g[1] /= gn; g[2] /= gn;
    float nc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) nc[i] = R0[0 * 3 + i] * g[0] + R0[1 * 3 + i] * g[1] + R0[2 * 3 + i] * g[2];
    const float disp = nc[0] * pr[0] + nc[1] * pr[1] + nc[2] * pr[2];
    const float dispd = nc[0] * prd[0] + nc[1] * prd[1] + nc[2] * prd[2];
    const float* Rj = a.Rm[view];
    float RsT[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) RsT[i * 3 + j] = Rj[j * 3 + i];
    float Rrel[9];
    mat3_mul(RsT, R0, Rrel);
    const float cr[3] = {a.t[0][0] - a.t[view][0], a.t[0][1] - a.t[view][1], a.t[0][2] - a.t[view][2]};
    float tv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tv[i] = RsT[i * 3 + 0] * cr[0] + RsT[i * 3 + 1] * cr[1] + RsT[i * 3 + 2] * cr[2];
    float M[9], Md[9];
    const float den = disp + 1e-10f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        M[i * 3 + j] = Rrel[i * 3 + j] + (tv[i] * nc[j]) / den;
        Md[i * 3 + j] = -(tv[i] * nc[j]) * dispd / (den * den);
      }
    float M2[9];
    mat3_mul(a.K[view], M, M2);
    mat3_mul(M2, a.Kinv0, Hm);
    mat3_mul(a.K[view], Md, M2);
    mat3_mul(M2, a.Kinv0, Hd);
  }
  for (int p = lane; p < P; p += 64) {
    const float ux = pix_x + (float)(p % a.patch - hp), uy = pix_y + (float)(p / a.patch - hp);
    float sx, sy, sxd, syd;
    float *dst, *dtan;
    if (view == 0) {
      sx = ux; sy = uy; sxd = pix_xd; syd = pix_yd;
      dst = a.ref_out + ((int64_t)ray * P + p) * 12;
      dtan = a.ref_tan + ((int64_t)ray * P + p) * 12;
    } else {
      const float hx = Hm[0] * ux + Hm[1] * uy + Hm[2], hy = Hm[3] * ux + Hm[4] * uy + Hm[5], hz = Hm[6] * ux + Hm[7] * uy + Hm[8];
      const float hxd = Hd[0] * ux + Hd[1] * uy + Hd[2] + Hm[0] * pix_xd + Hm[1] * pix_yd;
      const float hyd = Hd[3] * ux + Hd[4] * uy + Hd[5] + Hm[3] * pix_xd + Hm[4] * pix_yd;
      const float hzd = Hd[6] * ux + Hd[7] * uy + Hd[8] + Hm[6] * pix_xd + Hm[7] * pix_yd;
      sx = hx / (hz + 1e-8f); sy = hy / (hz + 1e-8f);
      sxd = (hxd - sx * hzd) / (hz + 1e-8f); syd = (hyd - sy * hzd) / (hz + 1e-8f);
      dst = a.src_out + (((int64_t)(view - 1) * a.R + ray) * P + p) * 12;
      dtan = a.src_tan + (((int64_t)(view - 1) * a.R + ray) * P + p) * 12;
    }
    // the forward's normalise / un-normalise round trip (align_corners=True) is the identity on the pixel coordinates
    const float gx = 2.0f * sx / (float)(a.W - 1) - 1.0f, gy = 2.0f * sy / (float)(a.H - 1) - 1.0f;
    const float x = (gx + 1.0f) / 2.0f * (float)(a.W - 1), y = (gy + 1.0f) / 2.0f * (float)(a.H - 1);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      f32x4 v, dx, dy;
      bilinear_grad(a.maps[l] + (int64_t)view * a.H * a.W * 4, a.H, a.W, x, y, v, dx, dy);
      *reinterpret_cast<f32x4*>(dst + 4 * l) = v;
      *reinterpret_cast<f32x4*>(dtan + 4 * l) = dx * sxd + dy * syd;
    }
  }
}

// d ncc / d z0 per ray (lane = (element slice, channel) as in lncc.hip)
__global__ __launch_bounds__(256) void lncc_jvp_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                                                       const float* __restrict__ reft, const float* __restrict__ srct, int64_t R,
                                                       int nsrc, int P, int C, float* __restrict__ out_ncc, float* __restrict__ out_d) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int nsl = 64 / C;
  const int ch = lane % C, sl = lane / C;
  const bool act = sl < nsl;
  const float* __restrict__ rp = ref + ray * (int64_t)P * C;
  const float* __restrict__ rt = reft + ray * (int64_t)P * C;
  float r1 = 0.f, r2 = 0.f, r1d = 0.f, r2d = 0.f;
  if (act)
    for (int e = sl; e < P; e += nsl) {
      const float v = rp[e * C + ch], d = rt[e * C + ch];
      r1 += v; r2 = fmaf(v, v, r2); r1d += d; r2d = fmaf(2.0f * v, d, r2d);
    }
  float best0 = 3.0e38f, best1 = 3.0e38f, d0 = 0.f, d1 = 0.f;
  for (int v = 0; v < nsrc; ++v) {
    const float* __restrict__ sp = src + ((int64_t)v * R + ray) * (int64_t)P * C;
    const float* __restrict__ st = srct + ((int64_t)v * R + ray) * (int64_t)P * C;
    float s1 = 0.f, s2 = 0.f, rs = 0.f, s1d = 0.f, s2d = 0.f, rsd = 0.f;
    if (act)
      for (int e = sl; e < P; e += nsl) {
        const float a = rp[e * C + ch], ad = rt[e * C + ch], b = sp[e * C + ch], bd = st[e * C + ch];
        s1 += b; s2 = fmaf(b, b, s2); rs = fmaf(a, b, rs);
        s1d += bd; s2d = fmaf(2.0f * b, bd, s2d); rsd += ad * b + a * bd;
      }
    float T[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float mine[10] = {r1, r2, s1, s2, rs, r1d, r2d, s1d, s2d, rsd};
    for (int k = 0; k < nsl; ++k) {
      const int from = ch + k * C;
#pragma unroll
      for (int q = 0; q < 10; ++q) T[q] += __shfl(mine[q], from);
    }
    const float n = (float)P;
    const float ur = T[0] / n, us = T[2] / n;
    const float cross = T[4] - us * T[0] - ur * T[2] + ur * us * n;
    const float rvar = T[1] - 2.0f * ur * T[0] + ur * ur * n;
    const float svar = T[3] - 2.0f * us * T[2] + us * us * n;
    const float D = rvar * svar + 1e-5f;
    const float cc = cross * cross / D;
    const float crossd = T[9] - (T[5] * T[2] + T[0] * T[7]) / n;
    const float rvard = T[6] - 2.0f * T[0] * T[5] / n, svard = T[8] - 2.0f * T[2] * T[7] / n;
    const float ccd = (2.0f * cross * crossd * D - cross * cross * (rvard * svar + rvar * svard)) / (D * D);
    const float raw = 1.0f - cc;
    const float ncc = fminf(fmaxf(raw, 0.0f), 2.0f);
    const float nccd = (raw > 0.0f && raw < 2.0f) ? -ccd : 0.0f;
    float m = lane < C ? ncc : 0.f, md = lane < C ? nccd : 0.f;
    m = wave_sum(m) / (float)C;
    md = wave_sum(md) / (float)C;
    if (m < best0) { best1 = best0; d1 = d0; best0 = m; d0 = md; }
    else if (m < best1) { best1 = m; d1 = md; }
  }
  if (lane == 0) {
    if (out_ncc) out_ncc[ray] = 0.5f * (best0 + best1);
    out_d[ray] = 0.5f * (d0 + d1);
  }
}

__global__ __launch_bounds__(256) void crossing_bwd_kernel(const float* __restrict__ sdf, const uint8_t* __restrict__ vmask,
                                                           const float* __restrict__ mid_z, int R, int S, const float* __restrict__ zmax,
                                                           const float* __restrict__ g_z0, float* __restrict__ d_sdf) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= R) return;
  const float g = g_z0[ray];
  if (g == 0.f) return;
  const int64_t base = (int64_t)ray * S;
  for (int k = 0; k + 1 < S; ++k) {
    if (!vmask[base + k] || !vmask[base + k + 1]) continue;
    const float s1 = sdf[base + k], s2 = sdf[base + k + 1];
    if (s1 * s2 > 0.f) continue;
    const float z1 = mid_z[base + k], z2 = mid_z[base + k + 1];
    const float den = s1 - s2 + 1e-10f, num = s1 * z2 - s2 * z1;
    const float z0 = num / den;
    if (z0 < 0.f || z0 > *zmax) return;               // clamped to 0 in the forward: no gradient (implicit_surface.py:217-219)
    d_sdf[base + k] += g * (z2 * den - num) / (den * den);
    d_sdf[base + k + 1] += g * (-z1 * den + num) / (den * den);
    return;
  }
}

}  // namespace

extern "C" int surf_patch_warp_tangent(const float* pts, const float* dirs, const float* grads, int n_rays, const float* const* h_maps,
                                       int nv, int H, int W, const float* h_intrs, const float* h_kinv_ref, const float* h_c2w,
                                       int patch_size, float* ref_out, float* src_out, float* ref_tan, float* src_tan, void* stream) {
  if (!pts || !dirs || !grads || !h_maps || !h_intrs || !h_kinv_ref || !h_c2w || !ref_out || !src_out || !ref_tan || !src_tan)
    return SURF_E_ARG;
  if (n_rays <= 0 || nv < 2 || H < 2 || W < 2 || patch_size < 1 || (patch_size & 1) == 0) return SURF_E_ARG;
  if (nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  TanArgs a;
  a.pts = pts; a.dirs = dirs; a.grads = grads; a.R = n_rays; a.nv = nv; a.H = H; a.W = W; a.patch = patch_size;
  a.ref_out = ref_out; a.src_out = src_out; a.ref_tan = ref_tan; a.src_tan = src_tan;
  for (int l = 0; l < 3; ++l) {
    if (!h_maps[l]) return SURF_E_ARG;
    a.maps[l] = h_maps[l];
  }
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int s = v < nv ? v : 0;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        a.K[v][r * 3 + c] = h_intrs[s * 16 + r * 4 + c];
        a.Rm[v][r * 3 + c] = h_c2w[s * 16 + r * 4 + c];
      }
      a.t[v][r] = h_c2w[s * 16 + r * 4 + 3];
    }
  }
  for (int i = 0; i < 9; ++i) a.Kinv0[i] = h_kinv_ref[i];
  const int64_t waves = (int64_t)n_rays * nv;
  hipLaunchKernelGGL(patch_tangent_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int surf_lncc_jvp(const float* ref, const float* src, const float* ref_tan, const float* src_tan, int64_t n_rays, int n_src,
                             int patch_elems, int channels, float* ncc, float* dncc, void* stream) {
  if (!ref || !src || !ref_tan || !src_tan || !dncc || n_rays <= 0) return SURF_E_ARG;
  if (n_src < 2 || patch_elems < 1 || channels < 1 || channels > 64) return SURF_E_ARG;
  hipLaunchKernelGGL(lncc_jvp_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, src, ref_tan,
                     src_tan, n_rays, n_src, patch_elems, channels, ncc, dncc);
  return surf_check_launch();
}

extern "C" int surf_crossing_backward(const float* sdf, const uint8_t* vmask, const float* mid_z, int n_rays, int S, const float* zmax,
                                      const float* g_z0, float* d_sdf, void* stream) {
  if (!sdf || !vmask || !mid_z || !zmax || !g_z0 || !d_sdf || n_rays <= 0 || S < 2) return SURF_E_ARG;
  hipLaunchKernelGGL(crossing_bwd_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, sdf, vmask, mid_z, n_rays, S,
                     zmax, g_z0, d_sdf);
  return surf_check_launch();
}
