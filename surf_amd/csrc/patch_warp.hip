// a15: plane-induced homography patches of the multi-level feature maps around every ray's SDF zero crossing, the inputs
// of the LNCC loss (training only; `validate` ignores them).
//
// Restates  render_core's tail          implicit_surface.py:217-245   (z clamp, surface point, feature stack)
//           surface_patch_warp2         projector.py:560-627
//           patch_homography            projector.py:630-645
//
// Kernels (all HBM / L2-gather bound, a few hundred thousand threads per 512-ray training batch):
//   upsample_t4     F.interpolate(mode="bilinear", align_corners=False) of a texel4 map to the finest level's size
//                   (the reference stacks FPN levels 0, 1, 2 at full resolution before sampling, :231-235)
//   surface_points  z0 -> 0 outside [0, max(z_vals)] (:217-219, the maximum is over the whole batch), p = o + d z0
//   patch_warp      one wavefront per (ray, view): lanes = the 121 patch pixels (two per lane); reference view samples the
//                   stack at pixel + offset, source views at Hom (pixel + offset, 1) with
//                   Hom = K_src (R_rel + (R_src C_rel) n^T / (n . p_ref + 1e-10)) K_ref^-1, bilinear, align_corners=True, zeros
#include <math.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void upsample_t4_kernel(const float* __restrict__ src, int n, int h, int w, int H, int W,
                                                          float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * H * W;
  if (i >= total) return;
  const int x = (int)(i % W), y = (int)((i / W) % H), v = (int)(i / ((int64_t)W * H));
  // ATen area_pixel_compute_source_index (align_corners = false, scale = in / out), upsample_bilinear2d
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
  if (fy < 0.f) fy = 0.f;
  if (fx < 0.f) fx = 0.f;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
  const float* base = src + (int64_t)v * h * w * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((int64_t)y0 * w + x0) * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(base + ((int64_t)y0 * w + x1) * 4);
  const f32x4 c = *reinterpret_cast<const f32x4*>(base + ((int64_t)y1 * w + x0) * 4);
  const f32x4 d = *reinterpret_cast<const f32x4*>(base + ((int64_t)y1 * w + x1) * 4);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = hy * (hx * a[k] + lx * b[k]) + ly * (hx * c[k] + lx * d[k]);
  *reinterpret_cast<f32x4*>(dst + i * 4) = o;
}

// order-preserving float <-> uint map for atomicMax on floats
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void zmax_init_kernel(unsigned* zmax) { *zmax = f2ord(-INFINITY); }
__global__ __launch_bounds__(256) void zmax_kernel(const float* __restrict__ z, int64_t n, unsigned* __restrict__ zmax) {
  float m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, z[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(zmax, f2ord(m));
}
__global__ __launch_bounds__(256) void surface_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const float* __restrict__ z0, int n, const unsigned* __restrict__ zmax,
                                                             float* __restrict__ pts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float z = z0[i];
  if (z < 0.f) z = 0.f;                 // implicit_surface.py:217
  if (z > ord2f(*zmax)) z = 0.f;        // :218-219
#pragma unroll
  for (int c = 0; c < 3; ++c) pts[i * 3 + c] = rays_o[i * 3 + c] + rays_d[i * 3 + c] * z;
}

struct WarpArgs {
  const float* pts;    // (R,3) surface points
  const float* grads;  // (R,3) raw SDF gradients at those points
  const float* maps[3];  // texel4 (nv,H,W,4) at full resolution: levels 0, 1, 2
  int R, nv, H, W, patch;
  float K[SURF_MAX_VIEWS][9], Kinv0[9], Rm[SURF_MAX_VIEWS][9], t[SURF_MAX_VIEWS][3];
  float* ref_out;  // (1, R, P, 12)
  float* src_out;  // (nv-1, R, P, 12)
};

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {  // C = A B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

// F.grid_sample(bilinear, zeros, align_corners=True) of the three stacked levels at normalised (gx, gy)
__device__ __forceinline__ void sample12(const WarpArgs& a, int view, float gx, float gy, float* out12) {
  const float x = (gx + 1.0f) / 2.0f * (float)(a.W - 1), y = (gy + 1.0f) / 2.0f * (float)(a.H - 1);
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const f32x4 v = bilinear_texel4(a.maps[l] + (int64_t)view * a.H * a.W * 4, a.H, a.W, x, y);
    out12[4 * l + 0] = v[0]; out12[4 * l + 1] = v[1]; out12[4 * l + 2] = v[2]; out12[4 * l + 3] = v[3];
  }
}

__global__ __launch_bounds__(256) void patch_warp_kernel(WarpArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)a.R * a.nv) return;
  const int ray = (int)(wid / a.nv), view = (int)(wid % a.nv);
  const int P = a.patch * a.patch, hp = a.patch / 2;
  const float px = a.pts[ray * 3 + 0], py = a.pts[ray * 3 + 1], pz = a.pts[ray * 3 + 2];
  // reference camera frame: R0^T p - R0^T t0 (projector.py:575-578)
  const float* R0 = a.Rm[0];
  float pr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float rot = R0[0 * 3 + i] * px + R0[1 * 3 + i] * py + R0[2 * 3 + i] * pz;
    const float tt = -(R0[0 * 3 + i] * a.t[0][0] + R0[1 * 3 + i] * a.t[0][1] + R0[2 * 3 + i] * a.t[0][2]);
    pr[i] = rot + tt;
  }
  const float* K0 = a.K[0];
  const float qx = K0[0] * pr[0] + K0[1] * pr[1] + K0[2] * pr[2];
  const float qy = K0[3] * pr[0] + K0[4] * pr[1] + K0[5] * pr[2];
  const float qz = K0[6] * pr[0] + K0[7] * pr[1] + K0[8] * pr[2];
  const float pix_x = qx / (qz + 1e-8f), pix_y = qy / (qz + 1e-8f);
  float Hm[9];
  if (view > 0) {
    // unit normal in the reference camera frame (implicit_surface.py:224-228)
    float g[3] = {a.grads[ray * 3 + 0], a.grads[ray * 3 + 1], a.grads[ray * 3 + 2]};
    float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (gn <= 0.f) gn = 1e-8f;
    g[0] /= gn; g[1] /= gn; g[2] /= gn;
    float nc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) nc[i] = R0[0 * 3 + i] * g[0] + R0[1 * 3 + i] * g[1] + R0[2 * 3 + i] * g[2];
    const float disp = nc[0] * pr[0] + nc[1] * pr[1] + nc[2] * pr[2];
    const float* Rj = a.Rm[view];
    float RsT[9];  // R_src = Rj^T
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
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rrel[i * 3 + j] + (tv[i] * nc[j]) / (disp + 1e-10f);
    float M2[9];
    mat3_mul(a.K[view], M, M2);
    mat3_mul(M2, a.Kinv0, Hm);
  }
  for (int p = lane; p < P; p += 64) {
    const float ux = pix_x + (float)(p % a.patch - hp), uy = pix_y + (float)(p / a.patch - hp);
    float out12[12];
    float* dst;
    if (view == 0) {
      const float gx = 2.0f * ux / (float)(a.W - 1) - 1.0f, gy = 2.0f * uy / (float)(a.H - 1) - 1.0f;
      sample12(a, 0, gx, gy, out12);
      dst = a.ref_out + ((int64_t)ray * P + p) * 12;
    } else {
      const float hx = Hm[0] * ux + Hm[1] * uy + Hm[2], hy = Hm[3] * ux + Hm[4] * uy + Hm[5], hz = Hm[6] * ux + Hm[7] * uy + Hm[8];
      const float sx = hx / (hz + 1e-8f), sy = hy / (hz + 1e-8f);
      const float gx = 2.0f * sx / (float)(a.W - 1) - 1.0f, gy = 2.0f * sy / (float)(a.H - 1) - 1.0f;
      sample12(a, view, gx, gy, out12);
      dst = a.src_out + (((int64_t)(view - 1) * a.R + ray) * P + p) * 12;
    }
#pragma unroll
    for (int g4 = 0; g4 < 3; ++g4)
      *reinterpret_cast<f32x4*>(dst + 4 * g4) = f32x4{out12[4 * g4], out12[4 * g4 + 1], out12[4 * g4 + 2], out12[4 * g4 + 3]};
  }
}

}  // namespace

extern "C" int surf_upsample_bilinear_t4(const float* src, int n, int h, int w, int H, int W, float* dst, void* stream) {
  if (!src || !dst || n <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SURF_E_ARG;
  const int64_t total = (int64_t)n * H * W;
  hipLaunchKernelGGL(upsample_t4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n, h, w, H,
                     W, dst);
  return surf_check_launch();
}

extern "C" int surf_surface_points(const float* rays_o, const float* rays_d, const float* z_sdf0, int n_rays, const float* z_vals,
                                   int64_t n_z, unsigned* workspace, float* pts, void* stream) {
  if (!rays_o || !rays_d || !z_sdf0 || !z_vals || !workspace || !pts || n_rays <= 0 || n_z <= 0) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zmax_init_kernel, dim3(1), dim3(1), 0, st, workspace);
  const int64_t nb = (n_z + 255) / 256;
  hipLaunchKernelGGL(zmax_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, st, z_vals, n_z, workspace);
  hipLaunchKernelGGL(surface_points_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, st, rays_o, rays_d, z_sdf0, n_rays, workspace,
                     pts);
  return surf_check_launch();
}

extern "C" int surf_patch_warp(const float* pts, const float* grads, int n_rays, const float* const* h_maps, int nv, int H, int W,
                               const float* h_intrs, const float* h_kinv_ref, const float* h_c2w, int patch_size, float* ref_out,
                               float* src_out, void* stream) {
  if (!pts || !grads || !h_maps || !h_intrs || !h_kinv_ref || !h_c2w || !ref_out || !src_out) return SURF_E_ARG;
  if (n_rays <= 0 || nv < 2 || H < 2 || W < 2 || patch_size < 1 || (patch_size & 1) == 0) return SURF_E_ARG;
  if (nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  WarpArgs a;
  a.pts = pts; a.grads = grads; a.R = n_rays; a.nv = nv; a.H = H; a.W = W; a.patch = patch_size;
  a.ref_out = ref_out; a.src_out = src_out;
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
  hipLaunchKernelGGL(patch_warp_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
