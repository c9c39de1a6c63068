// K11b: backward of the NeuS compositing (first backward kernel of the training row).
// Differentiates what composite.hip computes from (sdf, grad, colour, inv_s) to (colour_fine, render_depth, the eikonal
// sums): the autograd of render_core's tail, implicit_surface.py:126-166, restated in closed form.
//   w_k = a_k T_k,  T_k = prod_{j<k} (1 - a_j + 1e-7),  u_k = dL/dw_k = g_C . c_k + g_D cos z_k
//   dL/da_k = u_k T_k - (sum_{j>k} u_j w_j) / (1 - a_k + 1e-7)                      (reverse scan across the ray)
//   a = clip((p - n + 1e-5) / (p + 1e-5), 0, 1) m,  p, n = sigmoid((s -/+ h) inv_s),  h = clamp(iter_cos, +-10) dist / 2,
//   iter_cos = -(relu(-c/2 + 1/2)(1 - A) + relu(-c) A) m,  c = d . grad
// plus the eikonal term  eik_scale * relax * d(|grad| - 1)^2 / d grad  (eik_scale = dL/d gradient_error / (sum relax + 1e-5)).
// One wavefront per ray with composite.hip's lane ownership; the sdf_depth branch (a detached, discontinuous selection
// in the reference's loss) carries no gradient.  Outputs: d_sdf (R,S), d_grad (R,S,3), d_color (R,S,3), d_inv_s (R) per-ray
// partial sums (the caller adds them and chains through inv_s = exp(10 variance)).
#include "common.h"

namespace {

struct CompBwdArgs {
  const float* sdf;
  const float* grad;
  const float* color;
  const float* mid_z;
  const float* dists;
  const float* pts;
  const uint8_t* vmask;
  const float* rays_d;
  int n_rays, S;
  float inv_s, anneal, eik_scale;
  const float* eik_up;   // device scalar multiplying eik_scale (dL/d gradient_error as autograd holds it) or null
  float rot[9];
  const float* g_color;  // (R,3)
  const float* g_depth;  // (R) or null
  float* d_sdf;
  float* d_grad;
  float* d_color;
  float* d_inv_s;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int MAXP = SURF_MAX_SAMPLES / 64;

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompBwdArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;
  const int S = a.S, P = (S + 63) / 64;
  const int64_t base = (int64_t)ray * S;
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  const float cz = a.rot[6] * dx + a.rot[7] * dy + a.rot[8] * dz;
  const float gC[3] = {a.g_color[ray * 3 + 0], a.g_color[ray * 3 + 1], a.g_color[ray * 3 + 2]};
  const float gD = a.g_depth ? a.g_depth[ray] * cz : 0.f;

  float alpha[MAXP], u[MAXP], da_ds[MAXP], da_dtc[MAXP], da_dis[MAXP];
  bool vm[MAXP];
  float prod = 1.0f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int k = lane * P + p;
    alpha[p] = 0.f; u[p] = 0.f; da_ds[p] = da_dtc[p] = da_dis[p] = 0.f; vm[p] = false;
    if (p < P && k < S) {
      const int64_t o = base + k;
      vm[p] = a.vmask[o] != 0;
      const float vmf = vm[p] ? 1.f : 0.f;
      const float dist = a.dists[o];
      const float s = vm[p] ? a.sdf[o] : 100.f;
      const float gx = vm[p] ? a.grad[o * 3 + 0] : 0.f, gy = vm[p] ? a.grad[o * 3 + 1] : 0.f, gz = vm[p] ? a.grad[o * 3 + 2] : 0.f;
      const float tc = dx * gx + dy * gy + dz * gz;
      const float r1 = -tc * 0.5f + 0.5f, r2 = -tc;
      const float ic = -(fmaxf(r1, 0.f) * (1.0f - a.anneal) + fmaxf(r2, 0.f) * a.anneal) * vmf;
      const float dic_dtc = ((r1 > 0.f ? 0.5f * (1.0f - a.anneal) : 0.f) + (r2 > 0.f ? a.anneal : 0.f)) * vmf;
      const bool inr = ic > -10.f && ic < 10.f;
      const float h = fminf(fmaxf(ic, -10.f), 10.f) * dist * 0.5f;
      const float dh_dtc = inr ? dic_dtc * dist * 0.5f : 0.f;
      const float pc = sigm((s - h) * a.inv_s), nc = sigm((s + h) * a.inv_s);
      const float den = pc + 1e-5f;
      const float al = (pc - nc + 1e-5f) / den;
      alpha[p] = fminf(fmaxf(al, 0.f), 1.f) * vmf;
      const float open = (al > 0.f && al < 1.f) ? vmf : 0.f;            // d clip / d al
      const float da_dp = nc / (den * den), da_dn = -1.0f / den;
      const float dp = pc * (1.0f - pc), dn = nc * (1.0f - nc);
      da_ds[p] = open * (da_dp * dp + da_dn * dn) * a.inv_s;
      da_dtc[p] = open * (-da_dp * dp + da_dn * dn) * a.inv_s * dh_dtc;
      da_dis[p] = open * (da_dp * dp * (s - h) + da_dn * dn * (s + h));
      float uc = gD * a.mid_z[o];
      if (vm[p]) uc += gC[0] * a.color[o * 3 + 0] + gC[1] * a.color[o * 3 + 1] + gC[2] * a.color[o * 3 + 2];
      u[p] = uc;
      prod *= (1.0f - alpha[p] + 1e-7f);
    }
  }
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o);
    if (lane >= o) incl *= t;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.0f;
  float w[MAXP], Tk[MAXP], lane_sum = 0.f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    w[p] = 0.f; Tk[p] = 0.f;
    const int k = lane * P + p;
    if (p < P && k < S) {
      Tk[p] = T;
      w[p] = alpha[p] * T;
      T *= (1.0f - alpha[p] + 1e-7f);
      lane_sum += u[p] * w[p];
    }
  }
  // exclusive suffix sum over lanes of lane_sum
  float sfx = lane_sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(sfx, o);
    if (lane + o < 64) sfx += t;
  }
  float A = __shfl_down(sfx, 1);
  if (lane == 63) A = 0.f;
  float dis = 0.f;
#pragma unroll
  for (int p = MAXP - 1; p >= 0; --p) {
    const int k = lane * P + p;
    if (p < P && k < S) {
      const int64_t o = base + k;
      const float dal = u[p] * Tk[p] - A / (1.0f - alpha[p] + 1e-7f);
      A += u[p] * w[p];
      a.d_sdf[o] = dal * da_ds[p];
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (vm[p]) {
        const float t = dal * da_dtc[p];
        gx = t * dx; gy = t * dy; gz = t * dz;
        // eikonal: relax (|g| - 1)^2, relax = |pts| < 1.2 (masked-in samples)
        const float px = a.pts[o * 3 + 0], py = a.pts[o * 3 + 1], pz = a.pts[o * 3 + 2];
        const float eik_scale = a.eik_up ? a.eik_scale * *a.eik_up : a.eik_scale;
        if (eik_scale != 0.f && sqrtf(px * px + py * py + pz * pz) < 1.2f) {
          const float ex = a.grad[o * 3 + 0], ey = a.grad[o * 3 + 1], ez = a.grad[o * 3 + 2];
          const float n = sqrtf(ex * ex + ey * ey + ez * ez);
          if (n > 0.f) {
            const float f = eik_scale * 2.0f * (n - 1.0f) / n;
            gx += f * ex; gy += f * ey; gz += f * ez;
          }
        }
      }
      a.d_grad[o * 3 + 0] = gx; a.d_grad[o * 3 + 1] = gy; a.d_grad[o * 3 + 2] = gz;
      const float wc = vm[p] ? w[p] : 0.f;
      a.d_color[o * 3 + 0] = wc * gC[0]; a.d_color[o * 3 + 1] = wc * gC[1]; a.d_color[o * 3 + 2] = wc * gC[2];
      dis += dal * da_dis[p];
    }
  }
  dis = wave_sum(dis);
  if (lane == 0) a.d_inv_s[ray] = dis;
}

}  // namespace

extern "C" int surf_composite_backward_s(const float* sdf, const float* grad, const float* color, const float* mid_z,
                                         const float* dists, const float* pts, const uint8_t* vmask, const float* rays_d, int n_rays,
                                         int S, float inv_s, float cos_anneal_ratio, const float* h_rot_ref, const float* g_color,
                                         const float* g_depth, float eik_scale, const float* eik_upstream, float* d_sdf,
                                         float* d_grad, float* d_color, float* d_inv_s, void* stream) {
  if (!sdf || !grad || !color || !mid_z || !dists || !pts || !vmask || !rays_d || !h_rot_ref || !g_color) return SURF_E_ARG;
  if (!d_sdf || !d_grad || !d_color || !d_inv_s || n_rays <= 0 || S < 2) return SURF_E_ARG;
  if (S > SURF_MAX_SAMPLES) return SURF_E_LIMIT;
  CompBwdArgs a;
  a.sdf = sdf; a.grad = grad; a.color = color; a.mid_z = mid_z; a.dists = dists; a.pts = pts; a.vmask = vmask; a.rays_d = rays_d;
  a.n_rays = n_rays; a.S = S; a.inv_s = inv_s; a.anneal = cos_anneal_ratio; a.eik_scale = eik_scale; a.eik_up = eik_upstream;
  for (int i = 0; i < 9; ++i) a.rot[i] = h_rot_ref[i];
  a.g_color = g_color; a.g_depth = g_depth; a.d_sdf = d_sdf; a.d_grad = d_grad; a.d_color = d_color; a.d_inv_s = d_inv_s;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int surf_composite_backward(const float* sdf, const float* grad, const float* color, const float* mid_z,
                                       const float* dists, const float* pts, const uint8_t* vmask, const float* rays_d, int n_rays,
                                       int S, float inv_s, float cos_anneal_ratio, const float* h_rot_ref, const float* g_color,
                                       const float* g_depth, float eik_scale, float* d_sdf, float* d_grad, float* d_color,
                                       float* d_inv_s, void* stream) {
  return surf_composite_backward_s(sdf, grad, color, mid_z, dists, pts, vmask, rays_d, n_rays, S, inv_s, cos_anneal_ratio, h_rot_ref,
                                   g_color, g_depth, eik_scale, nullptr, d_sdf, d_grad, d_color, d_inv_s, stream);
}
