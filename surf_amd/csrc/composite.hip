// K11: NeuS SDF->alpha compositing, first zero-crossing depth and per-ray reductions.
// One wavefront = one ray; every lane owns a contiguous run of P = ceil(S/64) samples, the
// transmittance is an exclusive product scan across lanes.
//
// Restates render_core's tail  implicit_surface.py:123-166,181-216  and validate's normal sum :380-382.
#include "common.h"

namespace {

struct CompositeArgs {
  const float* sdf;
  const float* grad;
  const float* color;
  const uint8_t* n_valid;
  const float* mid_z;
  const float* dists;
  const float* pts;
  const uint8_t* vmask;
  const float* rays_d;
  int n_rays;
  int S;
  float inv_s;
  float anneal;
  float rot[9];
  float* out_color;
  float* out_depth;
  float* out_sdf_depth;
  float* out_normal;
  float* out_normal_val;
  uint8_t* out_valid;
  uint8_t* out_mid_inside;
  float* out_weights;
  float* out_inside;
  float* out_eik;
  float* out_z_sdf0;  // (R) zero crossing ray parameter, before validity / cosine
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int MAXP = SURF_MAX_SAMPLES / 64;

__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;  // wave-uniform, no block barriers below
  const int S = a.S;
  const int P = (S + 63) / 64;
  const int64_t base = (int64_t)ray * S;
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];

  float alpha[MAXP], sdfv[MAXP], gx[MAXP], gy[MAXP], gz[MAXP], midv[MAXP], insv[MAXP];
  bool vm[MAXP];
  float prod = 1.0f;
  float eik_num = 0.f, eik_den = 0.f;
  int cnt_valid = 0;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    int k = lane * P + p;
    alpha[p] = 0.f; sdfv[p] = 100.f; gx[p] = gy[p] = gz[p] = 0.f; midv[p] = 0.f; insv[p] = 0.f; vm[p] = false;
    if (p < P && k < S) {
      int64_t o = base + k;
      vm[p] = a.vmask[o] != 0;
      float dist = a.dists[o];
      midv[p] = a.mid_z[o];
      if (vm[p]) {
        sdfv[p] = a.sdf[o];
        gx[p] = a.grad[o * 3 + 0]; gy[p] = a.grad[o * 3 + 1]; gz[p] = a.grad[o * 3 + 2];
        cnt_valid += (a.n_valid[o] > 1) ? 1 : 0;
      }
      float vmf = vm[p] ? 1.f : 0.f;
      float true_cos = dx * gx[p] + dy * gy[p] + dz * gz[p];
      float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.f) * (1.0f - a.anneal) + fmaxf(-true_cos, 0.f) * a.anneal);
      iter_cos = iter_cos * vmf;
      float half = fminf(fmaxf(iter_cos, -10.f), 10.f) * dist * 0.5f;
      float prev_cdf = sigmoidf_((sdfv[p] - half) * a.inv_s);
      float next_cdf = sigmoidf_((sdfv[p] + half) * a.inv_s);
      float al = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
      alpha[p] = fminf(fmaxf(al, 0.f), 1.f) * vmf;
      float px = a.pts[o * 3 + 0], py = a.pts[o * 3 + 1], pz = a.pts[o * 3 + 2];
      float pn = sqrtf(px * px + py * py + pz * pz);
      insv[p] = (pn < 1.0f) ? vmf : 0.f;
      float relax = (pn < 1.2f) ? vmf : 0.f;
      float gn = sqrtf(gx[p] * gx[p] + gy[p] * gy[p] + gz[p] * gz[p]) - 1.0f;
      eik_num += relax * gn * gn;
      eik_den += relax;
      prod *= (1.0f - alpha[p] + 1e-7f);
    }
  }
  // exclusive product scan over lanes
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(incl, o);
    if (lane >= o) incl *= t;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.0f;

  float cr = 0.f, cg = 0.f, cb = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, dep = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  int first = S;  // first zero crossing owned by this lane
  // neighbour (k+1) values of the last element of the run come from the next lane's first element
  float nsdf = __shfl_down(sdfv[0], 1);
  int nvm = __shfl_down((int)vm[0], 1);
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    int k = lane * P + p;
    if (p < P && k < S) {
      int64_t o = base + k;
      float w = alpha[p] * T;
      T *= (1.0f - alpha[p] + 1e-7f);
      if (a.out_weights) a.out_weights[o] = w;
      if (a.out_inside) a.out_inside[o] = insv[p];
      if (vm[p]) {
        cr += a.color[o * 3 + 0] * w; cg += a.color[o * 3 + 1] * w; cb += a.color[o * 3 + 2] * w;
      }
      nx += gx[p] * w; ny += gy[p] * w; nz += gz[p] * w;
      vx += gx[p] * w * insv[p]; vy += gy[p] * w * insv[p]; vz += gz[p] * w * insv[p];
      dep += midv[p] * w;
      // zero crossing between k and k+1
      float s_next;
      bool m_next;
      if (p + 1 < P) { s_next = sdfv[p + 1 < MAXP ? p + 1 : p]; m_next = vm[p + 1 < MAXP ? p + 1 : p]; }
      else { s_next = nsdf; m_next = nvm != 0; }
      if (k + 1 < S && first == S && (sdfv[p] * s_next <= 0.f) && vm[p] && m_next) first = k;
    }
  }
  cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb);
  nx = wave_sum(nx); ny = wave_sum(ny); nz = wave_sum(nz);
  vx = wave_sum(vx); vy = wave_sum(vy); vz = wave_sum(vz);
  dep = wave_sum(dep);
  eik_num = wave_sum(eik_num); eik_den = wave_sum(eik_den);
  cnt_valid = (int)wave_sum((float)cnt_valid);
  first = wave_min_i(first);

  if (lane == 0) {
    const float* R = a.rot;
    float cz = R[6] * dx + R[7] * dy + R[8] * dz;
    if (a.out_color) { a.out_color[ray * 3 + 0] = cr; a.out_color[ray * 3 + 1] = cg; a.out_color[ray * 3 + 2] = cb; }
    if (a.out_normal) {
      a.out_normal[ray * 3 + 0] = R[0] * nx + R[1] * ny + R[2] * nz;
      a.out_normal[ray * 3 + 1] = R[3] * nx + R[4] * ny + R[5] * nz;
      a.out_normal[ray * 3 + 2] = R[6] * nx + R[7] * ny + R[8] * nz;
    }
    if (a.out_normal_val) { a.out_normal_val[ray * 3 + 0] = vx; a.out_normal_val[ray * 3 + 1] = vy; a.out_normal_val[ray * 3 + 2] = vz; }
    if (a.out_depth) a.out_depth[ray] = dep * cz;
    if (a.out_valid) a.out_valid[ray] = cnt_valid > 8 ? 1 : 0;
    if (a.out_eik) { a.out_eik[ray * 2 + 0] = eik_num; a.out_eik[ray * 2 + 1] = eik_den; }

    // first zero crossing (index 0 when there is none, exactly like argmax of an all-zero row)
    bool any = first < S;
    int k0 = any ? first : 0, k1 = k0 + 1;
    float s[2], z[2], ins[2], g[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t o = base + (i ? k1 : k0);
      bool m = a.vmask[o] != 0;
      s[i] = m ? a.sdf[o] : 100.f;
      z[i] = a.mid_z[o];
      float px = a.pts[o * 3 + 0], py = a.pts[o * 3 + 1], pz = a.pts[o * 3 + 2];
      ins[i] = (m && sqrtf(px * px + py * py + pz * pz) < 1.0f) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) g[i][c] = m ? a.grad[o * 3 + c] : 0.f;
    }
    float mis = ((0.5f * (ins[0] + ins[1])) > 0.5f) ? 1.f : 0.f;
    if (!any) mis = 0.f;
    float dot = g[0][0] * g[1][0] + g[0][1] * g[1][1] + g[0][2] * g[1][2];
    float n0 = sqrtf(g[0][0] * g[0][0] + g[0][1] * g[0][1] + g[0][2] * g[0][2]);
    float n1 = sqrtf(g[1][0] * g[1][0] + g[1][1] * g[1][1] + g[1][2] * g[1][2]);
    float cosd = dot / (n0 * n1 + 1e-8f);
    if (!(cosd > 0.5f)) mis = 0.f;
    float z0 = (s[0] * z[1] - s[1] * z[0]) / (s[0] - s[1] + 1e-10f);
    if (a.out_sdf_depth) a.out_sdf_depth[ray] = z0 * cz * mis;
    if (a.out_z_sdf0) a.out_z_sdf0[ray] = z0;
    if (a.out_mid_inside) a.out_mid_inside[ray] = mis > 0.f ? 1 : 0;
  }
}

}  // namespace

extern "C" int surf_composite(const float* sdf, const float* grad, const float* color, const uint8_t* n_valid,
                              const float* mid_z, const float* dists, const float* pts, const uint8_t* vmask,
                              const float* rays_d, int n_rays, int S, float inv_s, float cos_anneal_ratio,
                              const float* h_rot_ref, float* out_color, float* out_depth, float* out_sdf_depth,
                              float* out_normal, float* out_normal_val, uint8_t* out_valid,
                              uint8_t* out_mid_inside, float* out_weights, float* out_inside, float* out_eik,
                              float* out_z_sdf0, void* stream) {
  if (!sdf || !grad || !color || !n_valid || !mid_z || !dists || !pts || !vmask || !rays_d || !h_rot_ref)
    return SURF_E_ARG;
  if (n_rays <= 0 || S < 2) return SURF_E_ARG;
  if (S > SURF_MAX_SAMPLES) return SURF_E_LIMIT;
  CompositeArgs a;
  a.sdf = sdf; a.grad = grad; a.color = color; a.n_valid = n_valid; a.mid_z = mid_z; a.dists = dists;
  a.pts = pts; a.vmask = vmask; a.rays_d = rays_d; a.n_rays = n_rays; a.S = S; a.inv_s = inv_s;
  a.anneal = cos_anneal_ratio;
  for (int i = 0; i < 9; ++i) a.rot[i] = h_rot_ref[i];
  a.out_color = out_color; a.out_depth = out_depth; a.out_sdf_depth = out_sdf_depth; a.out_normal = out_normal;
  a.out_normal_val = out_normal_val; a.out_valid = out_valid; a.out_mid_inside = out_mid_inside;
  a.out_weights = out_weights; a.out_inside = out_inside; a.out_eik = out_eik; a.out_z_sdf0 = out_z_sdf0;
  hipLaunchKernelGGL(composite_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
