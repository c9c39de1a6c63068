// K15  local normalised cross-correlation of the surface patches (training loss term `mfc_loss`).
// Replaces compute_LNCC2  models/losses/ncc.py:7-51  (called at losses/loss.py:43 on the outputs of surface_patch_warp2):
// the reference's five grouped conv2d's with an all-ones patch-wide filter, read at the patch centre, are plain sums over
// the P patch elements.  Per (ray, source view, channel): cc = cov^2 / (var_ref var_src + 1e-5); ncc = clamp(1 - cc, 0, 2),
// averaged over the C channels; output = mean of the two smallest views (torch.topk(2, largest=False)).
//
// One wavefront per ray; lane = (element slice, channel): 64 = 5 slices x 12 channels + 4 idle for C = 12; the patches are
// read once, channel-contiguous (48-byte rows).  Byte-bound: (1 + nsrc) P C 4 bytes per ray.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void lncc_kernel(const float* __restrict__ ref, const float* __restrict__ src, int64_t R,
                                                   int nsrc, int P, int C, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int nsl = 64 / C;                       // element slices per channel
  const int ch = lane % C, sl = lane / C;
  const bool act = sl < nsl;
  const float* __restrict__ rp = ref + ray * (int64_t)P * C;
  // sums of the reference patch for this lane's (slice, channel)
  float r1 = 0.f, r2 = 0.f;
  if (act)
    for (int e = sl; e < P; e += nsl) {
      const float v = rp[e * C + ch];
      r1 += v;
      r2 = fmaf(v, v, r2);
    }
  float best0 = 3.0e38f, best1 = 3.0e38f;       // the two smallest per-view means
  for (int v = 0; v < nsrc; ++v) {
    const float* __restrict__ sp = src + ((int64_t)v * R + ray) * (int64_t)P * C;
    float s1 = 0.f, s2 = 0.f, rs = 0.f;
    if (act)
      for (int e = sl; e < P; e += nsl) {
        const float a = rp[e * C + ch], b = sp[e * C + ch];
        s1 += b;
        s2 = fmaf(b, b, s2);
        rs = fmaf(a, b, rs);
      }
    // reduce over the slices of a channel (lanes ch, ch + C, ...): wave-wide sums per channel through LDS-free shuffles
    float t_r1 = 0.f, t_r2 = 0.f, t_s1 = 0.f, t_s2 = 0.f, t_rs = 0.f;
    for (int k = 0; k < nsl; ++k) {
      const int from = ch + k * C;
      t_r1 += __shfl(r1, from); t_r2 += __shfl(r2, from);
      t_s1 += __shfl(s1, from); t_s2 += __shfl(s2, from); t_rs += __shfl(rs, from);
    }
    const float n = (float)P;
    const float ur = t_r1 / n, us = t_s1 / n;
    const float cross = t_rs - us * t_r1 - ur * t_s1 + ur * us * n;
    const float rvar = t_r2 - 2.0f * ur * t_r1 + ur * ur * n;
    const float svar = t_s2 - 2.0f * us * t_s1 + us * us * n;
    const float cc = cross * cross / (rvar * svar + 1e-5f);
    float ncc = fminf(fmaxf(1.0f - cc, 0.0f), 2.0f);
    // mean over channels: lanes 0..C-1 hold one channel each
    float m = lane < C ? ncc : 0.f;
    m = wave_sum(m) / (float)C;
    if (m < best0) { best1 = best0; best0 = m; }
    else if (m < best1) best1 = m;
  }
  if (lane == 0) out[ray] = 0.5f * (best0 + best1);
}


// Backward of the kernel above (the autograd of compute_LNCC2 under loss.backward(), runner.py:163): g_ref (1,R,P,C) and
// g_src (nsrc,R,P,C) for an upstream gradient g_out (R) of the per-ray value.  The value is the mean of the two smallest
// per-view means (torch.topk(2, largest=False): the earlier view wins a tie, as above), so only those two views receive
// gradient; inside a view every channel contributes -(1/C) d cc with cc = cross^2 / (rvar svar + 1e-5) and
//   d cross / d r_e = s_e - us,  d rvar / d r_e = 2 (r_e - ur),  d cross / d s_e = r_e - ur,  d svar / d s_e = 2 (s_e - us);
// the clamp passes gradient on [0, 2] inclusive (torch.clamp).  Same lane mapping as the forward; three passes over the
// patches: view statistics, then the two selected views again for the element-wise gradients, zeros for the others.
__global__ __launch_bounds__(256) void lncc_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                                                       const float* __restrict__ g_out, int64_t R, int nsrc, int P, int C,
                                                       float* __restrict__ g_ref, float* __restrict__ g_src) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int nsl = 64 / C;
  const int ch = lane % C, sl = lane / C;
  const bool act = sl < nsl;
  const float* __restrict__ rp = ref + ray * (int64_t)P * C;
  const float n = (float)P;
  float r1 = 0.f, r2 = 0.f;
  if (act)
    for (int e = sl; e < P; e += nsl) {
      const float v = rp[e * C + ch];
      r1 += v;
      r2 = fmaf(v, v, r2);
    }
  float t_r1 = 0.f, t_r2 = 0.f;
  for (int k = 0; k < nsl; ++k) {
    t_r1 += __shfl(r1, ch + k * C);
    t_r2 += __shfl(r2, ch + k * C);
  }
  const float ur = t_r1 / n;
  const float rvar = t_r2 - 2.0f * ur * t_r1 + ur * ur * n;
  float best0 = 3.0e38f, best1 = 3.0e38f;
  int sel0 = -1, sel1 = -1;
  // coefficients of the selected views for this lane's channel: d cc = ca d cross + cr d rvar + cs d svar
  float ca[2] = {0.f, 0.f}, cr[2] = {0.f, 0.f}, cs[2] = {0.f, 0.f}, usv[2] = {0.f, 0.f};
  for (int v = 0; v < nsrc; ++v) {
    const float* __restrict__ sp = src + ((int64_t)v * R + ray) * (int64_t)P * C;
    float s1 = 0.f, s2 = 0.f, rs = 0.f;
    if (act)
      for (int e = sl; e < P; e += nsl) {
        const float a = rp[e * C + ch], b = sp[e * C + ch];
        s1 += b;
        s2 = fmaf(b, b, s2);
        rs = fmaf(a, b, rs);
      }
    float t_s1 = 0.f, t_s2 = 0.f, t_rs = 0.f;
    for (int k = 0; k < nsl; ++k) {
      const int from = ch + k * C;
      t_s1 += __shfl(s1, from); t_s2 += __shfl(s2, from); t_rs += __shfl(rs, from);
    }
    const float us = t_s1 / n;
    const float cross = t_rs - us * t_r1 - ur * t_s1 + ur * us * n;
    const float svar = t_s2 - 2.0f * us * t_s1 + us * us * n;
    const float den = rvar * svar + 1e-5f;
    const float cc = cross * cross / den;
    const float raw = 1.0f - cc;
    const float ncc = fminf(fmaxf(raw, 0.0f), 2.0f);
    const float pass = (raw >= 0.0f && raw <= 2.0f) ? 1.0f : 0.0f;
    float m = lane < C ? ncc : 0.f;
    m = wave_sum(m) / (float)C;
    const float a_ = pass * 2.0f * cross / den, q_ = -pass * cross * cross / (den * den);
    if (m < best0) {
      best1 = best0; sel1 = sel0; ca[1] = ca[0]; cr[1] = cr[0]; cs[1] = cs[0]; usv[1] = usv[0];
      best0 = m; sel0 = v; ca[0] = a_; cr[0] = q_ * svar; cs[0] = q_ * rvar; usv[0] = us;
    } else if (m < best1) {
      best1 = m; sel1 = v; ca[1] = a_; cr[1] = q_ * svar; cs[1] = q_ * rvar; usv[1] = us;
    }
  }
  // d out / d ncc_view = 0.5, d ncc_view / d ncc_channel = 1/C, d ncc_channel / d cc = -1
  const float up = -g_out[ray] * 0.5f / (float)C;
  float* __restrict__ gr = g_ref + ray * (int64_t)P * C;
  const float* __restrict__ sp0 = src + ((int64_t)(sel0 < 0 ? 0 : sel0) * R + ray) * (int64_t)P * C;
  const float* __restrict__ sp1 = src + ((int64_t)(sel1 < 0 ? 0 : sel1) * R + ray) * (int64_t)P * C;
  float* __restrict__ gs0 = g_src + ((int64_t)(sel0 < 0 ? 0 : sel0) * R + ray) * (int64_t)P * C;
  float* __restrict__ gs1 = g_src + ((int64_t)(sel1 < 0 ? 0 : sel1) * R + ray) * (int64_t)P * C;
  for (int v = 0; v < nsrc; ++v) {                         // views outside the top two: zero gradient
    if (v == sel0 || v == sel1) continue;
    float* __restrict__ gz = g_src + ((int64_t)v * R + ray) * (int64_t)P * C;
    for (int e = lane; e < P * C; e += 64) gz[e] = 0.f;
  }
  if (act)
    for (int e = sl; e < P; e += nsl) {
      const int o = e * C + ch;
      const float r = rp[o], dr = r - ur;
      float g = 0.f;
      if (sel0 >= 0) {
        const float ds = sp0[o] - usv[0];
        g += ca[0] * ds + 2.0f * cr[0] * dr;
        gs0[o] = up * (ca[0] * dr + 2.0f * cs[0] * ds);
      }
      if (sel1 >= 0) {
        const float ds = sp1[o] - usv[1];
        g += ca[1] * ds + 2.0f * cr[1] * dr;
        gs1[o] = up * (ca[1] * dr + 2.0f * cs[1] * ds);
      }
      gr[o] = up * g;
    }
}

}  // namespace

extern "C" int surf_lncc(const float* ref, const float* src, int64_t n_rays, int n_src, int patch_elems, int channels,
                         float* out, void* stream) {
  if (!ref || !src || !out || n_rays <= 0) return SURF_E_ARG;
  if (n_src < 2 || patch_elems < 1 || channels < 1 || channels > 64) return SURF_E_ARG;
  hipLaunchKernelGGL(lncc_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, src, n_rays,
                     n_src, patch_elems, channels, out);
  return surf_check_launch();
}

extern "C" int surf_lncc_backward(const float* ref, const float* src, const float* g_out, int64_t n_rays, int n_src,
                                  int patch_elems, int channels, float* g_ref, float* g_src, void* stream) {
  if (!ref || !src || !g_out || !g_ref || !g_src || n_rays <= 0) return SURF_E_ARG;
  if (n_src < 2 || patch_elems < 1 || channels < 1 || channels > 64) return SURF_E_ARG;
  hipLaunchKernelGGL(lncc_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, src, g_out,
                     n_rays, n_src, patch_elems, channels, g_ref, g_src);
  return surf_check_launch();
}
