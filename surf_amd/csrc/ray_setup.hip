// K8: per-ray z-sampling guided by the matching volume, section mid-points and voxel mask.
// One wavefront = one ray; lanes = coarse taps / samples.
//
// Restates: ImplicitSurface.render  implicit_surface.py:268-311 (jitters of render.perturb > 0 supplied by the host)
//                         render_core head         implicit_surface.py:72-86
//                         lookup_volume            projector.py:392-420
#include "common.h"

namespace {

struct RaySetupArgs {
  const float* rays_o;
  const float* rays_d;
  const float* near;
  const float* far;
  int n_rays;
  const float* mvol;
  int Dm;
  const float* lin_depth;
  int n_depth;
  const float* lin_samples;
  const float* jitter;  // (n_rays, n_stage) or null
  int n_samples[SURF_MAX_STAGES];
  float ranges[SURF_MAX_STAGES];
  int n_stage;
  int S;
  float sample_dist;
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  int n_vol;
  float* z_vals;
  float* mid_z;
  float* dists;
  float* pts;
  uint8_t* vmask;
};

constexpr int RAYS_PER_BLOCK = 4;  // 256 threads

__global__ __launch_bounds__(256) void ray_setup_kernel(RaySetupArgs a) {
  __shared__ float s_z[RAYS_PER_BLOCK][SURF_MAX_SAMPLES];
  __shared__ float s_sorted[RAYS_PER_BLOCK][SURF_MAX_SAMPLES + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int ray = blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool live = ray < a.n_rays;
  if (!live) ray = a.n_rays - 1;  // keep every wave on the barrier path; it just does not store

  const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
  const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
  const float near = a.near[ray], far = a.far[ray];
  const float range = far - near;

  // ---- coarse surface estimate: softmax-expected z over n_depth taps (implicit_surface.py:281-291)
  // two lanes per tap: lane parity = dz, so the two z-corners of a trilinear row are adjacent lanes and adjacent floats (one
  // 8-byte request instead of two scattered 4-byte ones); the halves meet by one xor-shuffle.  Both lanes of a pair then hold the
  // same rho: den and num are both counted twice and their ratio is unchanged.
  float rho[8], zc[8];
  float m = -INFINITY;
  const int half = lane >> 1, cdz = lane & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = half + 32 * i;
    rho[i] = -INFINITY;
    zc[i] = 0.f;
    float part = 0.f;
    const bool on = k < a.n_depth;
    if (on) {
      const float z = near + range * a.lin_depth[k];
      const float gx = unnorm_acf(ox + dx * z, a.Dm), gy = unnorm_acf(oy + dy * z, a.Dm), gz = unnorm_acf(oz + dz * z, a.Dm);
      const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
      const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
      const int x0 = (int)fx, y0 = (int)fy, zi = (int)fz + cdz;
      const float wz = cdz ? tz : 1.0f - tz;
      if ((zi >= 0) & (zi < a.Dm)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xi = x0 + (c >> 1), yi = y0 + (c & 1);
          const float wgt = ((c >> 1) ? tx : 1.0f - tx) * ((c & 1) ? ty : 1.0f - ty) * wz;
          if ((xi >= 0) & (xi < a.Dm) & (yi >= 0) & (yi < a.Dm)) part += a.mvol[((int64_t)xi * a.Dm + yi) * a.Dm + zi] * wgt;
        }
      }
      zc[i] = z;
    }
    part += __shfl_xor(part, 1);
    if (on) {
      rho[i] = part;
      m = fmaxf(m, part);
    }
  }
  m = wave_max(m);
  float den = 0.f, num = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float e = (half + 32 * i < a.n_depth) ? expf(rho[i] - m) : 0.f;
    den += e;
    num += e * zc[i];
  }
  den = wave_sum(den);
  num = wave_sum(num);
  const float surf_z = num / den;

  // ---- per-stage linspaces (stage 0: [near,far]; stage s>0: a band around surf_z) :293-308
  for (int k = lane; k < a.S; k += 64) {
    int s = 0, off = 0;
    while (s + 1 < a.n_stage && k >= off + a.n_samples[s]) {
      off += a.n_samples[s];
      ++s;
    }
    float lo = near, hi = far;
    if (s > 0) {
      float r = range * a.ranges[s];
      lo = surf_z - r;
      hi = surf_z + r;
      if (hi > far) lo = lo - (hi - far);
      if (lo < near) hi = hi + (near - lo);
      lo = fminf(fmaxf(lo, near), far);
      hi = fminf(fmaxf(hi, near), far);
    }
    float z = lo + (hi - lo) * a.lin_samples[k];
    if (a.jitter) {  // t * 2.0 / n (stage 0), t * (far_stage - near_stage) / n (bands): the reference's evaluation order
      const float t = a.jitter[(int64_t)ray * a.n_stage + s];
      z = z + (s == 0 ? t * 2.0f / (float)a.n_samples[0] : t * (hi - lo) / (float)a.n_samples[s]);
    }
    s_z[wave][k] = z;
  }
  __syncthreads();

  // ---- sort by rank counting (values only matter; ties broken by index) :310-311
  for (int k = lane; k < a.S; k += 64) {
    float z = s_z[wave][k];
    int rank = 0;
    for (int j = 0; j < a.S; ++j) {
      float zj = s_z[wave][j];
      rank += (zj < z) | ((zj == z) & (j < k));
    }
    s_sorted[wave][rank] = z;
  }
  __syncthreads();

  // ---- section lengths, mid-points, sample positions, occupancy mask (implicit_surface.py:75-86)
  for (int k = lane; k < a.S; k += 64) {
    float z = s_sorted[wave][k];
    float dist = (k + 1 < a.S) ? s_sorted[wave][k + 1] - z : a.sample_dist;
    float mid = z + dist * 0.5f;
    float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
    bool occ = false;
    for (int v = 0; v < a.n_vol; ++v) occ |= occupied_nearest(a.tables[v], a.dims[v], px, py, pz);
    if (live) {
      int64_t o = (int64_t)ray * a.S + k;
      if (a.z_vals) a.z_vals[o] = z;
      a.mid_z[o] = mid;
      a.dists[o] = dist;
      a.pts[o * 3 + 0] = px;
      a.pts[o * 3 + 1] = py;
      a.pts[o * 3 + 2] = pz;
      a.vmask[o] = occ ? 1 : 0;
    }
  }
}

__global__ void pack_texel4_kernel(const float* __restrict__ src, int n, int C, int H, int W, float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)n * H * W;
  if (i >= total) return;
  int64_t hw = (int64_t)H * W;
  int64_t img = i / hw, p = i % hw;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) v[c] = src[(img * C + c) * hw + p];
  *reinterpret_cast<f32x4*>(dst + i * 4) = v;
}

}  // namespace

extern "C" int surf_abi_version(void) { return SURF_ABI_VERSION; }

extern "C" int surf_pack_texel4(const float* src, int n, int C, int H, int W, float* dst, void* stream) {
  if (!src || !dst || n <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return SURF_E_ARG;
  int64_t total = (int64_t)n * H * W;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_texel4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, n, C, H, W, dst);
  return surf_check_launch();
}

extern "C" int surf_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far, int n_rays,
                              const float* mvol, int Dm, const float* lin_depth, int n_depth,
                              const float* lin_samples, const int* h_n_samples, const float* h_sample_ranges,
                              int n_stage, const float* jitter, float sample_dist, const int32_t* const* h_tables, const int* h_dims,
                              int n_vol, float* z_vals, float* mid_z, float* dists, float* pts, uint8_t* vmask,
                              void* stream) {
  if (!rays_o || !rays_d || !near || !far || !mvol || !lin_depth || !lin_samples || !h_n_samples ||
      !h_sample_ranges || !h_tables || !h_dims || !mid_z || !dists || !pts || !vmask)
    return SURF_E_ARG;
  if (n_rays <= 0 || Dm <= 0 || n_depth <= 0) return SURF_E_ARG;
  if (n_stage <= 0 || n_stage > SURF_MAX_STAGES || n_vol <= 0 || n_vol > SURF_MAX_STAGES || n_depth > 256)
    return SURF_E_LIMIT;
  RaySetupArgs a;
  a.rays_o = rays_o; a.rays_d = rays_d; a.near = near; a.far = far; a.n_rays = n_rays;
  a.mvol = mvol; a.Dm = Dm; a.lin_depth = lin_depth; a.n_depth = n_depth; a.lin_samples = lin_samples;
  a.n_stage = n_stage; a.S = 0; a.sample_dist = sample_dist; a.n_vol = n_vol; a.jitter = jitter;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.n_samples[s] = s < n_stage ? h_n_samples[s] : 0;
    a.ranges[s] = s < n_stage ? h_sample_ranges[s] : 0.f;
    a.S += a.n_samples[s];
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    if (s < n_vol && (!h_tables[s] || h_dims[s] <= 0)) return SURF_E_ARG;
  }
  if (a.S <= 0) return SURF_E_ARG;
  if (a.S > SURF_MAX_SAMPLES) return SURF_E_LIMIT;
  a.z_vals = z_vals; a.mid_z = mid_z; a.dists = dists; a.pts = pts; a.vmask = vmask;
  int grid = (n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
  hipLaunchKernelGGL(ray_setup_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
