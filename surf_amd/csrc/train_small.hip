// K12c (round 5): the launch-bound tail of a training step (runner.py:152-165).  With every heavy kernel of the step in HIP,
// 11 of its 98 ms were torch's own elementwise / fill / reduce helpers: ~1,900 launches of ~5 us each on a stream that is never
// idle (profiles/r05_train_kernel_stats.csv, scripts/count_aten_ops.py books them to the Python lines that issue them).  The
// three worst offenders, each a chain of tiny torch ops on small tensors, as single launches:
//
//   occupied_any_kernel      lookup_volume(pts, mask_volumes, 'nearest').any(-1) (implicit_surface.py:175): 15 torch ops per level,
//                            4 levels, two calls a step = 120 launches -> 2
//   masked_l1_kernel (+bwd)  sum(|pred - target| mask) / (sum(mask) + 1e-8) of the per-stage depth terms (losses/loss.py:71-93: 12
//                            calls a step, 8 torch ops each forward and ~5 in autograd's backward) -> 1 + 1 launch per call
//   weight_norm_bwd_kernel   the closed-form backward of W = g v / |v|_row (sdf_network.py:88-89) for the 7 layers of the SDF
//                            network: 11 torch ops per layer -> 1 launch for all layers
//
// None of them is bandwidth- or compute-relevant (<= 5.5 MB per call); the point is the launch count.
#include "common.h"

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int WN_LAYERS = 8;

// ---- occupied_any -------------------------------------------------------------------------------------------------------------
struct OccArgs {
  const float* pts;       // (n, 3)
  int64_t n;
  const int32_t* table[MAX_LEVELS];
  int dim[MAX_LEVELS];
  int levels;
  uint8_t* out;           // (n) 0 / 1
};

__global__ __launch_bounds__(256) void occupied_any_kernel(OccArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float x = a.pts[3 * i], y = a.pts[3 * i + 1], z = a.pts[3 * i + 2];
  bool occ = false;
  for (int l = 0; l < a.levels; ++l) {
    const int D = a.dim[l];
    // grid_sample 'nearest', align_corners=False: round half to even of the unnormalised coordinate (rintf), inside test on it
    const float gx = rintf(unnorm_acf(x, D)), gy = rintf(unnorm_acf(y, D)), gz = rintf(unnorm_acf(z, D));
    const float Df = (float)D;
    if (gx >= 0.f && gx < Df && gy >= 0.f && gy < Df && gz >= 0.f && gz < Df)
      occ = occ || a.table[l][((int64_t)gx * D + (int64_t)gy) * D + (int64_t)gz] >= 0;
  }
  a.out[i] = occ ? 1 : 0;
}

// ---- masked L1 ----------------------------------------------------------------------------------------------------------------
enum { MASK_F32 = 0, MASK_U8 = 1, MASK_TARGET_POSITIVE = 2 };
constexpr int L1_BLOCKS = 256;

struct L1Args {
  const float* pred;
  const float* target;
  const void* mask;       // float / uint8 (bool) / unused
  int kind;
  int64_t n;
  double* part;           // (2, L1_BLOCKS) partial sums
  unsigned* counter;      // 0 between launches
  float* out;             // [0] = loss, [1] = 1 / (sum mask + 1e-8)
};

__device__ __forceinline__ float mask_at(const L1Args& a, int64_t i) {
  if (a.kind == MASK_F32) return reinterpret_cast<const float*>(a.mask)[i];
  if (a.kind == MASK_U8) return reinterpret_cast<const uint8_t*>(a.mask)[i] ? 1.f : 0.f;
  return a.target[i] > 0.f ? 1.f : 0.f;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sums of one workgroup's two values, in a fixed order (deterministic): result valid in thread 0
__device__ __forceinline__ void block_sum2(double& s0, double& s1, double (&red)[2][4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  s0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  __syncthreads();
}

__global__ __launch_bounds__(256) void masked_l1_kernel(L1Args a) {
  __shared__ double red[2][4];
  __shared__ bool last;
  float num = 0.f, den = 0.f;       // <= n / 65,536 terms per thread
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const float m = mask_at(a, i);
    num += fabsf(a.pred[i] - a.target[i]) * m;
    den += m;
  }
  double s0 = num, s1 = den;
  block_sum2(s0, s1, red);
  if (threadIdx.x == 0) {
    a.part[blockIdx.x] = s0;
    a.part[L1_BLOCKS + blockIdx.x] = s1;
    __threadfence();
    last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last workgroup to arrive adds the partial sums up in block order: the result does not depend on the arrival order
  s0 = threadIdx.x < gridDim.x ? a.part[threadIdx.x] : 0.0;
  s1 = threadIdx.x < gridDim.x ? a.part[L1_BLOCKS + threadIdx.x] : 0.0;
  block_sum2(s0, s1, red);
  if (threadIdx.x == 0) {
    const double inv = 1.0 / (s1 + 1e-8);
    a.out[0] = (float)(s0 * inv);
    a.out[1] = (float)inv;
    *a.counter = 0u;
  }
}

struct L1BwdArgs {
  L1Args f;               // pred, target, mask, kind, n
  const float* inv_den;   // out[1] of the forward
  const float* upstream;  // device scalar
  float* g_pred;          // (n)
};

__global__ __launch_bounds__(256) void masked_l1_bwd_kernel(L1BwdArgs b) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= b.f.n) return;
  const float d = b.f.pred[i] - b.f.target[i];
  const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);       // sgn(0) = 0 as torch's abs backward
  b.g_pred[i] = (b.upstream[0] * b.inv_den[0]) * (s * mask_at(b.f, i));
}

// ---- weight-norm backward -----------------------------------------------------------------------------------------------------
struct WnArgs {
  const float* v[WN_LAYERS];
  const float* g[WN_LAYERS];
  const float* dW[WN_LAYERS];
  float* dv[WN_LAYERS];
  float* dg[WN_LAYERS];
  int cols[WN_LAYERS];
  int row0[WN_LAYERS + 1];    // first global row of layer l
  int layers;
};

// one wavefront per row: dg = <dW, v> / |v|,  dv = (g / |v|) (dW - (dg / |v|) v)
__global__ __launch_bounds__(64) void weight_norm_bwd_kernel(WnArgs a) {
  const int row = blockIdx.x, lane = threadIdx.x;
  int l = 0;
  while (l + 1 < a.layers && row >= a.row0[l + 1]) ++l;
  const int r = row - a.row0[l], C = a.cols[l];
  const float* v = a.v[l] + (int64_t)r * C;
  const float* dW = a.dW[l] + (int64_t)r * C;
  float s2 = 0.f, dot = 0.f;
  for (int c = lane; c < C; c += 64) { s2 = fmaf(v[c], v[c], s2); dot = fmaf(dW[c], v[c], dot); }
  s2 = wave_sum(s2);
  dot = wave_sum(dot);
  const float nrm = sqrtf(s2), dg = dot / nrm, k = a.g[l][r] / nrm, q = dg / nrm;
  if (lane == 0) a.dg[l][r] = dg;
  float* dv = a.dv[l] + (int64_t)r * C;
  for (int c = lane; c < C; c += 64) dv[c] = k * (dW[c] - q * v[c]);
}

}  // namespace

extern "C" int surf_occupied_any(const float* pts, int64_t n, const int32_t* const* h_tables, const int* h_dims, int levels,
                                 uint8_t* out, void* stream) {
  if (!pts || !h_tables || !h_dims || !out || n < 0 || levels < 1) return SURF_E_ARG;
  if (levels > MAX_LEVELS) return SURF_E_LIMIT;
  if (n == 0) return 0;
  OccArgs a;
  a.pts = pts; a.n = n; a.levels = levels; a.out = out;
  for (int l = 0; l < levels; ++l) {
    if (!h_tables[l] || h_dims[l] < 1) return SURF_E_ARG;
    a.table[l] = h_tables[l]; a.dim[l] = h_dims[l];
  }
  hipLaunchKernelGGL(occupied_any_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int64_t surf_masked_l1_workspace_bytes(void) { return (int64_t)2 * L1_BLOCKS * sizeof(double); }

static int l1_args(L1Args& a, const float* pred, const float* target, const void* mask, int mask_kind, int64_t n) {
  if (!pred || !target || n <= 0 || mask_kind < 0 || mask_kind > 2 || (mask_kind != MASK_TARGET_POSITIVE && !mask)) return SURF_E_ARG;
  a.pred = pred; a.target = target; a.mask = mask; a.kind = mask_kind; a.n = n;
  a.part = nullptr; a.counter = nullptr; a.out = nullptr;
  return 0;
}

extern "C" int surf_masked_l1(const float* pred, const float* target, const void* mask, int mask_kind, int64_t n, void* workspace,
                              unsigned* counter, float* out2, void* stream) {
  L1Args a;
  if (int rc = l1_args(a, pred, target, mask, mask_kind, n)) return rc;
  if (!workspace || !counter || !out2) return SURF_E_ARG;
  a.part = (double*)workspace; a.counter = counter; a.out = out2;
  const int64_t want = (n + 1023) / 1024;     // >= 4 elements per thread
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > L1_BLOCKS ? L1_BLOCKS : want));
  hipLaunchKernelGGL(masked_l1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

extern "C" int surf_masked_l1_backward(const float* pred, const float* target, const void* mask, int mask_kind, int64_t n,
                                       const float* out2, const float* upstream, float* g_pred, void* stream) {
  L1BwdArgs b;
  if (int rc = l1_args(b.f, pred, target, mask, mask_kind, n)) return rc;
  if (!out2 || !upstream || !g_pred) return SURF_E_ARG;
  b.inv_den = out2 + 1; b.upstream = upstream; b.g_pred = g_pred;
  hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b);
  return surf_check_launch();
}

extern "C" int surf_weight_norm_backward(int layers, const float* const* h_v, const float* const* h_g, const float* const* h_dW,
                                         const int* h_rows, const int* h_cols, float* const* h_dv, float* const* h_dg, void* stream) {
  if (!h_v || !h_g || !h_dW || !h_rows || !h_cols || !h_dv || !h_dg || layers < 1) return SURF_E_ARG;
  if (layers > WN_LAYERS) return SURF_E_LIMIT;
  WnArgs a;
  a.layers = layers;
  int rows = 0;
  for (int l = 0; l < layers; ++l) {
    if (!h_v[l] || !h_g[l] || !h_dW[l] || !h_dv[l] || !h_dg[l] || h_rows[l] < 1 || h_cols[l] < 1) return SURF_E_ARG;
    a.v[l] = h_v[l]; a.g[l] = h_g[l]; a.dW[l] = h_dW[l]; a.dv[l] = h_dv[l]; a.dg[l] = h_dg[l];
    a.cols[l] = h_cols[l]; a.row0[l] = rows;
    rows += h_rows[l];
  }
  a.row0[layers] = rows;
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
