// K10b  backward of the multi-view blending network w.r.t. its parameters (third backward kernel of row f2 / K12).
// The autograd of BlendingNetwork.forward (blending_network.py:69-118) under loss.backward() (runner.py:163) for an upstream
// gradient of the per-sample colour, in closed form.  One wavefront per sample: the forward is recomputed with every
// intermediate kept in LDS (lanes own output neurons, inputs broadcast), then the reverse pass runs view by view with the
// cross-view couplings (softmax over views, weighted mean / variance, the normalised anti-aliasing weights and their min)
// in between.  For every linear layer the kernel writes, per (sample, view), the layer's INPUT vector and the ADJOINT of its
// pre-activation into one row of `rows`; the caller forms dW = adj^T in, db = sum adj as small GEMMs (rocBLAS through
// torch.matmul) and gets d|s| from `ds` (per-sample partials).  The adjoint of the fetched feature channels (f = rgb_feat +
// dfeat: channels 3..18 of the adjoint of f) is scattered through the bilinear taps into `gfeats` when given (generalisation
// training: the FPN receives gradient from the colour path, surf.py:139-146); the source images' and the sample position's
// gradients are not produced (nothing trainable lies behind them).
#include "blend_raw.h"
#include "common.h"

namespace {

using namespace blend_raw;

constexpr int MAXV = SURF_MAX_VIEWS - 1;
// row layout (floats) per (sample, view): [in | adj] of the 11 linear layers, in forward order
constexpr int C_A0_IN = 0, C_A0_AD = 4;          // ray_dir_fc.0   4 -> 16
constexpr int C_A2_IN = 20, C_A2_AD = 36;        // ray_dir_fc.2  16 -> 19
constexpr int C_B0_IN = 55, C_B0_AD = 112;       // base_fc.0     57 -> 64
constexpr int C_B2_IN = 176, C_B2_AD = 240;      // base_fc.2     64 -> 32
constexpr int C_V0_IN = 272, C_V0_AD = 304;      // vis_fc.0      32 -> 32
constexpr int C_V2_IN = 336, C_V2_AD = 368;      // vis_fc.2      32 -> 33
constexpr int C_W0_IN = 401, C_W0_AD = 433;      // vis_fc2.0     32 -> 32
constexpr int C_W2_IN = 465, C_W2_AD = 497;      // vis_fc2.2     32 -> 1
constexpr int C_R0_IN = 498, C_R0_AD = 535;      // rgb_fc.0      37 -> 16
constexpr int C_R2_IN = 551, C_R2_AD = 567;      // rgb_fc.2      16 -> 8
constexpr int C_R4_IN = 575, C_R4_AD = 583;      // rgb_fc.4       8 -> 1
constexpr int ROW = 584;

struct BbArgs {
  const float* pts;
  const int32_t* idx;      // sample indices (n entries) or null
  int64_t n;
  const float* gcolor;     // (n_total, 3) upstream gradient of the sample colours, indexed like pts
  const float* feats[4];
  int hw[8];
  const float* imgs;
  int nv;
  float K[SURF_MAX_VIEWS][9];
  float w2c[SURF_MAX_VIEWS][12];
  float cpos[SURF_MAX_VIEWS][3];
  const float* w;          // raw parameters (blend_raw.h)
  float* rows;             // (n, nv-1, ROW)
  float* ds;               // (n) per-sample d loss / d s
  float* color;            // (n, 3) recomputed forward colour (check against surf_blend), may be null
  float* gfeats[4];        // gradients of the feature maps (texel4, like feats; accumulated by float atomics) or null
};

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expf(x) - 1.0f; }
__device__ __forceinline__ float elu_d(float pre) { return pre > 0.f ? 1.0f : expf(pre); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// per-view LDS record (floats)
enum {
  F_RD = 0, F_F0 = 4, F_A1P = 23, F_A1 = 39, F_DFP = 55, F_F = 74, F_X0 = 93, F_B1P = 150, F_B1 = 214, F_XP = 278, F_X = 310,
  F_XW = 342, F_V1P = 374, F_V1 = 406, F_TP = 438, F_T = 471, F_X2 = 504, F_XV = 536, F_U1P = 568, F_U1 = 600, F_RIN = 632,
  F_R1P = 669, F_R1 = 685, F_R2P = 701, F_R2 = 709, F_SC = 717,   // scalars: +0 mask, +1 ex, +2 w, +3 vis, +4 vis2pre, +5 vis2, +6 r, +7 beta
  F_XB = 728,      // adjoint of x2 / x (32)
  F_FB = 760,      // adjoint of f (19)
  F_LEN = 780
};

// out[o] = b[o] + sum_i W[o][i] in[i]   (lane o < OUT)
template <int IN, int OUT>
__device__ __forceinline__ float matvec(const float* __restrict__ W, const float* __restrict__ b, const float* in, int lane) {
  float acc = 0.f;
  if (lane < OUT) {
    acc = b[lane];
#pragma unroll 4
    for (int i = 0; i < IN; ++i) acc = fmaf(W[lane * IN + i], in[i], acc);
  }
  return acc;
}
// inbar[i] = sum_o W[o][i] adj[o]   (lane i < IN)
template <int IN, int OUT>
__device__ __forceinline__ float matvec_t(const float* __restrict__ W, const float* adj, int lane) {
  float acc = 0.f;
  if (lane < IN) {
#pragma unroll 4
    for (int o = 0; o < OUT; ++o) acc = fmaf(W[o * IN + lane], adj[o], acc);
  }
  return acc;
}

__global__ __launch_bounds__(64) void blend_bwd_kernel(BbArgs a) {
  // (round 5: sized by the launch for the views that exist - V x F_LEN floats instead of MAXV x F_LEN: 16 instead of 22 KB per
  // one-wavefront workgroup at 4 source views, 10 instead of 7 of them per CU)
  extern __shared__ float L_dyn[];
  float (*L)[F_LEN] = reinterpret_cast<float (*)[F_LEN]>(L_dyn);
  __shared__ float meanv[DF], varv[DF], meanb[DF], varb[DF], adj[64], tmp[64];
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  const int V = a.nv - 1;
  const int64_t i = a.idx ? (int64_t)a.idx[s] : s;
  const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];
  const float* __restrict__ w = a.w;
  const float s_abs = fabsf(w[R_S]);

  // ---- fetch (lookup_feature / compute_angle, projector.py:485-556): lane v < V handles source view v --------------------
  if (lane < V) {
    const int cam = lane + 1;
    float* Lv = L[lane];
    float ax = a.cpos[0][0] - px, ay = a.cpos[0][1] - py, az = a.cpos[0][2] - pz;
    float nn = sqrtf(ax * ax + ay * ay + az * az) + 1e-6f;
    ax /= nn; ay /= nn; az /= nn;
    float bx = a.cpos[cam][0] - px, by = a.cpos[cam][1] - py, bz = a.cpos[cam][2] - pz;
    nn = sqrtf(bx * bx + by * by + bz * bz) + 1e-6f;
    bx /= nn; by /= nn; bz /= nn;
    const float ddx = ax - bx, ddy = ay - by, ddz = az - bz;
    const float dn = fmaxf(sqrtf(ddx * ddx + ddy * ddy + ddz * ddz), 1e-6f);
    Lv[F_RD + 0] = ddx / dn; Lv[F_RD + 1] = ddy / dn; Lv[F_RD + 2] = ddz / dn;
    Lv[F_RD + 3] = ax * bx + ay * by + az * bz;
    const float* M = a.w2c[cam];
    const float X = M[0] * px + M[1] * py + M[2] * pz + M[3];
    const float Y = M[4] * px + M[5] * py + M[6] * pz + M[7];
    const float Z = M[8] * px + M[9] * py + M[10] * pz + M[11];
    const float* K = a.K[cam];
    const float qx = K[0] * X + K[1] * Y + K[2] * Z, qy = K[3] * X + K[4] * Y + K[5] * Z, qz = K[6] * X + K[7] * Y + K[8] * Z;
    const float u0 = qx / qz, v0 = qy / qz;
    bool ok = qz > 0.f;
    float sc = 1.0f;
    for (int lv = 0; lv < 4; ++lv) {
      const int H = a.hw[2 * lv], W = a.hw[2 * lv + 1];
      const float u = u0 * sc, vv = v0 * sc;
      ok = ok && (u >= 0.f) && (u < (float)W) && (vv >= 0.f) && (vv < (float)H);
      const float nx = u / ((float)(W - 1) / 2.0f) - 1.0f, ny = vv / ((float)(H - 1) / 2.0f) - 1.0f;
      const float gx = ((nx + 1.0f) * (float)W - 1.0f) / 2.0f, gy = ((ny + 1.0f) * (float)H - 1.0f) / 2.0f;
      if (lv == 0) {
        const f32x4 c = bilinear_texel4(a.imgs + (int64_t)cam * H * W * 4, H, W, gx, gy);
        Lv[F_F0 + 0] = c[0]; Lv[F_F0 + 1] = c[1]; Lv[F_F0 + 2] = c[2];
      }
      const f32x4 t = bilinear_texel4(a.feats[lv] + (int64_t)cam * H * W * 4, H, W, gx, gy);
      Lv[F_F0 + 3 + 4 * lv + 0] = t[0]; Lv[F_F0 + 3 + 4 * lv + 1] = t[1]; Lv[F_F0 + 3 + 4 * lv + 2] = t[2]; Lv[F_F0 + 3 + 4 * lv + 3] = t[3];
      sc *= 0.5f;
    }
    Lv[F_SC + 0] = ok ? 1.0f : 0.0f;
    Lv[F_SC + 1] = expf(s_abs * (Lv[F_RD + 3] - 1.0f));
  }
  __syncthreads();

  // ---- forward A: direction feature, f = rgb_feat + dfeat ---------------------------------------------------------------------
  for (int v = 0; v < V; ++v) {
    float* Lv = L[v];
    const float p1 = matvec<4, 16>(w + R_RD0_W, w + R_RD0_B, Lv + F_RD, lane);
    if (lane < 16) { Lv[F_A1P + lane] = p1; Lv[F_A1 + lane] = elu_f(p1); }
    __syncthreads();
    const float p2 = matvec<16, DF>(w + R_RD2_W, w + R_RD2_B, Lv + F_A1, lane);
    if (lane < DF) { Lv[F_DFP + lane] = p2; Lv[F_F + lane] = Lv[F_F0 + lane] + elu_f(p2); }
    __syncthreads();
  }
  // ---- forward B, C: anti-aliasing weights, weighted mean / variance over views ----------------------------------------------
  float minex = 3.0e38f, sumw = 0.f;
  int argmin = 0;
  for (int v = 0; v < V; ++v) {
    const float e = L[v][F_SC + 1];
    if (e < minex) { minex = e; argmin = v; }
  }
  for (int v = 0; v < V; ++v) sumw += (L[v][F_SC + 1] - minex) * L[v][F_SC + 0];
  const float Sd = sumw + 1e-8f;
  if (lane < V) L[lane][F_SC + 2] = (L[lane][F_SC + 1] - minex) * L[lane][F_SC + 0] / Sd;
  __syncthreads();
  if (lane < DF) {
    float m = 0.f, q = 0.f;
    for (int v = 0; v < V; ++v) m += L[v][F_F + lane] * L[v][F_SC + 2];
    for (int v = 0; v < V; ++v) { const float d = L[v][F_F + lane] - m; q += L[v][F_SC + 2] * d * d; }
    meanv[lane] = m;
    varv[lane] = q;
  }
  __syncthreads();
  // ---- forward D..G per view ----------------------------------------------------------------------------------------------------
  for (int v = 0; v < V; ++v) {
    float* Lv = L[v];
    const float mk = Lv[F_SC + 0], wv = Lv[F_SC + 2];
    if (lane < DF) { Lv[F_X0 + lane] = meanv[lane]; Lv[F_X0 + DF + lane] = varv[lane]; Lv[F_X0 + 2 * DF + lane] = Lv[F_F + lane]; }
    __syncthreads();
    float p = matvec<57, 64>(w + R_B0_W, w + R_B0_B, Lv + F_X0, lane);
    Lv[F_B1P + lane] = p; Lv[F_B1 + lane] = elu_f(p);
    __syncthreads();
    p = matvec<64, 32>(w + R_B2_W, w + R_B2_B, Lv + F_B1, lane);
    if (lane < 32) { Lv[F_XP + lane] = p; const float x = elu_f(p); Lv[F_X + lane] = x; Lv[F_XW + lane] = x * wv; }
    __syncthreads();
    p = matvec<32, 32>(w + R_V0_W, w + R_V0_B, Lv + F_XW, lane);
    if (lane < 32) { Lv[F_V1P + lane] = p; Lv[F_V1 + lane] = elu_f(p); }
    __syncthreads();
    p = matvec<32, 33>(w + R_V2_W, w + R_V2_B, Lv + F_V1, lane);
    if (lane < 33) { Lv[F_TP + lane] = p; Lv[F_T + lane] = elu_f(p); }
    __syncthreads();
    const float vis = sigm(Lv[F_T + 32]) * mk;
    if (lane < 32) { const float x2 = Lv[F_X + lane] + Lv[F_T + lane]; Lv[F_X2 + lane] = x2; Lv[F_XV + lane] = x2 * vis; }
    if (lane == 0) Lv[F_SC + 3] = vis;
    __syncthreads();
    p = matvec<32, 32>(w + R_W0_W, w + R_W0_B, Lv + F_XV, lane);
    if (lane < 32) { Lv[F_U1P + lane] = p; Lv[F_U1 + lane] = elu_f(p); }
    __syncthreads();
    p = matvec<32, 1>(w + R_W2_W, w + R_W2_B, Lv + F_U1, lane);
    if (lane == 0) { Lv[F_SC + 4] = p; Lv[F_SC + 5] = sigm(p) * mk; }
    __syncthreads();
    if (lane < 32) Lv[F_RIN + lane] = Lv[F_X2 + lane];
    if (lane == 32) Lv[F_RIN + 32] = Lv[F_SC + 5];
    if (lane < 4) Lv[F_RIN + 33 + lane] = Lv[F_RD + lane];
    __syncthreads();
    p = matvec<37, 16>(w + R_R0_W, w + R_R0_B, Lv + F_RIN, lane);
    if (lane < 16) { Lv[F_R1P + lane] = p; Lv[F_R1 + lane] = elu_f(p); }
    __syncthreads();
    p = matvec<16, 8>(w + R_R2_W, w + R_R2_B, Lv + F_R1, lane);
    if (lane < 8) { Lv[F_R2P + lane] = p; Lv[F_R2 + lane] = elu_f(p); }
    __syncthreads();
    p = matvec<8, 1>(w + R_R4_W, w + R_R4_B, Lv + F_R2, lane);
    if (lane == 0) Lv[F_SC + 6] = mk == 0.f ? -1e9f : p;
    __syncthreads();
  }
  // softmax over views and the blended colour
  float rmax = -3.0e38f, esum = 0.f;
  for (int v = 0; v < V; ++v) rmax = fmaxf(rmax, L[v][F_SC + 6]);
  for (int v = 0; v < V; ++v) esum += expf(L[v][F_SC + 6] - rmax);
  if (lane < V) L[lane][F_SC + 7] = expf(L[lane][F_SC + 6] - rmax) / esum;
  __syncthreads();
  const float gc[3] = {a.gcolor[i * 3 + 0], a.gcolor[i * 3 + 1], a.gcolor[i * 3 + 2]};
  if (a.color && lane < 3) {
    float c = 0.f;
    for (int v = 0; v < V; ++v) c += L[v][F_SC + 7] * L[v][F_F0 + lane];
    a.color[s * 3 + lane] = c;
  }
  float bsum = 0.f;       // sum_u beta_u betabar_u
  for (int v = 0; v < V; ++v)
    bsum += L[v][F_SC + 7] * (gc[0] * L[v][F_F0 + 0] + gc[1] * L[v][F_F0 + 1] + gc[2] * L[v][F_F0 + 2]);

  // ---- reverse G..D per view ---------------------------------------------------------------------------------------------------
  if (lane < DF) { meanb[lane] = 0.f; varb[lane] = 0.f; }
  float wbar_l = 0.f;     // lane v < V: adjoint of w_v
  __syncthreads();
  for (int v = 0; v < V; ++v) {
    float* Lv = L[v];
    float* row = a.rows + ((int64_t)s * V + v) * ROW;
    const float mk = Lv[F_SC + 0], wv = Lv[F_SC + 2], vis = Lv[F_SC + 3];
    const float bbar = gc[0] * Lv[F_F0 + 0] + gc[1] * Lv[F_F0 + 1] + gc[2] * Lv[F_F0 + 2];
    const float rbar = mk == 0.f ? 0.f : Lv[F_SC + 7] * (bbar - bsum);
    // rgb_fc
    if (lane == 0) { adj[0] = rbar; row[C_R4_AD] = rbar; }
    if (lane < 8) row[C_R4_IN + lane] = Lv[F_R2 + lane];
    __syncthreads();
    float g = matvec_t<8, 1>(w + R_R4_W, adj, lane);
    __syncthreads();
    if (lane < 8) { g *= elu_d(Lv[F_R2P + lane]); adj[lane] = g; row[C_R2_AD + lane] = g; }
    if (lane < 16) row[C_R2_IN + lane] = Lv[F_R1 + lane];
    __syncthreads();
    g = matvec_t<16, 8>(w + R_R2_W, adj, lane);
    __syncthreads();
    if (lane < 16) { g *= elu_d(Lv[F_R1P + lane]); adj[lane] = g; row[C_R0_AD + lane] = g; }
    if (lane < 37) row[C_R0_IN + lane] = Lv[F_RIN + lane];
    __syncthreads();
    g = matvec_t<37, 16>(w + R_R0_W, adj, lane);             // adjoint of [x2 (32), vis2, ray_diff (4)]
    __syncthreads();
    if (lane < 32) Lv[F_XB + lane] = g;
    if (lane == 32) tmp[0] = g;                               // vis2bar
    __syncthreads();
    // vis_fc2
    const float z = Lv[F_SC + 4], sz = sigm(z);
    const float zbar = tmp[0] * sz * (1.0f - sz) * mk;
    __syncthreads();
    if (lane == 0) { adj[0] = zbar; row[C_W2_AD] = zbar; }
    if (lane < 32) row[C_W2_IN + lane] = Lv[F_U1 + lane];
    __syncthreads();
    g = matvec_t<32, 1>(w + R_W2_W, adj, lane);
    __syncthreads();
    if (lane < 32) { g *= elu_d(Lv[F_U1P + lane]); adj[lane] = g; row[C_W0_AD + lane] = g; row[C_W0_IN + lane] = Lv[F_XV + lane]; }
    __syncthreads();
    g = matvec_t<32, 32>(w + R_W0_W, adj, lane);             // adjoint of x2 * vis
    __syncthreads();
    float visbar = lane < 32 ? g * Lv[F_X2 + lane] : 0.f;
    visbar = wave_sum(visbar);
    if (lane < 32) Lv[F_XB + lane] += g * vis;               // x2bar complete
    __syncthreads();
    // vis_fc: t = elu(tpre) (33), x2 = x + t[:32], vis = sigmoid(t[32]) m
    const float t32 = Lv[F_T + 32], st = sigm(t32);
    if (lane < 33) {
      const float tb = lane < 32 ? Lv[F_XB + lane] : visbar * st * (1.0f - st) * mk;
      const float d = tb * elu_d(Lv[F_TP + lane]);
      adj[lane] = d;
      row[C_V2_AD + lane] = d;
    }
    if (lane < 32) row[C_V2_IN + lane] = Lv[F_V1 + lane];
    __syncthreads();
    g = matvec_t<32, 33>(w + R_V2_W, adj, lane);
    __syncthreads();
    if (lane < 32) { g *= elu_d(Lv[F_V1P + lane]); adj[lane] = g; row[C_V0_AD + lane] = g; row[C_V0_IN + lane] = Lv[F_XW + lane]; }
    __syncthreads();
    g = matvec_t<32, 32>(w + R_V0_W, adj, lane);             // adjoint of x * w_v
    __syncthreads();
    float wb = lane < 32 ? g * Lv[F_X + lane] : 0.f;
    wb = wave_sum(wb);
    if (lane == v) wbar_l += wb;
    // x = elu(xpre): xbar = x2bar + (x w)bar w
    if (lane < 32) {
      const float xb = Lv[F_XB + lane] + g * wv;
      const float d = xb * elu_d(Lv[F_XP + lane]);
      adj[lane] = d;
      row[C_B2_AD + lane] = d;
    }
    row[C_B2_IN + lane] = Lv[F_B1 + lane];
    __syncthreads();
    g = matvec_t<64, 32>(w + R_B2_W, adj, lane);
    __syncthreads();
    g *= elu_d(Lv[F_B1P + lane]);
    adj[lane] = g;
    row[C_B0_AD + lane] = g;
    if (lane < 57) row[C_B0_IN + lane] = Lv[F_X0 + lane];
    __syncthreads();
    g = matvec_t<57, 64>(w + R_B0_W, adj, lane);             // adjoint of [mean, var, f]
    __syncthreads();
    if (lane < DF) meanb[lane] += g;
    else if (lane < 2 * DF) varb[lane - DF] += g;
    else if (lane < 3 * DF) Lv[F_FB + lane - 2 * DF] = g;
    __syncthreads();
  }
  // ---- reverse C: mean / variance -------------------------------------------------------------------------------------------------
  //   var = sum_v w_v (f_v - mean)^2, mean = sum_v w_v f_v
  if (lane < DF) {
    float mb = meanb[lane];
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc += 2.0f * L[v][F_SC + 2] * (L[v][F_F + lane] - meanv[lane]);
    mb -= varb[lane] * acc;                                   // through (f - mean)
    for (int v = 0; v < V; ++v) {
      const float d = L[v][F_F + lane] - meanv[lane];
      L[v][F_FB + lane] += varb[lane] * 2.0f * L[v][F_SC + 2] * d + mb * L[v][F_SC + 2];
    }
    meanb[lane] = mb;
  }
  __syncthreads();
  for (int v = 0; v < V; ++v) {
    float t = 0.f;
    if (lane < DF) {
      const float d = L[v][F_F + lane] - meanv[lane];
      t = varb[lane] * d * d + meanb[lane] * L[v][F_F + lane];
    }
    t = wave_sum(t);
    if (lane == v) wbar_l += t;
  }
  // ---- reverse B: w_v = what_v / (sum what + 1e-8), what_v = (ex_v - min ex) m_v, ex_v = exp(|s| (dot_v - 1)) ---------------
  {
    const float what = lane < V ? (L[lane][F_SC + 1] - minex) * L[lane][F_SC + 0] : 0.f;
    const float dotsum = wave_sum(lane < V ? wbar_l * what : 0.f);
    const float whatbar = lane < V ? wbar_l / Sd - dotsum / (Sd * Sd) : 0.f;
    const float exbar_own = lane < V ? whatbar * L[lane][F_SC + 0] : 0.f;
    const float minbar = -wave_sum(exbar_own);
    float exbar = exbar_own + ((lane < V && lane == argmin) ? minbar : 0.f);
    float sb = lane < V ? exbar * L[lane][F_SC + 1] * (L[lane][F_RD + 3] - 1.0f) : 0.f;
    sb = wave_sum(sb);
    if (lane == 0) a.ds[s] = sb * (w[R_S] < 0.f ? -1.0f : 1.0f);
  }
  // ---- the fetched features' share of the adjoint of f -> the FPN maps (lane = view-in-chunk x level x tap, 4 channels each) ----
  __syncthreads();
  if (a.gfeats[0]) {
    for (int v0 = 0; v0 < V; v0 += 4) {
      const int v = v0 + (lane >> 4), lv = (lane >> 2) & 3, tap = lane & 3;
      if (v < V) {
        const int cam = v + 1;
        const float* M = a.w2c[cam];
        const float X = M[0] * px + M[1] * py + M[2] * pz + M[3];
        const float Y = M[4] * px + M[5] * py + M[6] * pz + M[7];
        const float Z = M[8] * px + M[9] * py + M[10] * pz + M[11];
        const float* K = a.K[cam];
        const float qx = K[0] * X + K[1] * Y + K[2] * Z, qy = K[3] * X + K[4] * Y + K[5] * Z, qz = K[6] * X + K[7] * Y + K[8] * Z;
        float sc = 1.0f;
        for (int q = 0; q < lv; ++q) sc *= 0.5f;
        const int H = a.hw[2 * lv], W = a.hw[2 * lv + 1];
        const float u = (qx / qz) * sc, vv = (qy / qz) * sc;
        const float nx = u / ((float)(W - 1) / 2.0f) - 1.0f, ny = vv / ((float)(H - 1) / 2.0f) - 1.0f;
        const float gx = ((nx + 1.0f) * (float)W - 1.0f) / 2.0f, gy = ((ny + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float fx = floorf(gx), fy = floorf(gy);
        const int xi = (int)fx + (tap & 1), yi = (int)fy + (tap >> 1);
        const float wgt = ((tap & 1) ? gx - fx : 1.0f - (gx - fx)) * ((tap >> 1) ? gy - fy : 1.0f - (gy - fy));
        if ((xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) && wgt != 0.f) {
          float* dst = a.gfeats[lv] + (((int64_t)cam * H + yi) * W + xi) * 4;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float g = L[v][F_FB + 3 + 4 * lv + c] * wgt;
            if (g != 0.f) atomicAdd(dst + c, g);
          }
        }
      }
    }
  }
  // ---- reverse A: f = f0 + elu(dfpre) --------------------------------------------------------------------------------------------
  __syncthreads();
  for (int v = 0; v < V; ++v) {
    float* Lv = L[v];
    float* row = a.rows + ((int64_t)s * V + v) * ROW;
    if (lane < DF) { const float d = Lv[F_FB + lane] * elu_d(Lv[F_DFP + lane]); adj[lane] = d; row[C_A2_AD + lane] = d; }
    if (lane < 16) row[C_A2_IN + lane] = Lv[F_A1 + lane];
    __syncthreads();
    float g = matvec_t<16, DF>(w + R_RD2_W, adj, lane);
    __syncthreads();
    if (lane < 16) { g *= elu_d(Lv[F_A1P + lane]); row[C_A0_AD + lane] = g; }
    if (lane < 4) row[C_A0_IN + lane] = Lv[F_RD + lane];
    __syncthreads();
  }
}

}  // namespace

extern "C" int surf_blend_backward_row_floats(void) { return ROW; }

extern "C" int surf_blend_backward(const float* pts, const int32_t* idx, int64_t n, const float* gcolor,
                                   const float* const* h_feats_t4, const int* h_feat_hw, const float* imgs_t4, int nv,
                                   const float* h_intrs, const float* h_w2c, const float* h_c2w, const float* raw_weights,
                                   float* rows, float* ds, float* color, float* const* h_gfeats_t4, void* stream) {
  if (!pts || !gcolor || !h_feats_t4 || !h_feat_hw || !imgs_t4 || !h_intrs || !h_w2c || !h_c2w || !raw_weights || !rows || !ds)
    return SURF_E_ARG;
  if (n <= 0 || nv < 2) return SURF_E_ARG;
  if (nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;
  BbArgs a;
  a.pts = pts; a.idx = idx; a.n = n; a.gcolor = gcolor; a.imgs = imgs_t4; a.nv = nv; a.w = raw_weights; a.rows = rows; a.ds = ds;
  a.color = color;
  for (int l = 0; l < 4; ++l) {
    if (!h_feats_t4[l]) return SURF_E_ARG;
    a.feats[l] = h_feats_t4[l];
    a.hw[2 * l] = h_feat_hw[2 * l];
    a.hw[2 * l + 1] = h_feat_hw[2 * l + 1];
    a.gfeats[l] = h_gfeats_t4 ? h_gfeats_t4[l] : nullptr;
    if (h_gfeats_t4 && !h_gfeats_t4[l]) return SURF_E_ARG;
  }
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int sidx = v < nv ? v : 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) a.K[v][r * 3 + c] = h_intrs[sidx * 16 + r * 4 + c];
    for (int k = 0; k < 12; ++k) a.w2c[v][k] = h_w2c[sidx * 16 + k];
    for (int r = 0; r < 3; ++r) a.cpos[v][r] = h_c2w[sidx * 16 + r * 4 + 3];
  }
  if (n > 0x7fffffff) return SURF_E_LIMIT;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3((unsigned)n), dim3(64), (size_t)(nv - 1) * F_LEN * sizeof(float), (hipStream_t)stream, a);
  return surf_check_launch();
}
