// K11s  second-order term of the SDF network: gradient and H.1 (row sums of the Hessian of the SDF wrt the point).
// Replaces the two chained torch.autograd.grad calls of SDFNetworkSparse.gradient (sdf_network.py:129-152); the second
// one (`smooth`, :143-150) feeds smooth_error (implicit_surface.py:172).  Training-only: the render hot path uses
// sdf_mlp_split.hip, which has no second-order part.
//
// Closed form (oracle/surf_oracle.py sdf_mlp_smooth).  With u = (1,1,1):
//   forward   t_l = W_l in_l + b_l,  t'_l = W_l in'_l      (' = directional derivative along u)
//             h_l = sp(t_l),         h'_l = sp'(t_l) t'_l
//   reverse   g = W_l^T d_l,         g' = W_l^T d'_l
//             d_{l-1} = sp'(t) g,    d'_{l-1} = sp''(t) t' g + sp'(t) g'
//   outputs   grad = J_e^T G_e + J_phi^T G_phi,   H.1 = J_e^T G'_e + J_e'^T G_e + J_phi^T G'_phi + J_phi'^T G_phi
// where J_e' is the second derivative of the positional encoding (diagonal) and J_phi' the mixed second derivatives of
// the trilinear gathers summed over the other two axes (a trilinear cell has no pure second derivative).
//
// One wavefront per SMOOTH_S points (round 6: SURF_TRAIN_WAVES wavefronts per workgroup sharing the weight stream through LDS,
// sdf_train_common.h), plain fp32 FMAs (this term multiplies sigmoid'' = 100 s(1-s) layer after layer, so
// it is kept in full fp32 rather than on the split 16-bit pipes).  Forward: a lane owns output neurons {lane, lane+64}
// and walks the input index k; the weights come from the transposed copy (coalesced), the inputs from LDS (broadcast).
// Reverse: a lane owns input indices {lane, lane+64, lane+128} and walks the neurons.  HBM traffic is the 1 MB weight
// image per wavefront out of L2; 2 x 2 x 99 k MACs per point.
#include "sdf_train_common.h"

// weight rows in flight per step of the k / neuron loops (the loops wait for one L2 round trip per unrolled group)
#ifndef SURF_TRAIN_UNROLL
#define SURF_TRAIN_UNROLL 4
#endif
#define SURF_STR2(x) #x
#define SURF_STR(x) SURF_STR2(x)
#define SURF_TRAIN_UNROLL_PRAGMA _Pragma(SURF_STR(unroll SURF_TRAIN_UNROLL))

namespace {

constexpr int S = 4;          // points per wavefront
constexpr int KP = 160;       // padded input width (156 used)
constexpr int NH = 128;       // hidden width
constexpr int N_E = 27, N_PHI = 28, N_H2 = 101;
constexpr int N_HID = 6;      // layers with an activation (lin0..lin5); lin6 contributes row 0 only
// packed image (floats): Wt[l] (KP x NH, input-major) l = 0..5 | W[l] (NH x KP, neuron-major) l = 0..5 | b[l] (NH) | W6 row 0 (KP)
constexpr int OFF_WT = 0;
constexpr int OFF_W = OFF_WT + N_HID * KP * NH;
constexpr int OFF_B = OFF_W + N_HID * NH * KP;
constexpr int OFF_W6 = OFF_B + N_HID * NH;
constexpr int PACKED_FLOATS = OFF_W6 + KP;

__host__ __device__ constexpr int layer_k(int l) { return l == 0 ? N_E : 156; }
__host__ __device__ constexpr int layer_n(int l) { return l == 2 ? N_H2 : NH; }

constexpr int XS = 2 * S + 4;   // row stride of the transposed LDS operand arrays (floats)
#define XIN(q, s, k) xin_t[(k) * XS + (q) * S + (s)]
#define DL(q, s, k) dl_t[(k) * XS + (q) * S + (s)]
static_assert(S == 4, "one 16-byte LDS read per stream");

struct SmoothArgs {
  const float* pts;
  const int32_t* idx;
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  const float* packed;
  float* grad;
  float* smooth;
};

struct Act { float h, s1, s2; };
// nn.Softplus(beta=100): value, first and second derivative; the linear branch (100 t > 20) has s1 = 1, s2 = 0
__device__ __forceinline__ Act softplus100(float t) {
  const float bt = t * 100.0f;
  Act a;
  if (bt > 20.0f) {
    a.h = t; a.s1 = 1.0f; a.s2 = 0.0f;
  } else {
    const float ex = expf(bt);
    a.h = log1pf(ex) / 100.0f;
    a.s1 = ex / (1.0f + ex);
    a.s2 = 100.0f * a.s1 / (1.0f + ex);
  }
  return a;
}

__global__ __launch_bounds__(surf_train::NT) void sdf_smooth_kernel(SmoothArgs a) {
  // layer input and its derivative along u / adjoint of the pre-activations and its derivative.  Round 5: [k][value | derivative]
  // [sample] rows (8 values + 4 of padding): two 16-byte LDS reads per weight pair instead of eight 4-byte ones (sdf_smooth_bwd.hip)
  __shared__ __attribute__((aligned(16))) float xin_all[surf_train::NW][KP * XS];
  __shared__ __attribute__((aligned(16))) float dl_all[surf_train::NW][NH * XS];
  __shared__ float ge_v_all[surf_train::NW][S][32], ge_d_all[surf_train::NW][S][32];     // skip-layer share of G_e
  __shared__ __attribute__((aligned(16))) float wbuf[surf_train::WBUF_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const xin_t = xin_all[wave];
  float* const dl_t = dl_all[wave];
  float (*const ge_v)[32] = ge_v_all[wave];
  float (*const ge_d)[32] = ge_d_all[wave];
  const int64_t base = ((int64_t)blockIdx.x * surf_train::NW + wave) * S;
  const float inv_sqrt2 = 0.70710678118654752440f;

  // ---- inputs: lane c < 27 owns encoding channel c, lane f < 28 owns gathered feature f --------------------------------
  float e[S], je[S], je2[S];                     // channel value, d/dx_axis, d2/dx_axis^2
  float jp[S][3], mp[S][3];                      // feature Jacobian row and its mixed-derivative row sums
  int64_t pid[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = base + s;
    const bool live = i < a.n;
    pid[s] = live ? (a.idx ? (int64_t)a.idx[i] : i) : -1;
    const float px = live ? a.pts[pid[s] * 3 + 0] : 0.f;
    const float py = live ? a.pts[pid[s] * 3 + 1] : 0.f;
    const float pz = live ? a.pts[pid[s] * 3 + 2] : 0.f;
    // positional encoding (embedder.py:11-36): block 0 = x, block 1+2k = sin(2^k x), block 2+2k = cos(2^k x)
    {
      const int c = lane < N_E ? lane : 0;
      const int axis = c % 3, blk = c / 3;
      const float x = axis == 0 ? px : (axis == 1 ? py : pz);
      if (blk == 0) {
        e[s] = x; je[s] = 1.0f; je2[s] = 0.0f;
      } else {
        const float f = (float)(1 << ((blk - 1) >> 1));
        float sn, cs;
        sincosf(x * f, &sn, &cs);
        if ((blk - 1) & 1) { e[s] = cs; je[s] = -f * sn; je2[s] = -f * f * cs; }
        else               { e[s] = sn; je[s] = f * cs;  je2[s] = -f * f * sn; }
      }
    }
    // sparse trilinear gather (projector.py:217-390) of feature f = 7 level + ch with first and mixed second derivatives
    float phi = 0.f;
    jp[s][0] = jp[s][1] = jp[s][2] = 0.f;
    mp[s][0] = mp[s][1] = mp[s][2] = 0.f;
    if (lane < N_PHI) {
      const int st = lane / 7, ch = lane % 7;
      const int D = a.dims[st];
      if (D > 1) {
        const int32_t* __restrict__ table = a.tables[st];
        const float* __restrict__ vol = a.vols[st];
        const float vs = 2.0f / ((float)D - 1.0f);
        const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
          const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
          const int row = table[((int64_t)xi * D + yi) * D + zi];
          const float f = row >= 0 ? vol[(int64_t)row * 8 + ch] : 0.f;
          const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
          const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
          phi += f * (wx * wy * wz);
          jp[s][0] += f * (sx * wy * wz / vs);
          jp[s][1] += f * (sy * wx * wz / vs);
          jp[s][2] += f * (sz * wx * wy / vs);
          const float dxy = sx * sy * wz / vs / vs, dxz = sx * sz * wy / vs / vs, dyz = sy * sz * wx / vs / vs;
          mp[s][0] += f * (dxy + dxz);
          mp[s][1] += f * (dxy + dyz);
          mp[s][2] += f * (dxz + dyz);
        }
      }
      XIN(0, s, NH + lane) = phi;
      XIN(1, s, NH + lane) = jp[s][0] + jp[s][1] + jp[s][2];
    } else if (lane < KP - NH) {
      XIN(0, s, NH + lane) = 0.f;
      XIN(1, s, NH + lane) = 0.f;
    }
    if (lane < N_E) {
      XIN(0, s, lane) = e[s];
      XIN(1, s, lane) = je[s];
    }
  }
  __syncthreads();

  // ---- forward sweep: keep sp' and sp'' t' of every hidden pre-activation -------------------------------------------------
  float s1[N_HID][2][S], s2t[N_HID][2][S];
#pragma unroll
  for (int l = 0; l < N_HID; ++l) {
    const float* __restrict__ wt = a.packed + OFF_WT + l * KP * NH;
    float acc[2][S], accd[2][S];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < S; ++s) acc[j][s] = accd[j][s] = 0.f;
    const int K = layer_k(l);
    surf_train::stream_rows<NH>(wt, K, wbuf, [&](int k, const float* __restrict__ wr) {
      const float w0 = wr[lane], w1 = wr[64 + lane];
      const f32x4 xv4 = *reinterpret_cast<const f32x4*>(&xin_t[k * XS]), xd4 = *reinterpret_cast<const f32x4*>(&xin_t[k * XS + S]);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float x = xv4[s], xd = xd4[s];
        acc[0][s] = fmaf(w0, x, acc[0][s]);
        acc[1][s] = fmaf(w1, x, acc[1][s]);
        accd[0][s] = fmaf(w0, xd, accd[0][s]);
        accd[1][s] = fmaf(w1, xd, accd[1][s]);
      }
    });
    __syncthreads();                                  // every lane has read in_* before the outputs overwrite it
    const int N = layer_n(l);
    const float post = l == 2 ? inv_sqrt2 : 1.0f;     // lin3's input is cat([h2, e]) / sqrt(2)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nrn = lane + 64 * j;
      const float b = a.packed[OFF_B + l * NH + nrn];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const Act t = softplus100(acc[j][s] + b);
        const bool real = nrn < N;
        s1[l][j][s] = real ? t.s1 : 0.f;
        s2t[l][j][s] = real ? t.s2 * accd[j][s] : 0.f;
        if (real) {
          XIN(0, s, nrn) = t.h * post;
          XIN(1, s, nrn) = t.s1 * accd[j][s] * post;
        } else {                                      // l == 2: slots 101..127 take the encoding of the skip connection
          XIN(0, s, nrn) = 0.f;
          XIN(1, s, nrn) = 0.f;
        }
      }
    }
    if (l == 2) {
      __syncthreads();
      if (lane < N_E) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          XIN(0, s, N_H2 + lane) = e[s] * inv_sqrt2;
          XIN(1, s, N_H2 + lane) = je[s] * inv_sqrt2;
        }
      }
    }
    __syncthreads();
  }

  // ---- reverse sweep -----------------------------------------------------------------------------------------------------
  float gphi[S], gphid[S], gev[S], ged[S];
#pragma unroll
  for (int s = 0; s < S; ++s) gphi[s] = gphid[s] = gev[s] = ged[s] = 0.f;
#pragma unroll
  for (int l = N_HID; l >= 0; --l) {
    float g[3][S], gd[3][S];
    if (l == N_HID) {                                 // d_6 = e_0: g = row 0 of lin6, g' = 0
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k = lane + 64 * j;
        const float w = k < KP ? a.packed[OFF_W6 + k] : 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) { g[j][s] = w; gd[j][s] = 0.f; }
      }
    } else {
      const float* __restrict__ w = a.packed + OFF_W + l * NH * KP;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < S; ++s) g[j][s] = gd[j][s] = 0.f;
      const int N = layer_n(l);
      const bool third = lane < KP - 128;
      surf_train::stream_rows<KP>(w, N, wbuf, [&](int nrn, const float* __restrict__ wr) {
        const float w0 = wr[lane], w1 = wr[64 + lane];
        const float w2 = third ? wr[128 + lane] : 0.f;
        const f32x4 dv4 = *reinterpret_cast<const f32x4*>(&dl_t[nrn * XS]), dd4 = *reinterpret_cast<const f32x4*>(&dl_t[nrn * XS + S]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float d = dv4[s], dd = dd4[s];
          g[0][s] = fmaf(w0, d, g[0][s]);
          g[1][s] = fmaf(w1, d, g[1][s]);
          g[2][s] = fmaf(w2, d, g[2][s]);
          gd[0][s] = fmaf(w0, dd, gd[0][s]);
          gd[1][s] = fmaf(w1, dd, gd[1][s]);
          gd[2][s] = fmaf(w2, dd, gd[2][s]);
        }
      });
      __syncthreads();                                // dl_* fully consumed
    }
    if (l == 0) {                                     // lin0 takes the encoding alone: k = lane < 27
#pragma unroll
      for (int s = 0; s < S; ++s) {
        gev[s] = g[0][s] + (lane < N_E ? ge_v[s][lane] : 0.f);
        ged[s] = gd[0][s] + (lane < N_E ? ge_d[s][lane] : 0.f);
      }
      break;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {                     // feature columns 128..155: lane f < 28
      gphi[s] += g[2][s];
      gphid[s] += gd[2][s];
    }
    const float pre = l == 3 ? inv_sqrt2 : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float gv = g[j][s] * pre, gdv = gd[j][s] * pre;
        if (l == 3 && k >= N_H2) {                    // skip connection: columns 101..127 are the encoding
          ge_v[s][k - N_H2] = gv;
          ge_d[s][k - N_H2] = gdv;
        }
        DL(0, s, k) = s1[l - 1][j][s] * gv;            // s1 = s2t = 0 on the 27 slots lin2 does not have
        DL(1, s, k) = fmaf(s2t[l - 1][j][s], gv, s1[l - 1][j][s] * gdv);
      }
    }
    __syncthreads();
  }

  // ---- outputs -----------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float gr[3] = {0.f, 0.f, 0.f}, sm[3] = {0.f, 0.f, 0.f};
    if (lane < N_E) {
      const int axis = lane % 3;
      const float gq = gev[s] * je[s];
      const float sq = fmaf(ged[s], je[s], gev[s] * je2[s]);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        gr[ax] = axis == ax ? gq : 0.f;
        sm[ax] = axis == ax ? sq : 0.f;
      }
    }
    if (lane < N_PHI) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        gr[ax] = fmaf(gphi[s], jp[s][ax], gr[ax]);
        sm[ax] = fmaf(gphid[s], jp[s][ax], fmaf(gphi[s], mp[s][ax], sm[ax]));
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      gr[ax] = wave_sum(gr[ax]);
      sm[ax] = wave_sum(sm[ax]);
    }
    if (lane < 3 && pid[s] >= 0) {
      if (a.grad) a.grad[pid[s] * 3 + lane] = lane == 0 ? gr[0] : (lane == 1 ? gr[1] : gr[2]);
      a.smooth[pid[s] * 3 + lane] = lane == 0 ? sm[0] : (lane == 1 ? sm[1] : sm[2]);
    }
  }
}

}  // namespace

extern "C" int64_t surf_sdf_smooth_packed_floats(void) { return PACKED_FLOATS; }

// h_W[l] (out_l x in_l row-major effective matrices, weight norm applied), h_b[l], l = 0..6: the same inputs as
// surf_sdf_pack_weights.  Host code.
extern "C" int surf_sdf_smooth_pack_weights(const float* const* h_W, const float* const* h_b, float* out) {
  if (!h_W || !h_b || !out) return SURF_E_ARG;
  for (int l = 0; l < 7; ++l)
    if (!h_W[l] || !h_b[l]) return SURF_E_ARG;
  for (int i = 0; i < PACKED_FLOATS; ++i) out[i] = 0.f;
  for (int l = 0; l < N_HID; ++l) {
    const int K = layer_k(l), N = layer_n(l);
    for (int n = 0; n < N; ++n) {
      for (int k = 0; k < K; ++k) {
        const float w = h_W[l][n * K + k];
        out[OFF_WT + l * KP * NH + k * NH + n] = w;
        out[OFF_W + l * NH * KP + n * KP + k] = w;
      }
      out[OFF_B + l * NH + n] = h_b[l][n];
    }
  }
  for (int k = 0; k < 156; ++k) out[OFF_W6 + k] = h_W[6][k];
  return 0;
}

extern "C" int surf_sdf_smooth(const float* pts, const int32_t* idx, int64_t n, const float* const* h_vols,
                               const int32_t* const* h_tables, const int* h_dims, int n_vol, const float* packed,
                               float* grad, float* smooth, void* stream) {
  if (!pts || !h_vols || !h_tables || !h_dims || !packed || !smooth) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  SmoothArgs a;
  a.pts = pts; a.idx = idx; a.n = n; a.packed = packed; a.grad = grad; a.smooth = smooth;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
  }
  const int64_t blocks = (n + S * surf_train::NW - 1) / (S * surf_train::NW);
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  hipLaunchKernelGGL(sdf_smooth_kernel, dim3((unsigned)blocks), dim3(surf_train::NT), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
