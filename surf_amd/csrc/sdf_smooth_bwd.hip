// K9c  backward of the smooth (H.1) loss term through the SDF network (row f2 / K12; the last loss term of losses/loss.py:40).
// smooth = H u with u = (1,1,1) (the second autograd.grad of sdf_network.py:143-150); for an upstream gradient s (3-vector per
// sample, d loss / d smooth) the term is  s . H u  =  D_u D_s y,  the MIXED second directional derivative of the SDF value y
// along u and s.  One reverse sweep over a forward sweep that carries (value, tangent along u, tangent along s, mixed tangent)
// - reverse over forward-over-forward - replaces the reference's triple backward (loss.backward() through two
// create_graph=True autograd.grad calls).  With sp1, sp2, sp3 the first three derivatives of the softplus:
//   forward   a = W x + b,  a_u = W x_u,  a_s = W x_s,  a_m = W x_m
//             h = sp(a),  h_u = sp1(a) a_u,  h_s = sp1(a) a_s,  h_m = sp2(a) a_u a_s + sp1(a) a_m
//   seed      adjoint of lin6's mixed input = lin6 row 0   (S = w6 . x_m: the bias and the other streams do not reach it)
//   reverse   c1 = sp1, cu = sp2 a_u, cs = sp2 a_s, cm = sp3 a_u a_s + sp2 a_m:
//             abar_m = c1 hbar_m;  abar_u = c1 hbar_u + cs hbar_m;  abar_s = c1 hbar_s + cu hbar_m;
//             abar   = c1 hbar + cu hbar_u + cs hbar_s + cm hbar_m
//   weights   dW_l = sum_n over the four streams of  abar_* (x) x_*,  db_l = sum_n abar   (GEMMs over the per-sample buffers
//             this kernel writes: IN (7, 4, n, 160) and AB (6, 4, n, 128), stream order value | u | s | mixed: a layer's four
//             streams are 4 n contiguous rows, one tall-skinny GEMM per layer)
//   features  dF[row_c] += w_c pbar + (grad w_c . u) pbar_u + (grad w_c . s) pbar_s + (u^T Hess w_c s) pbar_m
// Plain fp32 FMAs, one wavefront per 4 samples, same lane ownership and packed weight image as sdf_bwd.hip / sdf_smooth.hip.
#include <stdlib.h>

#include "common.h"

// weight rows in flight per step of the k / neuron loops (the loops wait for one L2 round trip per unrolled group)
#ifndef SURF_TRAIN_UNROLL
#define SURF_TRAIN_UNROLL 4
#endif
#define SURF_STR2(x) #x
#define SURF_STR(x) SURF_STR2(x)
#define SURF_TRAIN_UNROLL_PRAGMA _Pragma(SURF_STR(unroll SURF_TRAIN_UNROLL))

namespace {

constexpr int S = 4, KP = 160, NH = 128, N_E = 27, N_PHI = 28, N_H2 = 101, N_HID = 6, NS = 4;   // NS streams
constexpr int OFF_WT = 0;
constexpr int OFF_W = OFF_WT + N_HID * KP * NH;
constexpr int OFF_B = OFF_W + N_HID * NH * KP;
constexpr int OFF_W6 = OFF_B + N_HID * NH;

__host__ __device__ constexpr int layer_k(int l) { return l == 0 ? N_E : 156; }
__host__ __device__ constexpr int layer_n(int l) { return l == 2 ? N_H2 : NH; }

constexpr int XS = NS * S + 4;   // row stride of the transposed LDS operand arrays (floats): 16-byte aligned, 8 banks apart
#define XIN(q, s, k) xin_t[(k) * XS + (q) * S + (s)]
#define DL(q, s, k) dl_t[(k) * XS + (q) * S + (s)]
static_assert(S == 4, "one 16-byte LDS read per stream");

struct SmBwdArgs {
  const float* pts;
  const float* sbar;   // (n,3) d loss / d smooth
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  float* dvols[SURF_MAX_STAGES];   // gradient rows (N_s, 8), accumulated with atomics
  const float* packed;
  float* in;     // (7, 4, n, KP)  layer inputs of the four streams
  float* ab;     // (6, 4, n, NH)  adjoints of the four pre-activation streams
};

struct Act3 { float h, s1, s2, s3; };
__device__ __forceinline__ Act3 softplus100_3(float t) {
  const float bt = t * 100.0f;
  Act3 a;
  if (bt > 20.0f) {
    a.h = t; a.s1 = 1.0f; a.s2 = 0.0f; a.s3 = 0.0f;
  } else {
    const float ex = expf(bt);
    a.h = log1pf(ex) / 100.0f;
    a.s1 = ex / (1.0f + ex);
    a.s2 = 100.0f * a.s1 / (1.0f + ex);
    a.s3 = 100.0f * a.s2 * (1.0f - 2.0f * a.s1);
  }
  return a;
}

__global__ __launch_bounds__(64) void sdf_smooth_bwd_kernel(SmBwdArgs a) {
  // Round 5: [k][stream][sample] rows (16 values + 4 of padding) instead of [stream][sample][k]: the k / neuron loops below read
  // the 16 broadcast operands of a weight pair as FOUR 16-byte LDS reads instead of sixteen 4-byte ones (the loops were bound by
  // their LDS instructions: 16 ds_read per 32 FMAs, at one wavefront per SIMD).
  __shared__ __attribute__((aligned(16))) float xin_t[KP * XS];
  __shared__ __attribute__((aligned(16))) float dl_t[NH * XS];
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * S;
  const float inv_sqrt2 = 0.70710678118654752440f;
  float e[NS][S];
  float px[S], py[S], pz[S], vx[S], vy[S], vz[S];
  bool live[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = base + s;
    live[s] = i < a.n;
    const int64_t ic = live[s] ? i : a.n - 1;
    px[s] = a.pts[ic * 3 + 0]; py[s] = a.pts[ic * 3 + 1]; pz[s] = a.pts[ic * 3 + 2];
    vx[s] = live[s] ? a.sbar[ic * 3 + 0] : 0.f; vy[s] = live[s] ? a.sbar[ic * 3 + 1] : 0.f; vz[s] = live[s] ? a.sbar[ic * 3 + 2] : 0.f;
    {
      const int c = lane < N_E ? lane : 0;
      const int axis = c % 3, blk = c / 3;
      const float x = axis == 0 ? px[s] : (axis == 1 ? py[s] : pz[s]);
      float e0, e1, e2;                              // channel, d/dx, d2/dx2 (each channel depends on one coordinate)
      if (blk == 0) {
        e0 = x; e1 = 1.0f; e2 = 0.0f;
      } else {
        const float f = (float)(1 << ((blk - 1) >> 1));
        float sn, cs;
        sincosf(x * f, &sn, &cs);
        if ((blk - 1) & 1) { e0 = cs; e1 = -f * sn; e2 = -f * f * cs; }
        else               { e0 = sn; e1 = f * cs; e2 = -f * f * sn; }
      }
      const float va = axis == 0 ? vx[s] : (axis == 1 ? vy[s] : vz[s]);
      e[0][s] = e0; e[1][s] = e1; e[2][s] = e1 * va; e[3][s] = e2 * va;       // u = (1,1,1)
    }
    float phi[NS] = {0.f, 0.f, 0.f, 0.f};
    if (lane < N_PHI) {
      const int st = lane / 7, ch = lane % 7;
      const int D = a.dims[st];
      if (D > 1) {
        const int32_t* __restrict__ table = a.tables[st];
        const float* __restrict__ vol = a.vols[st];
        const float vs = 2.0f / ((float)D - 1.0f);
        const float gx = (px[s] + 1.0f) / vs, gy = (py[s] + 1.0f) / vs, gz = (pz[s] + 1.0f) / vs;
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
          const float sx = (dx ? 1.0f : -1.0f) / vs, sy = (dy ? 1.0f : -1.0f) / vs, sz = (dz ? 1.0f : -1.0f) / vs;
          const float gwx = sx * wy * wz, gwy = sy * wx * wz, gwz = sz * wx * wy;          // grad w_c
          const float hxy = sx * sy * wz, hxz = sx * sz * wy, hyz = sy * sz * wx;          // mixed second derivatives
          phi[0] += f * (wx * wy * wz);
          phi[1] += f * (gwx + gwy + gwz);
          phi[2] += f * (gwx * vx[s] + gwy * vy[s] + gwz * vz[s]);
          phi[3] += f * (hxy * (vx[s] + vy[s]) + hxz * (vx[s] + vz[s]) + hyz * (vy[s] + vz[s]));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      XIN(q, s, lane) = 0.f; XIN(q, s, lane + 64) = 0.f;                         // columns 0..127
      if (lane < KP - NH) XIN(q, s, NH + lane) = lane < N_PHI ? phi[q] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (lane < N_E) {
#pragma unroll
      for (int q = 0; q < NS; ++q) XIN(q, s, lane) = e[q][s];
    }
  __syncthreads();
  auto dump_inputs = [&](int l) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int s = 0; s < S; ++s)
        if (live[s]) {
          const int64_t o = (((int64_t)l * NS + q) * a.n + base + s) * KP;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int k = lane + 64 * j;
            if (k < KP) a.in[o + k] = XIN(q, s, k);
          }
        }
  };

  // ---- forward sweep with the three tangent streams --------------------------------------------------------------------
  // Round 5: the four softplus-coefficient streams c1 | cu | cs | cm of every layer (192 values a lane) wait for the reverse sweep
  // in the AB buffer itself - AB[l] (4, n, 128) is exactly their shape and is only written when the reverse sweep reaches layer
  // l, by the same lane at the same addresses - instead of in registers: 256 + 99 registers (one wavefront per SIMD) -> see the
  // resource remarks; the weight loads of the k / neuron loops are the latency the extra wavefronts hide.
  const int64_t qs = a.n * NH;
#pragma unroll
  for (int l = 0; l < N_HID; ++l) {
    dump_inputs(l);
    const float* __restrict__ wt = a.packed + OFF_WT + l * KP * NH;
    float acc[NS][2][S];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < S; ++s) acc[q][j][s] = 0.f;
    const int K = layer_k(l);
SURF_TRAIN_UNROLL_PRAGMA
    for (int k = 0; k < K; ++k) {
      const float w0 = wt[k * NH + lane], w1 = wt[k * NH + 64 + lane];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const f32x4 xq = *reinterpret_cast<const f32x4*>(&xin_t[k * XS + q * S]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
          acc[q][0][s] = fmaf(w0, xq[s], acc[q][0][s]);
          acc[q][1][s] = fmaf(w1, xq[s], acc[q][1][s]);
        }
      }
    }
    __syncthreads();
    const int N = layer_n(l);
    const float post = l == 2 ? inv_sqrt2 : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nrn = lane + 64 * j;
      const float b = a.packed[OFF_B + l * NH + nrn];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const Act3 t = softplus100_3(acc[0][j][s] + b);
        const bool real = nrn < N;
        const float au = acc[1][j][s], as = acc[2][j][s], am = acc[3][j][s];
        if (live[s]) {
          const int64_t o = ((int64_t)l * NS * a.n + base + s) * NH + nrn;
          a.ab[o] = real ? t.s1 : 0.f;
          a.ab[o + qs] = real ? t.s2 * au : 0.f;
          a.ab[o + 2 * qs] = real ? t.s2 * as : 0.f;
          a.ab[o + 3 * qs] = real ? fmaf(t.s3 * au, as, t.s2 * am) : 0.f;
        }
        XIN(0, s, nrn) = real ? t.h * post : 0.f;
        XIN(1, s, nrn) = real ? t.s1 * au * post : 0.f;
        XIN(2, s, nrn) = real ? t.s1 * as * post : 0.f;
        XIN(3, s, nrn) = real ? fmaf(t.s2 * au, as, t.s1 * am) * post : 0.f;
      }
    }
    if (l == 2) {
      __syncthreads();
      if (lane < N_E) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int s = 0; s < S; ++s) XIN(q, s, N_H2 + lane) = e[q][s] * inv_sqrt2;
      }
    }
    __syncthreads();
  }
  dump_inputs(N_HID);   // inputs of lin6 (only its mixed stream reaches the term)

  // ---- reverse sweep --------------------------------------------------------------------------------------------------------
  float pbar[NS][S];   // adjoints of the four feature streams for lane f < 28
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int s = 0; s < S; ++s) pbar[q][s] = 0.f;
#pragma unroll
  for (int l = N_HID; l >= 1; --l) {
    float ck[2][S][NS];   // c1 | cu | cs | cm of layer l - 1 for this lane's two neurons (dead samples: zeros)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int64_t o = ((int64_t)(l - 1) * NS * a.n + base + (live[s] ? s : 0)) * NH + lane + 64 * j;
#pragma unroll
        for (int q = 0; q < NS; ++q) ck[j][s][q] = live[s] ? a.ab[o + q * qs] : 0.f;
      }
    float g[NS][3][S];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < S; ++s) g[q][j][s] = 0.f;
    if (l == N_HID) {   // S = w6 . x_m: the adjoint of lin6's mixed input is lin6 row 0
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k = lane + 64 * j;
        const float w = k < KP ? a.packed[OFF_W6 + k] : 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) g[3][j][s] = live[s] ? w : 0.f;
      }
    } else {
      const float* __restrict__ w = a.packed + OFF_W + l * NH * KP;
      const int N = layer_n(l);
      const bool third = lane < KP - 128;
SURF_TRAIN_UNROLL_PRAGMA
      for (int nrn = 0; nrn < N; ++nrn) {
        const float w0 = w[nrn * KP + lane], w1 = w[nrn * KP + 64 + lane];
        const float w2 = third ? w[nrn * KP + 128 + lane] : 0.f;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const f32x4 dq = *reinterpret_cast<const f32x4*>(&dl_t[nrn * XS + q * S]);
#pragma unroll
          for (int s = 0; s < S; ++s) {
            g[q][0][s] = fmaf(w0, dq[s], g[q][0][s]);
            g[q][1][s] = fmaf(w1, dq[s], g[q][1][s]);
            g[q][2][s] = fmaf(w2, dq[s], g[q][2][s]);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int s = 0; s < S; ++s) pbar[q][s] += g[q][2][s];
    const float pre = l == 3 ? inv_sqrt2 : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float hb = g[0][j][s] * pre, hub = g[1][j][s] * pre, hsb = g[2][j][s] * pre, hmb = g[3][j][s] * pre;
        const float k1 = ck[j][s][0], ku = ck[j][s][1], ks = ck[j][s][2], km = ck[j][s][3];
        const float ab0 = fmaf(km, hmb, fmaf(ks, hsb, fmaf(ku, hub, k1 * hb)));
        const float ab1 = fmaf(ks, hmb, k1 * hub);
        const float ab2 = fmaf(ku, hmb, k1 * hsb);
        const float ab3 = k1 * hmb;
        DL(0, s, k) = ab0; DL(1, s, k) = ab1; DL(2, s, k) = ab2; DL(3, s, k) = ab3;
        if (live[s]) {
          const int64_t o = ((int64_t)(l - 1) * NS * a.n + base + s) * NH + k;
          a.ab[o] = ab0; a.ab[o + qs] = ab1; a.ab[o + 2 * qs] = ab2; a.ab[o + 3 * qs] = ab3;
        }
      }
    }
    __syncthreads();
  }

  // ---- feature gradients ------------------------------------------------------------------------------------------------------
  if (lane < N_PHI) {
    const int st = lane / 7, ch = lane % 7;
    const int D = a.dims[st];
    if (D > 1 && a.dvols[st]) {
      const int32_t* __restrict__ table = a.tables[st];
      float* __restrict__ dvol = a.dvols[st];
      const float vs = 2.0f / ((float)D - 1.0f);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (!live[s]) continue;
        const float gx = (px[s] + 1.0f) / vs, gy = (py[s] + 1.0f) / vs, gz = (pz[s] + 1.0f) / vs;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
          const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
          const int row = table[((int64_t)xi * D + yi) * D + zi];
          if (row < 0) continue;
          const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
          const float sx = (dx ? 1.0f : -1.0f) / vs, sy = (dy ? 1.0f : -1.0f) / vs, sz = (dz ? 1.0f : -1.0f) / vs;
          const float gwx = sx * wy * wz, gwy = sy * wx * wz, gwz = sz * wx * wy;
          const float hxy = sx * sy * wz, hxz = sx * sz * wy, hyz = sy * sz * wx;
          const float w0 = wx * wy * wz, wu = gwx + gwy + gwz, ws = gwx * vx[s] + gwy * vy[s] + gwz * vz[s];
          const float wm = hxy * (vx[s] + vy[s]) + hxz * (vx[s] + vz[s]) + hyz * (vy[s] + vz[s]);
          atomicAdd(dvol + (int64_t)row * 8 + ch, w0 * pbar[0][s] + wu * pbar[1][s] + ws * pbar[2][s] + wm * pbar[3][s]);
        }
      }
    }
  }
}

}  // namespace

// sdf_train_mfma.hip (round 6, the default): the same function layer by layer on the bf16 matrix pipe (bf16x3, fp32-equivalent);
// SURF_SDF_TRAIN_VALU=1 in the environment keeps the monolithic FMA kernel of this file (A/B switch, tests).
int surf_sdf_smooth_backward_layers(const float* pts, const float* sbar, int64_t n, const float* const* h_vols,
                                    const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                                    const float* packed, float* in, float* ab, hipStream_t stream);

// per-sample buffers (floats): in: 7 x 4 x n x 160; ab: 6 x 4 x n x 128 (stream order value | u | s | mixed)
extern "C" int surf_sdf_smooth_backward(const float* pts, const float* sbar, int64_t n, const float* const* h_vols,
                                        const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                                        const float* packed, float* in, float* ab, void* stream) {
  if (!pts || !sbar || !h_vols || !h_tables || !h_dims || !packed || !in || !ab) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  SmBwdArgs a;
  a.pts = pts; a.sbar = sbar; a.n = n; a.packed = packed; a.in = in; a.ab = ab;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    a.dvols[s] = (s < n_vol && h_dvols) ? h_dvols[s] : nullptr;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
  }
  if (!getenv("SURF_SDF_TRAIN_VALU"))
    return surf_sdf_smooth_backward_layers(pts, sbar, n, h_vols, h_tables, h_dims, n_vol, h_dvols, packed, in, ab, (hipStream_t)stream);
  const int64_t blocks = (n + S - 1) / S;
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  hipLaunchKernelGGL(sdf_smooth_bwd_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
