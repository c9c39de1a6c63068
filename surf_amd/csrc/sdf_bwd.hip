// K9b  backward of the SDF network for the training loss (second backward kernel of row f2 / K12).
// The loss sees, per sample, the SDF value y and its spatial gradient g = dy/dx (alpha compositing through the cosine term,
// the eikonal term, the normals): with upstream gradients ybar (scalar) and gbar (3-vector),
//     gbar . g  =  d/d eps  y(x + eps gbar)  =  ydot,     the forward-mode TANGENT of y along v = gbar,
// so d(ybar y + gbar . g)/d(weights, features) is ONE reverse sweep over the forward sweep that carries (value, tangent) -
// reverse over forward - instead of the reference's double backward (torch.autograd.grad(create_graph=True),
// sdf_network.py:129-141 under loss.backward(), runner.py:163):
//   forward   t_l = W_l in_l + b_l, t'_l = W_l in'_l;   h = sp(t), h' = sp'(t) t'
//   seeds     tbar_6[0] = ybar,  t'bar_6[0] = 1
//   reverse   inbar = W^T tbar, in'bar = W^T t'bar;   tbar_{l-1} = sp'(t) hbar + sp''(t) t' h'bar,   t'bar_{l-1} = sp'(t) h'bar
//   weights   dW_l = sum_n tbar_l (x) in_l + t'bar_l (x) in'_l,  db_l = sum_n tbar_l      (left to the caller as GEMMs over
//             the per-sample buffers this kernel writes: IN / IND (n,160) and TB / TDB (n,128) per layer)
//   features  dF[row_c] += w_c phibar + (grad w_c . v) phi'bar     (float atomics into the sparse volumes' gradient rows)
// Plain fp32 FMAs, one wavefront per 4 samples, same lane ownership as sdf_smooth.hip (whose packed weight image it reads);
// round 6: SURF_TRAIN_WAVES wavefronts per workgroup share the weight stream through LDS (sdf_train_common.h).
#include <stdlib.h>

#include "sdf_train_common.h"

// weight rows in flight per step of the k / neuron loops (the loops wait for one L2 round trip per unrolled group)
#ifndef SURF_TRAIN_UNROLL
#define SURF_TRAIN_UNROLL 4
#endif
#define SURF_STR2(x) #x
#define SURF_STR(x) SURF_STR2(x)
#define SURF_TRAIN_UNROLL_PRAGMA _Pragma(SURF_STR(unroll SURF_TRAIN_UNROLL))

namespace {

constexpr int S = 4, KP = 160, NH = 128, N_E = 27, N_PHI = 28, N_H2 = 101, N_HID = 6;
constexpr int OFF_WT = 0;
constexpr int OFF_W = OFF_WT + N_HID * KP * NH;
constexpr int OFF_B = OFF_W + N_HID * NH * KP;
constexpr int OFF_W6 = OFF_B + N_HID * NH;

__host__ __device__ constexpr int layer_k(int l) { return l == 0 ? N_E : 156; }
__host__ __device__ constexpr int layer_n(int l) { return l == 2 ? N_H2 : NH; }

constexpr int XS = 2 * S + 4;   // row stride of the transposed LDS operand arrays (floats)
#define XIN(q, s, k) xin_t[(k) * XS + (q) * S + (s)]
#define DL(q, s, k) dl_t[(k) * XS + (q) * S + (s)]
static_assert(S == 4, "one 16-byte LDS read per stream");

struct BwdArgs {
  const float* pts;
  const float* ybar;   // (n)
  const float* gbar;   // (n,3)
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  float* dvols[SURF_MAX_STAGES];   // gradient rows (N_s, 8), accumulated with atomics
  const float* packed;
  float* in_v;   // (7, n, KP)  layer inputs
  float* in_d;   // (7, n, KP)  their tangents along gbar
  float* tb;     // (6, n, NH)  adjoints of the pre-activations
  float* tdb;    // (6, n, NH)  adjoints of their tangents
};

struct Act { float h, s1, s2; };
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

__global__ __launch_bounds__(surf_train::NT) void sdf_bwd_kernel(BwdArgs a) {
  // Round 5: [k][value | tangent][sample] rows (8 values + 4 of padding): the k / neuron loops read a weight pair's eight
  // broadcast operands as two 16-byte LDS reads instead of eight 4-byte ones (see sdf_smooth_bwd.hip)
  __shared__ __attribute__((aligned(16))) float xin_all[surf_train::NW][KP * XS];
  __shared__ __attribute__((aligned(16))) float dl_all[surf_train::NW][NH * XS];
  __shared__ __attribute__((aligned(16))) float wbuf[surf_train::WBUF_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const xin_t = xin_all[wave];
  float* const dl_t = dl_all[wave];
  const int64_t base = ((int64_t)blockIdx.x * surf_train::NW + wave) * S;
  const float inv_sqrt2 = 0.70710678118654752440f;
  float e[S], je[S];
  float px[S], py[S], pz[S], vx[S], vy[S], vz[S], yb[S];
  bool live[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t i = base + s;
    live[s] = i < a.n;
    const int64_t ic = live[s] ? i : a.n - 1;
    px[s] = a.pts[ic * 3 + 0]; py[s] = a.pts[ic * 3 + 1]; pz[s] = a.pts[ic * 3 + 2];
    vx[s] = live[s] ? a.gbar[ic * 3 + 0] : 0.f; vy[s] = live[s] ? a.gbar[ic * 3 + 1] : 0.f; vz[s] = live[s] ? a.gbar[ic * 3 + 2] : 0.f;
    yb[s] = live[s] ? a.ybar[ic] : 0.f;
    {
      const int c = lane < N_E ? lane : 0;
      const int axis = c % 3, blk = c / 3;
      const float x = axis == 0 ? px[s] : (axis == 1 ? py[s] : pz[s]);
      if (blk == 0) {
        e[s] = x; je[s] = 1.0f;
      } else {
        const float f = (float)(1 << ((blk - 1) >> 1));
        float sn, cs;
        sincosf(x * f, &sn, &cs);
        if ((blk - 1) & 1) { e[s] = cs; je[s] = -f * sn; }
        else               { e[s] = sn; je[s] = f * cs; }
      }
      const float va = axis == 0 ? vx[s] : (axis == 1 ? vy[s] : vz[s]);
      je[s] *= va;                                  // tangent of the encoding channel along v
    }
    float phi = 0.f, phid = 0.f;
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
          const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
          phi += f * (wx * wy * wz);
          phid += f * ((sx * wy * wz / vs) * vx[s] + (sy * wx * wz / vs) * vy[s] + (sz * wx * wy / vs) * vz[s]);
        }
      }
      XIN(0, s, NH + lane) = phi;
      XIN(1, s, NH + lane) = phid;
    } else if (lane < KP - NH) {
      XIN(0, s, NH + lane) = 0.f;
      XIN(1, s, NH + lane) = 0.f;
    }
    XIN(0, s, lane) = 0.f; XIN(1, s, lane) = 0.f; XIN(0, s, lane + 64) = 0.f; XIN(1, s, lane + 64) = 0.f;   // columns 0..127
    if (lane < N_E) { XIN(0, s, lane) = e[s]; XIN(1, s, lane) = je[s]; }
  }
  __syncthreads();
  auto dump_inputs = [&](int l) {
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (live[s]) {
        const int64_t o = ((int64_t)l * a.n + base + s) * KP;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int k = lane + 64 * j;
          if (k < KP) { a.in_v[o + k] = XIN(0, s, k); a.in_d[o + k] = XIN(1, s, k); }
        }
      }
  };

  // ---- forward sweep with tangents -----------------------------------------------------------------------------------------
  // Round 5: sp'(t) and sp''(t) t' of every layer (96 values a lane) wait for the reverse sweep in TB / TDB themselves - row l of
  // each (n, 128) is their shape and is only written when the reverse sweep reaches layer l, by the same lane at the same
  // addresses - instead of in registers (sdf_smooth_bwd.hip has the same arrangement)
#pragma unroll
  for (int l = 0; l < N_HID; ++l) {
    dump_inputs(l);
    const float* __restrict__ wt = a.packed + OFF_WT + l * KP * NH;
    surf_train::V4 acc[2], accd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[j].zero(); accd[j].zero(); }
    const int K = layer_k(l);
    surf_train::stream_rows<NH>(wt, K, wbuf, [&](int k, const float* __restrict__ wr) {
      const float w0 = wr[lane], w1 = wr[64 + lane];
      const f32x4 xv4 = *reinterpret_cast<const f32x4*>(&xin_t[k * XS]), xd4 = *reinterpret_cast<const f32x4*>(&xin_t[k * XS + S]);
      acc[0].fma(w0, xv4);
      acc[1].fma(w1, xv4);
      accd[0].fma(w0, xd4);
      accd[1].fma(w1, xd4);
    });
    __syncthreads();
    const int N = layer_n(l);
    const float post = l == 2 ? inv_sqrt2 : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nrn = lane + 64 * j;
      const float b = a.packed[OFF_B + l * NH + nrn];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const Act t = softplus100(acc[j][s] + b);
        const bool real = nrn < N;
        if (live[s]) {
          const int64_t o = ((int64_t)l * a.n + base + s) * NH + nrn;
          a.tb[o] = real ? t.s1 : 0.f;
          a.tdb[o] = real ? t.s2 * accd[j][s] : 0.f;
        }
        XIN(0, s, nrn) = real ? t.h * post : 0.f;
        XIN(1, s, nrn) = real ? t.s1 * accd[j][s] * post : 0.f;
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
  dump_inputs(N_HID);   // inputs of lin6 (its row 0 alone reaches the loss)

  // ---- reverse sweep: adjoints of (pre-activation, its tangent) -------------------------------------------------------------
  float pbar[S], pdbar[S];   // adjoints of (phi, phi') for lane f < 28
#pragma unroll
  for (int s = 0; s < S; ++s) pbar[s] = pdbar[s] = 0.f;
#pragma unroll
  for (int l = N_HID; l >= 1; --l) {
    float c1[2][S], c2[2][S];   // sp' and sp'' t' of layer l - 1 for this lane's two neurons (dead samples: zeros)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int64_t o = ((int64_t)(l - 1) * a.n + base + (live[s] ? s : 0)) * NH + lane + 64 * j;
        c1[j][s] = live[s] ? a.tb[o] : 0.f;
        c2[j][s] = live[s] ? a.tdb[o] : 0.f;
      }
    surf_train::V4 g[3], gd[3];
    if (l == N_HID) {   // tbar_6 = ybar e_0, t'bar_6 = e_0
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k = lane + 64 * j;
        const float w = k < KP ? a.packed[OFF_W6 + k] : 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) { g[j].set(s, w * yb[s]); gd[j].set(s, live[s] ? w : 0.f); }
      }
    } else {
      const float* __restrict__ w = a.packed + OFF_W + l * NH * KP;
#pragma unroll
      for (int j = 0; j < 3; ++j) { g[j].zero(); gd[j].zero(); }
      const int N = layer_n(l);
      const bool third = lane < KP - 128;
      surf_train::stream_rows<KP>(w, N, wbuf, [&](int nrn, const float* __restrict__ wr) {
        const float w0 = wr[lane], w1 = wr[64 + lane];
        const float w2 = third ? wr[128 + lane] : 0.f;
        const f32x4 dv4 = *reinterpret_cast<const f32x4*>(&dl_t[nrn * XS]), dd4 = *reinterpret_cast<const f32x4*>(&dl_t[nrn * XS + S]);
        g[0].fma(w0, dv4);
        g[1].fma(w1, dv4);
        g[2].fma(w2, dv4);
        gd[0].fma(w0, dd4);
        gd[1].fma(w1, dd4);
        gd[2].fma(w2, dd4);
      });
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < S; ++s) { pbar[s] += g[2][s]; pdbar[s] += gd[2][s]; }
    const float pre = l == 3 ? inv_sqrt2 : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float hb = g[j][s] * pre, hdb = gd[j][s] * pre;       // adjoints of (h, h') of layer l-1 (zero weight beyond its width)
        const float tbv = fmaf(c2[j][s], hdb, c1[j][s] * hb);
        const float tdbv = c1[j][s] * hdb;
        DL(0, s, k) = tbv;
        DL(1, s, k) = tdbv;
        if (live[s]) {
          const int64_t o = ((int64_t)(l - 1) * a.n + base + s) * NH + k;
          a.tb[o] = tbv;
          a.tdb[o] = tdbv;
        }
      }
    }
    __syncthreads();
  }

  // ---- feature gradients: dF[row_c] += w_c phibar + (grad w_c . v) phi'bar ---------------------------------------------------
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
          const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
          const float wv = (sx * wy * wz / vs) * vx[s] + (sy * wx * wz / vs) * vy[s] + (sz * wx * wy / vs) * vz[s];
          atomicAdd(dvol + (int64_t)row * 8 + ch, (wx * wy * wz) * pbar[s] + wv * pdbar[s]);
        }
      }
    }
  }
}

}  // namespace

// sdf_train_mfma.hip (round 6, the default): the same function layer by layer on the bf16 matrix pipe (bf16x3, fp32-equivalent);
// SURF_SDF_TRAIN_VALU=1 in the environment keeps the monolithic FMA kernel of this file (A/B switch, tests).
int surf_sdf_backward_layers(const float* pts, const float* ybar, const float* gbar, int64_t n, const float* const* h_vols,
                             const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                             const float* packed, float* in_v, float* in_d, float* tb, float* tdb, hipStream_t stream);

// per-sample buffers (floats): in_v / in_d: 7 n 160 each; tb / tdb: 6 n 128 each
extern "C" int surf_sdf_backward(const float* pts, const float* ybar, const float* gbar, int64_t n, const float* const* h_vols,
                                 const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                                 const float* packed, float* in_v, float* in_d, float* tb, float* tdb, void* stream) {
  if (!pts || !ybar || !gbar || !h_vols || !h_tables || !h_dims || !packed || !in_v || !in_d || !tb || !tdb) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  BwdArgs a;
  a.pts = pts; a.ybar = ybar; a.gbar = gbar; a.n = n; a.packed = packed; a.in_v = in_v; a.in_d = in_d; a.tb = tb; a.tdb = tdb;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    a.dvols[s] = (s < n_vol && h_dvols) ? h_dvols[s] : nullptr;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
  }
  if (!getenv("SURF_SDF_TRAIN_VALU"))
    return surf_sdf_backward_layers(pts, ybar, gbar, n, h_vols, h_tables, h_dims, n_vol, h_dvols, packed, in_v, in_d, tb, tdb,
                                    (hipStream_t)stream);
  const int64_t blocks = (n + S * surf_train::NW - 1) / (S * surf_train::NW);
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  hipLaunchKernelGGL(sdf_bwd_kernel, dim3((unsigned)blocks), dim3(surf_train::NT), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
