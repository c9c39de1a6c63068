// K9t (round 6): the training kernels of the SDF network, layer by layer on the bf16 matrix pipe.
//
// surf_sdf_backward / surf_sdf_smooth_backward keep their signatures and their per-sample buffers (sdf_bwd.hip,
// sdf_smooth_bwd.hip: IN / IND (7, n, 160), TB / TDB (6, n, 128); IN (7, 4, n, 160), AB (6, 4, n, 128)) - the weight-gradient
// GEMMs of the caller (surf_colgram) read exactly those - but the sweeps are no longer one monolithic kernel that gives a
// wavefront 4 samples and multiplies with LDS-broadcast fp32 FMAs (34 TFLOP/s: bound by the LDS return path, round 6
// measurements in sdf_train_common.h).  The per-sample buffers ARE the layer inputs and the adjoints, so every layer is one
// streaming GEMM launch over them:
//   forward layer l    T_q (n x 128)  = IN_q[l] (n x 160) . W_l^T           for the NS streams q, then the softplus algebra in
//                                                                            registers -> IN_q[l + 1], parked coefficients
//   reverse layer l    G_q (n x 160)  = ADJ_q[l] (n x 128) . W_l           -> adjoint algebra with the parked coefficients
//                                                                            -> ADJ_q[l - 1]; the feature columns 128..155 are
//                                                                            accumulated per sample
// on v_mfma_f32_32x32x16_bf16 with both operands split exactly into three bf16 pieces (six products, fp32 accumulate:
// fp32-equivalent, like the inference kernels).  WEIGHTS STATIONARY IN REGISTERS: a workgroup takes 32 samples at a time and its
// wavefront t owns row tile t (32 neurons) of the layer for all NS streams - it splits its A fragments ONCE (all k-steps: up to
// 10 x 3 sixteen-byte registers a lane, from the fp32 weight images of sdf_smooth.hip: k-major for the forward, neuron-major for
// the reverse) and keeps them over every sample tile the workgroup walks; B = the sample's row segment (two 16-byte loads per
// stream and k-step - the four wavefronts read the same rows, three of them out of L1 / L2), split in registers; no LDS, no
// barrier.  (The first form staged a k-step's row tiles into LDS per k-step and barrier: a chain of L2 round trips, 108 us per
// layer launch with every ingredient but the chain ablated at 65.)  The accumulator tile leaves lane (sample c, half h) with
// neurons 32 t + 8 g + 4 h + i, i = 0..3: 16-byte loads / stores of the parked rows.
// What bounds it: HBM traffic.  A layer launch reads and writes its streams' rows once (the monolithic kernels wrote the same rows
// and never re-read them): sdf_backward 2.3 GB, sdf_smooth_backward 4.8 GB per 65,536 samples, at the 2.3 - 2.5 TB/s these
// row-per-lane streams reach.  Measured at 65,536 samples (scripts/time_sdf_train.py, same box): surf_sdf_backward 1.59 - 1.75 ms
// (monolithic FMA kernel) -> 1.20 ms (setup 0.09 + 6 forward launches 0.06 - 0.09 + 6 reverse 0.05 - 0.09 + scatter 0.12);
// surf_sdf_smooth_backward 3.43 -> 2.48 ms.  Forms measured on the way (profiles/r06_train_experiments.txt): A staged per k-step
// through LDS with a barrier each (108 us per layer launch: a chain of L2 round trips); every wavefront loading its own sample
// rows (133 us: the four row-tile wavefronts repeat the 32-byte-per-lane row reads and the split).  surf_sdf_smooth (forward only,
// no per-sample buffers to stream) stays on its FMA kernel.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int KP = 160, NH = 128, N_E = 27, N_H2 = 101, N_HID = 6;
constexpr int OFF_WT = 0;
constexpr int OFF_W = OFF_WT + N_HID * KP * NH;
constexpr int OFF_B = OFF_W + N_HID * NH * KP;
constexpr int OFF_W6 = OFF_B + N_HID * NH;
constexpr int ACC_COL = 32;      // columns 32..63 of a layer-0 input row (27 used, 32 multiplied) hold the feature adjoints

__host__ __device__ constexpr int layer_k(int l) { return l == 0 ? N_E : 156; }
__host__ __device__ constexpr int layer_n(int l) { return l == 2 ? N_H2 : NH; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Act { float h, s1, s2, s3; };
// nn.Softplus(beta=100) and its first three derivatives; the linear branch (100 t > 20) has s1 = 1, s2 = s3 = 0
// Raw v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp) as in the inference kernels (sdf_mlp.hip): 1 + e >= 1 is never denormal, an e that
// underflows to 0 gives h = 0, s1 = s2 = s3 = 0, the fp32 limits; absolute error of h <= 2e-8, of s1 <= 3e-7.
__device__ __forceinline__ Act softplus100(float t) {
  const float bt = t * 100.0f;
  const float e = __builtin_amdgcn_exp2f(fminf(bt, 20.0f) * 1.44269504088896341f);
  const float d = 1.0f + e, rd = __builtin_amdgcn_rcpf(d);
  const bool lin = bt > 20.0f;
  Act a;
  a.h = lin ? t : __builtin_amdgcn_logf(d) * (0.69314718055994531f * 0.01f);
  a.s1 = lin ? 1.0f : e * rd;
  a.s2 = lin ? 0.0f : 100.0f * a.s1 * rd;
  a.s3 = lin ? 0.0f : 100.0f * a.s2 * (1.0f - 2.0f * a.s1);
  return a;
}

#ifndef SURF_TM_GRID
#define SURF_TM_GRID 512          // workgroups per layer launch (two per CU), each walking its share of the 32-sample tiles
#endif
#ifndef SURF_X_TM
#define SURF_X_TM 0      // timing-only ablations (wrong results): 1 no softplus, 2 no epilogue stores, 4 no weight split, 8 no row loads
#endif
#define SURF_TM_MFMA(x, y, c) \
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)

// ---- weights-stationary GEMM pieces ----------------------------------------------------------------------------------------------
// A element (row = 32 t + r, k) = wsrc[k * ROW + 32 t + r]  (ROW = NH: forward, k = input index; ROW = KP: reverse, k = neuron)
template <int KS, int ROW>
__device__ __forceinline__ void load_a(u32x4 (&areg)[KS][3], const float* __restrict__ wsrc, int t) {
  const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float* __restrict__ ws = wsrc + (int64_t)(16 * ks + 8 * h) * ROW + 32 * t + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t q[3];
      if (SURF_X_TM & 4) { q[0] = q[1] = q[2] = __builtin_bit_cast(uint32_t, ws[(2 * j) * ROW]); }
      else surf_split3_bf16(ws[(2 * j) * ROW], ws[(2 * j + 1) * ROW], q);
      areg[ks][0][j] = q[0]; areg[ks][1][j] = q[1]; areg[ks][2][j] = q[2];
    }
  }
}

// B operand of a 32-sample tile, shared by the row-tile wavefronts of the workgroup: the NS streams' row segments are read ONCE,
// coalesced (4 consecutive threads = one 128-byte line of a sample's row), split into three bf16 pieces and parked in LDS in
// B-fragment order: blds[((q * KS + ks) * 3 + piece) * 64 + 32 h + c] = the 8 k-slots (16 bytes) of lane (sample c, half h).
// xbase[q] + sample * XROW = the row of `sample` in stream q.  Caller: barrier before (the previous tile's reads) and after.
template <int NS, int KS, int XROW, int NT>
__device__ __forceinline__ void stage_b(u32x4* __restrict__ blds, const float* const (&xbase)[NS], int64_t s0, int64_t n) {
  constexpr int U = 2 * KS;                                   // 8-float units per row
  for (int idx = threadIdx.x; idx < NS * ((U + 3) / 4) * 128; idx += 64 * NT) {
    const int u4 = idx & 3, cc = (idx >> 2) & 31, rest = idx >> 7;      // rest = (q, u / 4)
    const int u = 4 * (rest % ((U + 3) / 4)) + u4, q = rest / ((U + 3) / 4);
    if (u >= U) continue;
    int64_t smp = s0 + cc;
    if (smp >= n) smp = n - 1;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(xbase[q] + smp * XROW + 8 * u);
    const f32x4 v0 = src[0], v1 = src[1];
    u32x4 p[3];
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t pc[3];
      surf_split3_bf16(v[2 * j], v[2 * j + 1], pc);
      p[0][j] = pc[0]; p[1][j] = pc[1]; p[2][j] = pc[2];
    }
    const int ks = u >> 1, h = u & 1;
#pragma unroll
    for (int e = 0; e < 3; ++e) blds[((q * KS + ks) * 3 + e) * 64 + 32 * h + cc] = p[e];
  }
}

// acc[q] (one 32-row tile, NS streams) += A (registers) . B_q (LDS fragments)
template <int NS, int KS>
__device__ __forceinline__ void gemm_lds(f32x16 (&acc)[NS], const u32x4 (&areg)[KS][3], const u32x4* __restrict__ blds) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const u32x4* __restrict__ bp = blds + ((q * KS + ks) * 3) * 64 + lane;
      const u32x4 b0 = bp[0], b1 = bp[64], b2 = bp[128];
      SURF_TM_MFMA(areg[ks][2], b0, acc[q]);  // smallest terms first
      SURF_TM_MFMA(areg[ks][0], b2, acc[q]);
      SURF_TM_MFMA(areg[ks][1], b1, acc[q]);
      SURF_TM_MFMA(areg[ks][1], b0, acc[q]);
      SURF_TM_MFMA(areg[ks][0], b1, acc[q]);
      SURF_TM_MFMA(areg[ks][0], b0, acc[q]);
    }
}

// =================================================================================================================================
// surf_sdf_backward: streams (value, tangent along gbar); derivation at the head of sdf_bwd.hip
// =================================================================================================================================
struct BwdArgs {
  const float* pts;
  const float* ybar;
  const float* gbar;
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  float* dvols[SURF_MAX_STAGES];
  const float* packed;
  float* in_v;   // (7, n, KP)
  float* in_d;   // (7, n, KP)
  float* tb;     // (6, n, NH)
  float* tdb;    // (6, n, NH)
};

// inputs.  Phase 1: thread = (sample, part): part p < 4 owns the positional-encoding frequency 2^p and the sparse-volume stage p,
// results into LDS.  Phase 2: the workgroup writes the rows of its 64 samples with 16-byte stores: layer 0 columns 0..63
// (encoding | zero k-slots | feature-adjoint accumulators), layer 3 columns 100..127 are the skip connection's encoding block
// (column 100 belongs to layer 2's output: written element-wise), columns 128..159 of layers 1..6 the features (+ zero padding).
__global__ __launch_bounds__(256) void bwd_setup_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float enc[2][64][32];    // [stream][sample][channel 0..26 | 0]
  __shared__ __attribute__((aligned(16))) float fea[2][64][32];    // [stream][sample][feature 0..27 | 0]
  const int sl = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const int64_t s = s0 + sl;
  const float inv_sqrt2 = 0.70710678118654752440f;
  {
    const int64_t sc = s < a.n ? s : a.n - 1;
    const float p3[3] = {a.pts[sc * 3 + 0], a.pts[sc * 3 + 1], a.pts[sc * 3 + 2]};
    const float v3[3] = {a.gbar[sc * 3 + 0], a.gbar[sc * 3 + 1], a.gbar[sc * 3 + 2]};
    const float f = (float)(1 << part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {                              // embedder.py:11-36
      float sn, cs;
      sincosf(p3[ax] * f, &sn, &cs);
      const int c_s = 3 * (1 + 2 * part) + ax, c_c = 3 * (2 + 2 * part) + ax;
      enc[0][sl][c_s] = sn; enc[1][sl][c_s] = f * cs * v3[ax];
      enc[0][sl][c_c] = cs; enc[1][sl][c_c] = -f * sn * v3[ax];
      if (part == 0) { enc[0][sl][ax] = p3[ax]; enc[1][sl][ax] = v3[ax]; }
    }
    if (part == 3) {
#pragma unroll
      for (int c = N_E; c < 32; ++c) { enc[0][sl][c] = 0.f; enc[1][sl][c] = 0.f; fea[0][sl][c + 1] = 0.f; fea[1][sl][c + 1] = 0.f; }
    }
    // sparse trilinear gather of stage `part` (projector.py:217-390): value and tangent along v
    float phi[7], phid[7];
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) phi[ch] = phid[ch] = 0.f;
    const int D = a.dims[part];
    if (D > 1) {
      const int32_t* __restrict__ table = a.tables[part];
      const float* __restrict__ vol = a.vols[part];
      const float vs = 2.0f / ((float)D - 1.0f);
      const float gx = (p3[0] + 1.0f) / vs, gy = (p3[1] + 1.0f) / vs, gz = (p3[2] + 1.0f) / vs;
      const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
      const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
      const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
        const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
        const int row = table[((int64_t)xi * D + yi) * D + zi];
        if (row < 0) continue;
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8), f1 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8 + 4);
        const float fv[7] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2]};
        const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
        const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
        const float w0 = wx * wy * wz;
        const float wv = (sx * wy * wz / vs) * v3[0] + (sy * wx * wz / vs) * v3[1] + (sz * wx * wy / vs) * v3[2];
#pragma unroll
        for (int ch = 0; ch < 7; ++ch) {
          phi[ch] += fv[ch] * w0;
          phid[ch] += fv[ch] * wv;
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) { fea[0][sl][7 * part + ch] = phi[ch]; fea[1][sl][7 * part + ch] = phid[ch]; }
  }
  __syncthreads();
  // phase 2: items (sample, stream, job, 16-byte vector); jobs 0..5: features of layers 1..6, job 6: layer 0 columns 0..63,
  // job 7: layer 3 columns 100..127 (the encoding block of the skip connection)
  for (int it = threadIdx.x; it < 64 * 2 * 8 * 16; it += 256) {
    const int vec = it & 15, job = (it >> 4) & 7, q = (it >> 7) & 1, sm = it >> 8;
    const int64_t smp = s0 + sm;
    if (smp >= a.n) continue;
    float* __restrict__ base = q ? a.in_d : a.in_v;
    if (job < 6) {
      if (vec >= 8) continue;
      *reinterpret_cast<f32x4*>(base + ((int64_t)(job + 1) * a.n + smp) * KP + NH + 4 * vec) = *reinterpret_cast<const f32x4*>(&fea[q][sm][4 * vec]);
    } else if (job == 6) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec < 8) v = *reinterpret_cast<const f32x4*>(&enc[q][sm][4 * vec]);
      *reinterpret_cast<f32x4*>(base + smp * KP + 4 * vec) = v;            // columns 0..31 encoding (27 + zeros), 32..63 zeros
    } else {
      if (vec >= 7) continue;
      float* __restrict__ dst = base + ((int64_t)3 * a.n + smp) * KP + 100 + 4 * vec;     // columns 100..127: channel c at 101 + c
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = 4 * vec + i - 1;
        if (ch >= 0) dst[i] = enc[q][sm][ch] * inv_sqrt2;
      }
    }
  }
}

// forward layer l: (value, tangent) rows of layer l -> rows of layer l + 1, sp' in TB[l], sp'' t' in TDB[l].
// Workgroup = 4 wavefronts = the 4 row tiles of the layer, walking 32-sample tiles.
template <int KS>
__global__ __launch_bounds__(256) void bwd_forward_kernel(BwdArgs a, int l) {
  __shared__ u32x4 blds[2 * KS * 3 * 64];                      // the tile's B fragments: 60 KB at KS = 10
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) / 32;
  const int N = layer_n(l);
  const float post = l == 2 ? 0.70710678118654752440f : 1.0f;  // lin3's input is cat([h2, e]) / sqrt(2)
  u32x4 areg[KS][3];
  load_a<KS, NH>(areg, a.packed + OFF_WT + l * KP * NH, t);
  const float* const xbase[2] = {a.in_v + (int64_t)l * a.n * KP, a.in_d + (int64_t)l * a.n * KP};
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t smp = tile * 32 + c;
    const bool live = smp < a.n;
    __syncthreads();                                           // the previous tile's fragments are consumed
    stage_b<2, KS, KP, 4>(blds, xbase, tile * 32, a.n);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    gemm_lds<2, KS>(acc, areg, blds);
    if (!live) continue;
    float* __restrict__ nv = a.in_v + ((int64_t)(l + 1) * a.n + smp) * KP;
    float* __restrict__ nd = a.in_d + ((int64_t)(l + 1) * a.n + smp) * KP;
    float* __restrict__ ptb = a.tb + ((int64_t)l * a.n + smp) * NH;
    float* __restrict__ ptd = a.tdb + ((int64_t)l * a.n + smp) * NH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = 32 * t + 8 * g + 4 * h;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.packed + OFF_B + l * NH + n0);
      f32x4 ov, od, o1, o2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Act A;
        if (SURF_X_TM & 1) { const float tt = acc[0][4 * g + i] + bias[i]; A.h = fmaxf(tt, 0.f); A.s1 = tt > 0.f ? 1.f : 0.f; A.s2 = tt * 0.5f; A.s3 = 0.f; }
        else A = softplus100(acc[0][4 * g + i] + bias[i]);
        const float td = acc[1][4 * g + i];
        const bool real = n0 + i < N;
        ov[i] = real ? A.h * post : 0.f;
        od[i] = real ? A.s1 * td * post : 0.f;
        o1[i] = real ? A.s1 : 0.f;                               // parked for the reverse sweep
        o2[i] = real ? A.s2 * td : 0.f;
      }
      if ((SURF_X_TM & 2) && o1[0] + o2[1] + ov[2] + od[3] != 12345.f) continue;
      *reinterpret_cast<f32x4*>(ptb + n0) = o1;
      *reinterpret_cast<f32x4*>(ptd + n0) = o2;
      if (n0 + 3 < N) {
        *reinterpret_cast<f32x4*>(nv + n0) = ov;
        *reinterpret_cast<f32x4*>(nd + n0) = od;
      } else {                                                   // layer 2: slots 101..127 hold the skip connection's encoding
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n0 + i < N) { nv[n0 + i] = ov[i]; nd[n0 + i] = od[i]; }
      }
    }
  }
}

// reverse layer l (N_HID .. 1): adjoints (tbar_l, t'bar_l) -> (tbar_{l-1}, t'bar_{l-1}); feature adjoints accumulated per sample.
// Workgroup = 5 wavefronts = the 5 row tiles of W_l^T (input indices 0..159; tile 4 = the feature columns).
// KS == 0 (l == N_HID): tbar_6 = ybar e_0, t'bar_6 = e_0: the adjoints of lin6's inputs are its row 0 (x ybar), no GEMM.
template <int KS>
__global__ __launch_bounds__(320) void bwd_reverse_kernel(BwdArgs a, int l) {
  __shared__ u32x4 blds[2 * (KS > 0 ? KS : 1) * 3 * 64];       // 48 KB at KS = 8
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) / 32;
  const float pre = l == 3 ? 0.70710678118654752440f : 1.0f;
  u32x4 areg[KS > 0 ? KS : 1][3];
  if constexpr (KS > 0) load_a<KS, KP>(areg, a.packed + OFF_W + l * NH * KP, t);
  const float* const xbase[2] = {a.tb + (int64_t)l * a.n * NH, a.tdb + (int64_t)l * a.n * NH};
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t smp = tile * 32 + c;
    const bool live = smp < a.n;
    const int64_t sc = live ? smp : a.n - 1;
    f32x16 acc[2];
    if constexpr (KS > 0) {
      __syncthreads();
      stage_b<2, KS, NH, 5>(blds, xbase, tile * 32, a.n);
      __syncthreads();
    }
    if constexpr (KS == 0) {
      const float yb = a.ybar[sc];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w = a.packed[OFF_W6 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        acc[0][r] = w * yb;
        acc[1][r] = w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      gemm_lds<2, KS>(acc, areg, blds);
    }
    if (!live) continue;
    if (t < 4) {
      float* __restrict__ ptb = a.tb + ((int64_t)(l - 1) * a.n + smp) * NH;
      float* __restrict__ ptd = a.tdb + ((int64_t)(l - 1) * a.n + smp) * NH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = 32 * t + 8 * g + 4 * h;
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(ptb + k0), c2 = *reinterpret_cast<const f32x4*>(ptd + k0);
        f32x4 o1, o2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float hb = acc[0][4 * g + i] * pre, hdb = acc[1][4 * g + i] * pre;   // adjoints of (h, h') of layer l - 1
          o1[i] = fmaf(c2[i], hdb, c1[i] * hb);
          o2[i] = c1[i] * hdb;
        }
        *reinterpret_cast<f32x4*>(ptb + k0) = o1;
        *reinterpret_cast<f32x4*>(ptd + k0) = o2;
      }
    } else {        // feature columns 128..155: accumulated in the free columns of the layer-0 rows
      float* __restrict__ pv = a.in_v + smp * KP + ACC_COL;
      float* __restrict__ pd = a.in_d + smp * KP + ACC_COL;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 8 * g + 4 * h;
        f32x4 sv = *reinterpret_cast<const f32x4*>(pv + f0), sd = *reinterpret_cast<const f32x4*>(pd + f0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sv[i] += acc[0][4 * g + i]; sd[i] += acc[1][4 * g + i]; }
        *reinterpret_cast<f32x4*>(pv + f0) = sv;
        *reinterpret_cast<f32x4*>(pd + f0) = sd;
      }
    }
  }
}

// feature gradients: dF[row_c] += w_c phibar + (grad w_c . v) phi'bar.  thread = (sample, stage, corner, channel slot of 8): the
// 8 lanes of a corner add into one 32-byte row - an add instruction covers 8 rows instead of 64 (float atomics cost per distinct
// segment at the memory side; the first mapping, one lane per (sample, stage, corner pair), took 400 us)
__global__ __launch_bounds__(256) void bwd_scatter_kernel(BwdArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t s = t >> 8;
  const int st = (int)((t >> 6) & 3), k = (int)((t >> 3) & 7), ch = (int)(t & 7);
  if (s >= a.n || ch >= 7) return;
  const int D = a.dims[st];
  if (D <= 1 || !a.dvols[st]) return;
  const float px = a.pts[s * 3 + 0], py = a.pts[s * 3 + 1], pz = a.pts[s * 3 + 2];
  const float vx = a.gbar[s * 3 + 0], vy = a.gbar[s * 3 + 1], vz = a.gbar[s * 3 + 2];
  const float pb = a.in_v[s * KP + ACC_COL + 7 * st + ch], pdb = a.in_d[s * KP + ACC_COL + 7 * st + ch];
  const float vs = 2.0f / ((float)D - 1.0f);
  const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
  const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
  const int row = a.tables[st][((int64_t)xi * D + yi) * D + zi];
  if (row < 0) return;
  const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
  const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
  const float w0 = wx * wy * wz;
  const float wv = (sx * wy * wz / vs) * vx + (sy * wx * wz / vs) * vy + (sz * wx * wy / vs) * vz;
  atomicAdd(a.dvols[st] + (int64_t)row * 8 + ch, w0 * pb + wv * pdb);
}

// =================================================================================================================================
// surf_sdf_smooth_backward: streams (value, tangent along u = (1,1,1), tangent along sbar, mixed); derivation at the head of
// sdf_smooth_bwd.hip.  IN (7, 4, n, 160), AB (6, 4, n, 128), stream-major.  The B fragments of a tile are staged two streams at a time
// (60 KB of LDS: two workgroups per CU).
// =================================================================================================================================
struct SmArgs {
  const float* pts;
  const float* sbar;
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  float* dvols[SURF_MAX_STAGES];
  const float* packed;
  float* in;     // (7, 4, n, KP)
  float* ab;     // (6, 4, n, NH)
};

__global__ __launch_bounds__(256) void sm_setup_kernel(SmArgs a) {
  __shared__ __attribute__((aligned(16))) float enc[4][64][32];    // [stream][sample][channel 0..26 | 0]
  __shared__ __attribute__((aligned(16))) float fea[4][64][32];    // [stream][sample][feature 0..27 | 0]
  const int sl = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const float inv_sqrt2 = 0.70710678118654752440f;
  {
    const int64_t s = s0 + sl;
    const int64_t sc = s < a.n ? s : a.n - 1;
    const float p3[3] = {a.pts[sc * 3 + 0], a.pts[sc * 3 + 1], a.pts[sc * 3 + 2]};
    const float v3[3] = {a.sbar[sc * 3 + 0], a.sbar[sc * 3 + 1], a.sbar[sc * 3 + 2]};
    const float f = (float)(1 << part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {     // channel, d/dx, d2/dx2 (each channel depends on one coordinate); u = (1,1,1)
      float sn, cs;
      sincosf(p3[ax] * f, &sn, &cs);
      const int c_s = 3 * (1 + 2 * part) + ax, c_c = 3 * (2 + 2 * part) + ax;
      enc[0][sl][c_s] = sn; enc[1][sl][c_s] = f * cs; enc[2][sl][c_s] = f * cs * v3[ax]; enc[3][sl][c_s] = -f * f * sn * v3[ax];
      enc[0][sl][c_c] = cs; enc[1][sl][c_c] = -f * sn; enc[2][sl][c_c] = -f * sn * v3[ax]; enc[3][sl][c_c] = -f * f * cs * v3[ax];
      if (part == 0) { enc[0][sl][ax] = p3[ax]; enc[1][sl][ax] = 1.0f; enc[2][sl][ax] = v3[ax]; enc[3][sl][ax] = 0.f; }
    }
    if (part == 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = N_E; c < 32; ++c) { enc[q][sl][c] = 0.f; fea[q][sl][c + 1] = 0.f; }
    }
    float phi[4][7];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) phi[q][ch] = 0.f;
    const int D = a.dims[part];
    if (D > 1) {
      const int32_t* __restrict__ table = a.tables[part];
      const float* __restrict__ vol = a.vols[part];
      const float vs = 2.0f / ((float)D - 1.0f);
      const float gx = (p3[0] + 1.0f) / vs, gy = (p3[1] + 1.0f) / vs, gz = (p3[2] + 1.0f) / vs;
      const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
      const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
      const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
        const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
        const int row = table[((int64_t)xi * D + yi) * D + zi];
        if (row < 0) continue;
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8), f1 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8 + 4);
        const float fv[7] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2]};
        const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
        const float sx = (dx ? 1.0f : -1.0f) / vs, sy = (dy ? 1.0f : -1.0f) / vs, sz = (dz ? 1.0f : -1.0f) / vs;
        const float gwx = sx * wy * wz, gwy = sy * wx * wz, gwz = sz * wx * wy;          // grad w_c
        const float hxy = sx * sy * wz, hxz = sx * sz * wy, hyz = sy * sz * wx;          // mixed second derivatives
        const float w4[4] = {wx * wy * wz, gwx + gwy + gwz, gwx * v3[0] + gwy * v3[1] + gwz * v3[2],
                             hxy * (v3[0] + v3[1]) + hxz * (v3[0] + v3[2]) + hyz * (v3[1] + v3[2])};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int ch = 0; ch < 7; ++ch) phi[q][ch] += fv[ch] * w4[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) fea[q][sl][7 * part + ch] = phi[q][ch];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 64 * 4 * 8 * 16; it += 256) {       // as bwd_setup_kernel, four streams
    const int vec = it & 15, job = (it >> 4) & 7, q = (it >> 7) & 3, sm = it >> 9;
    const int64_t smp = s0 + sm;
    if (smp >= a.n) continue;
    if (job < 6) {
      if (vec >= 8) continue;
      *reinterpret_cast<f32x4*>(a.in + (((int64_t)(job + 1) * 4 + q) * a.n + smp) * KP + NH + 4 * vec) = *reinterpret_cast<const f32x4*>(&fea[q][sm][4 * vec]);
    } else if (job == 6) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec < 8) v = *reinterpret_cast<const f32x4*>(&enc[q][sm][4 * vec]);
      *reinterpret_cast<f32x4*>(a.in + ((int64_t)q * a.n + smp) * KP + 4 * vec) = v;
    } else {
      if (vec >= 7) continue;
      float* __restrict__ dst = a.in + (((int64_t)3 * 4 + q) * a.n + smp) * KP + 100 + 4 * vec;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = 4 * vec + i - 1;
        if (ch >= 0) dst[i] = enc[q][sm][ch] * inv_sqrt2;
      }
    }
  }
}

template <int KS>
__global__ __launch_bounds__(256) void sm_forward_kernel(SmArgs a, int l) {
  __shared__ u32x4 blds[2 * KS * 3 * 64];
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) / 32;
  const int N = layer_n(l);
  const float post = l == 2 ? 0.70710678118654752440f : 1.0f;
  const int64_t qs_in = a.n * KP, qs_ab = a.n * NH;
  u32x4 areg[KS][3];
  load_a<KS, NH>(areg, a.packed + OFF_WT + l * KP * NH, t);
  const float* const x01[2] = {a.in + (int64_t)l * 4 * qs_in, a.in + ((int64_t)l * 4 + 1) * qs_in};
  const float* const x23[2] = {a.in + ((int64_t)l * 4 + 2) * qs_in, a.in + ((int64_t)l * 4 + 3) * qs_in};
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t smp = tile * 32 + c;
    const bool live = smp < a.n;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __syncthreads();
    stage_b<2, KS, KP, 4>(blds, x01, tile * 32, a.n);
    __syncthreads();
    gemm_lds<2, KS>(reinterpret_cast<f32x16(&)[2]>(acc[0]), areg, blds);
    __syncthreads();
    stage_b<2, KS, KP, 4>(blds, x23, tile * 32, a.n);
    __syncthreads();
    gemm_lds<2, KS>(reinterpret_cast<f32x16(&)[2]>(acc[2]), areg, blds);
    if (!live) continue;
    float* __restrict__ nx = a.in + ((int64_t)(l + 1) * 4 * a.n + smp) * KP;      // + q qs_in
    float* __restrict__ pab = a.ab + ((int64_t)l * 4 * a.n + smp) * NH;           // + q qs_ab
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = 32 * t + 8 * g + 4 * h;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(a.packed + OFF_B + l * NH + n0);
      f32x4 ox[4], ok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const Act A = softplus100(acc[0][4 * g + i] + bias[i]);
        const float au = acc[1][4 * g + i], as = acc[2][4 * g + i], am = acc[3][4 * g + i];
        const bool real = n0 + i < N;
        ok[0][i] = real ? A.s1 : 0.f;                                      // parked: c1 | cu | cs | cm
        ok[1][i] = real ? A.s2 * au : 0.f;
        ok[2][i] = real ? A.s2 * as : 0.f;
        ok[3][i] = real ? fmaf(A.s3 * au, as, A.s2 * am) : 0.f;
        ox[0][i] = real ? A.h * post : 0.f;
        ox[1][i] = real ? A.s1 * au * post : 0.f;
        ox[2][i] = real ? A.s1 * as * post : 0.f;
        ox[3][i] = real ? fmaf(A.s2 * au, as, A.s1 * am) * post : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4*>(pab + q * qs_ab + n0) = ok[q];
        if (n0 + 3 < N) {
          *reinterpret_cast<f32x4*>(nx + q * qs_in + n0) = ox[q];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (n0 + i < N) nx[q * qs_in + n0 + i] = ox[q][i];
        }
      }
    }
  }
}

template <int KS>
__global__ __launch_bounds__(320) void sm_reverse_kernel(SmArgs a, int l) {
  __shared__ u32x4 blds[2 * (KS > 0 ? KS : 1) * 3 * 64];
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int64_t n_tiles = (a.n + 31) / 32;
  const float pre = l == 3 ? 0.70710678118654752440f : 1.0f;
  const int64_t qs_in = a.n * KP, qs_ab = a.n * NH;
  u32x4 areg[KS > 0 ? KS : 1][3];
  if constexpr (KS > 0) load_a<KS, KP>(areg, a.packed + OFF_W + l * NH * KP, t);
  const float* const x01[2] = {a.ab + (int64_t)l * 4 * qs_ab, a.ab + ((int64_t)l * 4 + 1) * qs_ab};
  const float* const x23[2] = {a.ab + ((int64_t)l * 4 + 2) * qs_ab, a.ab + ((int64_t)l * 4 + 3) * qs_ab};
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t smp = tile * 32 + c;
    const bool live = smp < a.n;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if constexpr (KS == 0) {          // S = w6 . x_m: the adjoint of lin6's mixed input is lin6 row 0
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[3][r] = a.packed[OFF_W6 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    } else {
      __syncthreads();
      stage_b<2, KS, NH, 5>(blds, x01, tile * 32, a.n);
      __syncthreads();
      gemm_lds<2, KS>(reinterpret_cast<f32x16(&)[2]>(acc[0]), areg, blds);
      __syncthreads();
      stage_b<2, KS, NH, 5>(blds, x23, tile * 32, a.n);
      __syncthreads();
      gemm_lds<2, KS>(reinterpret_cast<f32x16(&)[2]>(acc[2]), areg, blds);
    }
    if (!live) continue;
    if (t < 4) {
      float* __restrict__ pab = a.ab + ((int64_t)(l - 1) * 4 * a.n + smp) * NH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = 32 * t + 8 * g + 4 * h;
        f32x4 ck[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ck[q] = *reinterpret_cast<const f32x4*>(pab + q * qs_ab + k0);     // c1 | cu | cs | cm of layer l - 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float hb = acc[0][4 * g + i] * pre, hub = acc[1][4 * g + i] * pre, hsb = acc[2][4 * g + i] * pre, hmb = acc[3][4 * g + i] * pre;
          const float k1 = ck[0][i], ku = ck[1][i], ks = ck[2][i], km = ck[3][i];
          o[0][i] = fmaf(km, hmb, fmaf(ks, hsb, fmaf(ku, hub, k1 * hb)));
          o[1][i] = fmaf(ks, hmb, k1 * hub);
          o[2][i] = fmaf(ku, hmb, k1 * hsb);
          o[3][i] = k1 * hmb;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(pab + q * qs_ab + k0) = o[q];
      }
    } else {        // feature columns 128..155: accumulated in the free columns of the layer-0 rows of each stream
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* __restrict__ pq = a.in + ((int64_t)q * a.n + smp) * KP + ACC_COL;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 8 * g + 4 * h;
          f32x4 sv = *reinterpret_cast<const f32x4*>(pq + f0);
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[i] += acc[q][4 * g + i];
          *reinterpret_cast<f32x4*>(pq + f0) = sv;
        }
      }
    }
  }
}

// dF[row_c] += w_c pbar + (grad w_c . u) pbar_u + (grad w_c . s) pbar_s + (u^T Hess w_c s) pbar_m; thread mapping of bwd_scatter_kernel
__global__ __launch_bounds__(256) void sm_scatter_kernel(SmArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t s = t >> 8;
  const int st = (int)((t >> 6) & 3), k = (int)((t >> 3) & 7), ch = (int)(t & 7);
  if (s >= a.n || ch >= 7) return;
  const int D = a.dims[st];
  if (D <= 1 || !a.dvols[st]) return;
  const float px = a.pts[s * 3 + 0], py = a.pts[s * 3 + 1], pz = a.pts[s * 3 + 2];
  const float vx = a.sbar[s * 3 + 0], vy = a.sbar[s * 3 + 1], vz = a.sbar[s * 3 + 2];
  float pb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pb[q] = a.in[((int64_t)q * a.n + s) * KP + ACC_COL + 7 * st + ch];
  const float vs = 2.0f / ((float)D - 1.0f);
  const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
  const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
  const int row = a.tables[st][((int64_t)xi * D + yi) * D + zi];
  if (row < 0) return;
  const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
  const float sx = (dx ? 1.0f : -1.0f) / vs, sy = (dy ? 1.0f : -1.0f) / vs, sz = (dz ? 1.0f : -1.0f) / vs;
  const float gwx = sx * wy * wz, gwy = sy * wx * wz, gwz = sz * wx * wy;
  const float hxy = sx * sy * wz, hxz = sx * sz * wy, hyz = sy * sz * wx;
  const float w0 = wx * wy * wz, wu = gwx + gwy + gwz, ws = gwx * vx + gwy * vy + gwz * vz;
  const float wm = hxy * (vx + vy) + hxz * (vx + vz) + hyz * (vy + vz);
  atomicAdd(a.dvols[st] + (int64_t)row * 8 + ch, w0 * pb[0] + wu * pb[1] + ws * pb[2] + wm * pb[3]);
}

inline unsigned grid1(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

// called from surf_sdf_backward (sdf_bwd.hip) with validated arguments
int surf_sdf_backward_layers(const float* pts, const float* ybar, const float* gbar, int64_t n, const float* const* h_vols,
                             const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                             const float* packed, float* in_v, float* in_d, float* tb, float* tdb, hipStream_t st) {
  BwdArgs a;
  a.pts = pts; a.ybar = ybar; a.gbar = gbar; a.n = n; a.packed = packed; a.in_v = in_v; a.in_d = in_d; a.tb = tb; a.tdb = tdb;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    a.dvols[s] = (s < n_vol && h_dvols) ? h_dvols[s] : nullptr;
  }
  if (n > (int64_t)0x7fffffff) return SURF_E_LIMIT;
  const int64_t tiles = (n + 31) / 32;
  const unsigned grid = (unsigned)(tiles < SURF_TM_GRID ? tiles : SURF_TM_GRID);   // workgroups walking their share of the 32-sample tiles
  hipLaunchKernelGGL(bwd_setup_kernel, dim3(grid1(n, 64)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bwd_forward_kernel<2>, dim3(grid), dim3(256), 0, st, a, 0);
  for (int l = 1; l < N_HID; ++l) hipLaunchKernelGGL(bwd_forward_kernel<10>, dim3(grid), dim3(256), 0, st, a, l);
  hipLaunchKernelGGL(bwd_reverse_kernel<0>, dim3(grid), dim3(320), 0, st, a, N_HID);
  for (int l = N_HID - 1; l >= 1; --l) {
    if (l == 2) hipLaunchKernelGGL(bwd_reverse_kernel<7>, dim3(grid), dim3(320), 0, st, a, l);    // 101 neurons: 7 k-steps
    else hipLaunchKernelGGL(bwd_reverse_kernel<8>, dim3(grid), dim3(320), 0, st, a, l);
  }
  if (h_dvols) hipLaunchKernelGGL(bwd_scatter_kernel, dim3(grid1(n * 256, 256)), dim3(256), 0, st, a);
  return surf_check_launch();
}

// called from surf_sdf_smooth_backward (sdf_smooth_bwd.hip) with validated arguments
int surf_sdf_smooth_backward_layers(const float* pts, const float* sbar, int64_t n, const float* const* h_vols,
                                    const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                                    const float* packed, float* in, float* ab, hipStream_t st) {
  SmArgs a;
  a.pts = pts; a.sbar = sbar; a.n = n; a.packed = packed; a.in = in; a.ab = ab;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    a.dvols[s] = (s < n_vol && h_dvols) ? h_dvols[s] : nullptr;
  }
  if (n > (int64_t)0x7fffff) return SURF_E_LIMIT;
  const int64_t tiles = (n + 31) / 32;
  const unsigned grid = (unsigned)(tiles < SURF_TM_GRID ? tiles : SURF_TM_GRID);
  hipLaunchKernelGGL(sm_setup_kernel, dim3(grid1(n, 64)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(sm_forward_kernel<2>, dim3(grid), dim3(256), 0, st, a, 0);
  for (int l = 1; l < N_HID; ++l) hipLaunchKernelGGL(sm_forward_kernel<10>, dim3(grid), dim3(256), 0, st, a, l);
  hipLaunchKernelGGL(sm_reverse_kernel<0>, dim3(grid), dim3(320), 0, st, a, N_HID);
  for (int l = N_HID - 1; l >= 1; --l) {
    if (l == 2) hipLaunchKernelGGL(sm_reverse_kernel<7>, dim3(grid), dim3(320), 0, st, a, l);
    else hipLaunchKernelGGL(sm_reverse_kernel<8>, dim3(grid), dim3(320), 0, st, a, l);
  }
  if (h_dvols) hipLaunchKernelGGL(sm_scatter_kernel, dim3(grid1(n * 256, 256)), dim3(256), 0, st, a);
  return surf_check_launch();
}
