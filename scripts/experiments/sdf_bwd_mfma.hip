// K9b-m (round 6): sdf_bwd_kernel (sdf_bwd.hip) on the fp32 matrix pipe.  Same function, same inputs / outputs / buffers:
// the backward of the SDF network for upstream gradients of its value (ybar) and of its spatial gradient (gbar) as ONE reverse
// sweep over a forward sweep that carries (value, tangent along gbar) - see the derivation at the head of sdf_bwd.hip
// (sdf_network.py:129-141 under loss.backward(), runner.py:163).
//
// Why: the VALU form gives a wavefront 4 samples; its k / neuron loops read the 8 broadcast operands of a weight pair out of LDS
// (two 16-byte reads returning 2 KB per 16 packed FMAs) and sit at 34 TFLOP/s - bound by the LDS return path and the FMA issue,
// not by the weight stream (round 6 measured the LDS-shared weight stream: no gain).  Here a wavefront owns 16 samples x 2
// streams = the 32 columns of a v_mfma_f32_32x32x2_f32 tile (exact fp32: the products and the accumulation are those of an fmaf
// chain):
//   forward   T[neuron][col]  = sum_k W_l[neuron][k] X[k][col]          4 row tiles x K / 2 MFMAs per layer
//   reverse   G[kin][col]     = sum_n W_l[n][kin]    D[n][col]          5 row tiles x N / 2 MFMAs per layer
// A operand = the packed weight images of sdf_smooth.hip as they are (k-major for the forward, neuron-major for the reverse),
// streamed through LDS in 16-row chunks with the register-prefetch pipeline of sdf_train_common.h (16-byte coalesced loads one
// chunk = 32 MFMAs ahead; the first form read every A element with its own 4-byte global load and waited for L2 once per k-step:
// 2.6 ms against the VALU kernel's 1.6): lane (row r = l % 32, k-slot h = l / 32) reads W[2 kk + h][32 t + r] with one
// ds_read_b32 per MFMA; B operand = the layer's columns in LDS, X[k][col] (one ds_read_b32 per MFMA quadruple); the accumulator tile leaves lane (col, h) with rows
// (r & 3) + 8 (r >> 2) + 4 h of its column, i.e. four consecutive neurons per register quadruple: the softplus algebra runs on
// them in registers (the other stream's value of the same (neuron, sample) is lane ^ 16: one cross-lane read), the saved
// coefficients / adjoints go to TB / TDB as 16-byte stores, the next layer's column back to LDS.
// Column c = 16 q + s: stream q (0 = value, 1 = tangent), sample s.
#define SURF_TRAIN_WAVES 1
#define SURF_TRAIN_CH 16
#include "sdf_train_common.h"

namespace {

constexpr int KP = 160, NH = 128, N_E = 27, N_PHI = 28, N_H2 = 101, N_HID = 6;
constexpr int OFF_WT = 0;
constexpr int OFF_W = OFF_WT + N_HID * KP * NH;
constexpr int OFF_B = OFF_W + N_HID * NH * KP;
constexpr int OFF_W6 = OFF_B + N_HID * NH;
constexpr int SW = 16;     // samples per wavefront
constexpr int XC = 32;     // columns of the LDS operand array

__host__ __device__ constexpr int layer_k(int l) { return l == 0 ? N_E : 156; }
__host__ __device__ constexpr int layer_n(int l) { return l == 2 ? N_H2 : NH; }

struct BwdArgs {
  const float* pts;
  const float* ybar;   // (n)
  const float* gbar;   // (n,3)
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

#define SURF_MFMA32(av, bv, cv) cv = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, cv, 0, 0, 0)

__global__ __launch_bounds__(64) void sdf_bwd_mfma_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[KP * XC];      // X[k][col], then D[n][col]: 20 KB
  __shared__ __attribute__((aligned(16))) float wbuf[surf_train::WBUF_FLOATS];   // two 16-row weight chunks: 20 KB (4 wavefronts per CU)
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5, q = c >> 4, s = c & 15;
  const int part = lane >> 4;                                     // set-up role: (sample s, part = 2 h + q)
  const int64_t ntot = a.n;
  const int64_t smp = (int64_t)blockIdx.x * SW + s;
  const bool live = smp < ntot;
  const int64_t sc = live ? smp : ntot - 1;
  const float inv_sqrt2 = 0.70710678118654752440f;
  const float px = a.pts[sc * 3 + 0], py = a.pts[sc * 3 + 1], pz = a.pts[sc * 3 + 2];
  const float vx = live ? a.gbar[sc * 3 + 0] : 0.f, vy = live ? a.gbar[sc * 3 + 1] : 0.f, vz = live ? a.gbar[sc * 3 + 2] : 0.f;
  const float yb = live ? a.ybar[sc] : 0.f;

  // ---- inputs ---------------------------------------------------------------------------------------------------------------
  for (int u = lane; u < KP * XC / 4; u += 64) reinterpret_cast<f32x4*>(xs)[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // positional encoding (embedder.py:11-36): part p owns the frequency 2^p (sin, cos x 3 axes); part 0 also the identity block
  float ev[9], jev[9];                                            // value and tangent along v of this lane's channels
  {
    const float p3[3] = {px, py, pz}, v3[3] = {vx, vy, vz};
    const float f = (float)(1 << part);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      float sn, cs;
      sincosf(p3[ax] * f, &sn, &cs);
      ev[ax] = sn; jev[ax] = f * cs * v3[ax];
      ev[3 + ax] = cs; jev[3 + ax] = -f * sn * v3[ax];
      ev[6 + ax] = p3[ax]; jev[6 + ax] = v3[ax];
    }
  }
  auto put_encoding = [&](int row0, float scale) {                // rows row0 + channel of both columns of sample s
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const int cs_ = 3 * (1 + 2 * part) + ax, cc_ = 3 * (2 + 2 * part) + ax;
      xs[(row0 + cs_) * XC + s] = ev[ax] * scale;
      xs[(row0 + cs_) * XC + 16 + s] = jev[ax] * scale;
      xs[(row0 + cc_) * XC + s] = ev[3 + ax] * scale;
      xs[(row0 + cc_) * XC + 16 + s] = jev[3 + ax] * scale;
      if (part == 0) {
        xs[(row0 + ax) * XC + s] = ev[6 + ax] * scale;
        xs[(row0 + ax) * XC + 16 + s] = jev[6 + ax] * scale;
      }
    }
  };
  put_encoding(0, 1.0f);
  // sparse trilinear gather (projector.py:217-390): part p owns stage p (7 channels), value and tangent along v
  {
    const int st = part;
    const int D = st < SURF_MAX_STAGES ? a.dims[st] : 0;
    float phi[7], phid[7];
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) phi[ch] = phid[ch] = 0.f;
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
        if (row < 0) continue;
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8), f1 = *reinterpret_cast<const f32x4*>(vol + (int64_t)row * 8 + 4);
        const float fv[7] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2]};
        const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
        const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
        const float w0 = wx * wy * wz;
        const float wv = (sx * wy * wz / vs) * vx + (sy * wx * wz / vs) * vy + (sz * wx * wy / vs) * vz;
#pragma unroll
        for (int ch = 0; ch < 7; ++ch) {
          phi[ch] += fv[ch] * w0;
          phid[ch] += fv[ch] * wv;
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) {
      xs[(NH + 7 * st + ch) * XC + s] = phi[ch];
      xs[(NH + 7 * st + ch) * XC + 16 + s] = phid[ch];
    }
  }
  __syncthreads();

  // this lane's column of the layer inputs -> IN_V (q = 0) / IN_D (q = 1); the two halves of a column split the rows
  auto dump_inputs = [&](int l) {
    if (!live) return;
    float* __restrict__ dst = (q ? a.in_d : a.in_v) + ((int64_t)l * ntot + smp) * KP;
#pragma unroll 4
    for (int k0 = 80 * h; k0 < 80 * h + 80; k0 += 4) {
      const f32x4 v = {xs[k0 * XC + c], xs[(k0 + 1) * XC + c], xs[(k0 + 2) * XC + c], xs[(k0 + 3) * XC + c]};
      *reinterpret_cast<f32x4*>(dst + k0) = v;
    }
  };

  // ---- forward sweep with tangents ------------------------------------------------------------------------------------------
  for (int l = 0; l < N_HID; ++l) {
    dump_inputs(l);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // (an odd K ends with a zero row of both operands: the loader zero-fills it, X row 27 is zero)
    surf_train::stream_row_pairs<NH>(a.packed + OFF_WT + l * KP * NH, layer_k(l), wbuf, [&](int kk, const float* __restrict__ wr) {
      const float b = xs[(2 * kk + h) * XC + c];
      const float* __restrict__ wa = wr + h * NH + c;
      const float a0 = wa[0], a1 = wa[32], a2 = wa[64], a3 = wa[96];
      SURF_MFMA32(a0, b, acc[0]);
      SURF_MFMA32(a1, b, acc[1]);
      SURF_MFMA32(a2, b, acc[2]);
      SURF_MFMA32(a3, b, acc[3]);
    });
    __syncthreads();                                              // every lane has read X before the outputs overwrite it
    const int N = layer_n(l);
    const float post = l == 2 ? inv_sqrt2 : 1.0f;                 // lin3's input is cat([h2, e]) / sqrt(2)
    float* __restrict__ park = (q ? a.tdb : a.tb) + ((int64_t)l * ntot + smp) * NH;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = 32 * t + 8 * g + 4 * h;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.packed + OFF_B + l * NH + n0);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mine = acc[t][4 * g + i], other = __shfl_xor(mine, 16);
          const float tv = q ? other : mine, td = q ? mine : other;
          const Act A = softplus100(tv + bias[i]);
          const bool real = n0 + i < N;
          // parked for the reverse sweep: sp'(t) in TB (value lanes), sp''(t) t' in TDB (tangent lanes)
          o[i] = real ? (q ? A.s2 * td : A.s1) : 0.f;
          xs[(n0 + i) * XC + c] = real ? (q ? A.s1 * td * post : A.h * post) : 0.f;
        }
        if (live) *reinterpret_cast<f32x4*>(park + n0) = o;
      }
    if (l == 2) {                                                 // slots 101..127 take the encoding of the skip connection
      __syncthreads();
      put_encoding(N_H2, inv_sqrt2);
    }
    __syncthreads();
  }
  dump_inputs(N_HID);   // inputs of lin6 (its row 0 alone reaches the loss)

  // ---- reverse sweep: adjoints of (pre-activation, its tangent) ---------------------------------------------------------------
  float padj[16];       // adjoint of feature (r & 3) + 8 (r >> 2) + 4 h of this column: phibar (q = 0) / phi'bar (q = 1)
#pragma unroll
  for (int r = 0; r < 16; ++r) padj[r] = 0.f;
  for (int l = N_HID; l >= 1; --l) {
    f32x16 g[5];
    if (l == N_HID) {   // tbar_6 = ybar e_0, t'bar_6 = e_0: the adjoints of lin6's inputs are its row 0 (x ybar)
      const float seed = q ? (live ? 1.0f : 0.f) : yb;
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = a.packed[OFF_W6 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * seed;
    } else {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = 0.f;
      // (an odd N ends with a zero row of both operands: D row 101 of the 101-wide layer is zero)
      surf_train::stream_row_pairs<KP>(a.packed + OFF_W + l * NH * KP, layer_n(l), wbuf, [&](int nn, const float* __restrict__ wr) {
        const float b = xs[(2 * nn + h) * XC + c];
        const float* __restrict__ wa = wr + h * KP + c;
        const float a0 = wa[0], a1 = wa[32], a2 = wa[64], a3 = wa[96], a4 = wa[128];
        SURF_MFMA32(a0, b, g[0]);
        SURF_MFMA32(a1, b, g[1]);
        SURF_MFMA32(a2, b, g[2]);
        SURF_MFMA32(a3, b, g[3]);
        SURF_MFMA32(a4, b, g[4]);
      });
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) padj[r] += g[4][r];              // feature columns 128..155
    const float pre = l == 3 ? inv_sqrt2 : 1.0f;
    float* __restrict__ ptb = a.tb + ((int64_t)(l - 1) * ntot + sc) * NH;
    float* __restrict__ ptdb = a.tdb + ((int64_t)(l - 1) * ntot + sc) * NH;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int k0 = 32 * t + 8 * gq + 4 * h;
        f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = c1;                 // sp' and sp'' t' of layer l - 1 (dead samples: zeros)
        if (live) {
          c1 = *reinterpret_cast<const f32x4*>(ptb + k0);
          if (!q) c2 = *reinterpret_cast<const f32x4*>(ptdb + k0);
        }
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mine = g[t][4 * gq + i] * pre, other = __shfl_xor(mine, 16);
          const float hb = q ? other : mine, hdb = q ? mine : other;     // adjoints of (h, h') of layer l - 1
          o[i] = q ? c1[i] * hdb : fmaf(c2[i], hdb, c1[i] * hb);
          xs[(k0 + i) * XC + c] = o[i];
        }
        if (live) *reinterpret_cast<f32x4*>((q ? ptdb : ptb) + k0) = o;
      }
    __syncthreads();
  }

  // ---- feature gradients: dF[row_c] += w_c phibar + (grad w_c . v) phi'bar ----------------------------------------------------
  // lane (sample s, stream q, half h) holds phibar / phi'bar of features f(r) = (r & 3) + 8 (r >> 2) + 4 h; the partner stream's
  // value is lane ^ 16.  The two lanes of a pair share the 8 corners: q takes corners 4 q .. 4 q + 3.
  float pb[16], pdb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float other = __shfl_xor(padj[r], 16);
    pb[r] = q ? other : padj[r];
    pdb[r] = q ? padj[r] : other;
  }
  if (!live) return;
#pragma unroll
  for (int st = 0; st < SURF_MAX_STAGES; ++st) {
    const int D = a.dims[st];
    if (D <= 1 || !a.dvols[st]) continue;
    const int32_t* __restrict__ table = a.tables[st];
    float* __restrict__ dvol = a.dvols[st];
    const float vs = 2.0f / ((float)D - 1.0f);
    const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int k = 4 * q + kc;
      const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
      const int xi = min(max(x0 + dx, 0), D - 1), yi = min(max(y0 + dy, 0), D - 1), zi = min(max(z0 + dz, 0), D - 1);
      const int row = table[((int64_t)xi * D + yi) * D + zi];
      if (row < 0) continue;
      const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
      const float sx = dx ? 1.0f : -1.0f, sy = dy ? 1.0f : -1.0f, sz = dz ? 1.0f : -1.0f;
      const float w0 = wx * wy * wz;
      const float wv = (sx * wy * wz / vs) * vx + (sy * wx * wz / vs) * vy + (sz * wx * wy / vs) * vz;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (f >= 7 * st && f < 7 * st + 7) atomicAdd(dvol + (int64_t)row * 8 + (f - 7 * st), w0 * pb[r] + wv * pdb[r]);
      }
    }
  }
}

}  // namespace

// called from surf_sdf_backward (sdf_bwd.hip); the arguments are already validated there
int surf_sdf_backward_mfma_launch(const float* pts, const float* ybar, const float* gbar, int64_t n, const float* const* h_vols,
                                  const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                                  const float* packed, float* in_v, float* in_d, float* tb, float* tdb, hipStream_t stream) {
  BwdArgs a;
  a.pts = pts; a.ybar = ybar; a.gbar = gbar; a.n = n; a.packed = packed; a.in_v = in_v; a.in_d = in_d; a.tb = tb; a.tdb = tdb;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : nullptr;
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    a.dvols[s] = (s < n_vol && h_dvols) ? h_dvols[s] : nullptr;
  }
  const int64_t blocks = (n + SW - 1) / SW;
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  hipLaunchKernelGGL(sdf_bwd_mfma_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a);
  return surf_check_launch();
}
