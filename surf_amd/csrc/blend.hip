// K10: multi-view feature fetch + IBRNet-style blending MLP, fp32 MFMA.
//
// Restates lookup_feature / compute_angle   projector.py:485-556
//          BlendingNetwork.forward           blending_network.py:69-118
//
// Same register-resident scheme as sdf_mlp.hip: one wavefront owns 32 sample points, lane l =
// (sample j = l & 31, half h = l >> 5); every Linear is W (A operand) x activations^T (B operand) with
// v_mfma_f32_32x32x2_f32, so a layer's 32x32 accumulator tile is directly the next layer's B operand.
// The 19 per-view input channels [rgb(3) F0(4) F1(4) | F2(4) F3(4)] are split between the lane halves:
// half 0 fetches the image and pyramid levels 0,1 (11 channels), half 1 fetches levels 2,3 (8 channels),
// so no texel is fetched twice.  Source views are unrolled (template NS); per view the chain is 144 MFMAs,
// plus 48 per tile for the view-independent [mean | var] part of base_fc.0.
#include <math.h>

#include <type_traits>

#include "blend_raw.h"
#include "common.h"



namespace {

constexpr int TILE = 32;

using namespace blend_raw;

// ---- packed buffer (floats).  MFMA layers: [q][t][lane][4], NQ groups of 4 k-steps, NT tiles ---------
enum { L_RD0, L_RD2, L_B0S, L_B0V, L_B2, L_V0, L_V2, L_W0, L_R0, L_R2, N_MMA };
constexpr int LNQ[N_MMA] = {1, 2, 6, 3, 8, 4, 4, 4, 5, 2};
constexpr int LNT[N_MMA] = {1, 1, 2, 2, 1, 1, 1, 1, 1, 1};
constexpr int mma_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += LNQ[i] * LNT[i] * 256; return o; }
constexpr int MMA_END = mma_off(N_MMA);
// biases as accumulator init: [h][16] per tile
enum { B_RD0, B_RD2, B_B0_T0, B_B0_T1, B_B2, B_V0, B_V2, B_W0, B_R0, B_R2, N_BIAS };
constexpr int BIAS_OFF = MMA_END;
// per-lane dot-product rows [h][16]: vis (row 32 of vis_fc.2), vis_fc2.2, rgb_fc.4
enum { D_VIS, D_VIS2, D_RGB4, N_DOT };
constexpr int DOT_OFF = BIAS_OFF + N_BIAS * 32;
constexpr int SCAL_OFF = DOT_OFF + N_DOT * 32;  // [|s|, b_vis, b_vis2, b_rgb4]
constexpr int PACKED_FLOATS = SCAL_OFF + 4;

struct BlendArgs {
  const float* pts;
  const uint8_t* mask;
  const int32_t* idx;  // optional list of point indices (n entries)
  int64_t n;
  const float* feats[4];
  int hw[8];
  const float* imgs;
  float K[SURF_MAX_VIEWS][9];
  float w2c[SURF_MAX_VIEWS][12];
  float cpos[SURF_MAX_VIEWS][3];
  const float* w;
  float* color;
  uint8_t* n_valid;
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ f32x4 bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// ELU / sigmoid through the raw v_exp_f32 (absolute error <= 1e-7, far inside the 1e-3 colour tolerance)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : fexp(x) - 1.0f; }
// The same function on a pair, written so that the adds and multiplies are packed (v_pk_*) and the select disappears:
// elu(x) = max(x, 0) + (min(exp(x), 1) - 1); the min is the clamp modifier of v_exp_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 elu2(f32x2 x) {
  const f32x2 a = x * 1.44269504088896341f;
  f32x2 e, m;
  e[0] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(a[0]), 0.0f, 1.0f);
  e[1] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(a[1]), 0.0f, 1.0f);
  m[0] = __builtin_amdgcn_fmed3f(x[0], 0.0f, 3.0e38f);
  m[1] = __builtin_amdgcn_fmed3f(x[1], 0.0f, 3.0e38f);
  return m + (e - 1.0f);
}
// out[r] = elu(acc[r] + bias[r]) for r < N (N even)
template <int N>
__device__ __forceinline__ void elu_rows(const f32x16& acc, const f32x16& bias, float* out) {
#pragma unroll
  for (int r = 0; r < N; r += 2) {
    const f32x2 xa = {acc[r], acc[r + 1]}, xb = {bias[r], bias[r + 1]};
    const f32x2 y = elu2(xa + xb);
    out[r] = y[0];
    out[r + 1] = y[1];
  }
}
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }

// Bilinear fetch of a texel4 map with zero padding, branch-free: out-of-range taps read a clamped texel with
// weight 0, so all four 16-byte loads of a fetch (and of every fetch of a view) can be in flight together.
struct Tap4 {
  f32x4 v[4];
  float w[4];
};
__device__ __forceinline__ void tap_issue(Tap4& t, const float* __restrict__ map, int H, int W, float x, float y) {
  const float fx = floorf(x), fy = floorf(y);
  const float tx = x - fx, ty = y - fy;
  const int x0 = (int)fx, y0 = (int)fy;
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // order (y0,x0), (y0,x1), (y1,x0), (y1,x1) = grid_sample's nw, ne, sw, se
    const int dx = k & 1, dy = k >> 1;
    const int xi = x0 + dx, yi = y0 + dy;
    const bool ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
    const int xc = min(max(xi, 0), W - 1), yc = min(max(yi, 0), H - 1);
    t.v[k] = *reinterpret_cast<const f32x4*>(map + ((int64_t)yc * W + xc) * 4);
    t.w[k] = ok ? (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) : 0.0f;
  }
}
__device__ __forceinline__ f32x4 tap_finish(const Tap4& t) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += t.v[k] * t.w[k];
  return acc;
}

// ---- one weight stream per tile: [RD0, RD2] x NS, B0S, [B0V, B2, V0, V2, W0, R0, R2] x NS ----------------------
// A group = 16 bytes per lane = 4 k-steps of one 32-row tile.  A ring of RPF+1 register buffers keeps RPF groups in
// flight across layer, view and section boundaries, so no layer starts by waiting for its first weights.  Every
// section is a multiple of RPF+1 = 3 groups long, so a group's ring slot depends only on its position in its section
// (which keeps all register indices compile-time constants although the views are a loop).
constexpr int RPF = 2;
struct WRing { f32x4 a[RPF + 1]; };

struct SeqNone { static constexpr int LEN = 0; static constexpr int off(int) { return 0; } };
struct SeqPV {  // per-view chain of pass 2
  static constexpr int LEN = 33;
  static constexpr int off(int p) {
    if (p < 6) return mma_off(L_B0V) * 4 + p * 1024;
    if (p < 14) return mma_off(L_B2) * 4 + (p - 6) * 1024;
    if (p < 18) return mma_off(L_V0) * 4 + (p - 14) * 1024;
    if (p < 22) return mma_off(L_V2) * 4 + (p - 18) * 1024;
    if (p < 26) return mma_off(L_W0) * 4 + (p - 22) * 1024;
    if (p < 31) return mma_off(L_R0) * 4 + (p - 26) * 1024;
    return mma_off(L_R2) * 4 + (p - 31) * 1024;
  }
};
struct SeqBS {  // view-independent [mean | var] part of base_fc.0, two tiles, [q][t] order
  static constexpr int LEN = 12;
  static constexpr int off(int p) { return mma_off(L_B0S) * 4 + p * 1024; }
};
struct SeqP1 {  // direction MLP of pass 1
  static constexpr int LEN = 3;
  static constexpr int off(int p) { return p == 0 ? mma_off(L_RD0) * 4 : mma_off(L_RD2) * 4 + (p - 1) * 1024; }
};
static_assert(SeqPV::LEN % (RPF + 1) == 0 && SeqBS::LEN % (RPF + 1) == 0 && SeqP1::LEN % (RPF + 1) == 0, "ring phase");

// issue the prefetch that belongs to position p of section SEQ (it targets position p + RPF, possibly in the next
// repetition of SEQ or, for the last repetition, in section NEXT)
template <class SEQ, class NEXT>
__device__ __forceinline__ void ring_prefetch(WRing& ring, int p, bool last, rsrc_t wr, int lane16) {
  const int t = p + RPF;
  if (t < SEQ::LEN) ring.a[t % (RPF + 1)] = bload(wr, lane16, SEQ::off(t));
  else if (!last) ring.a[t % (RPF + 1)] = bload(wr, lane16, SEQ::off(t - SEQ::LEN));
  else if (NEXT::LEN > 0) ring.a[t % (RPF + 1)] = bload(wr, lane16, NEXT::off(t - SEQ::LEN));
}

// NQ groups of one 32-row tile starting at position POS of section SEQ
template <class SEQ, class NEXT, int POS, int NQ>
__device__ __forceinline__ void stream_mma(WRing& ring, f32x16& acc, const float* b, bool last, rsrc_t wr, int lane16) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ring_prefetch<SEQ, NEXT>(ring, POS + q, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + q) % (RPF + 1)][i], b[q * 4 + i], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// NQ k-groups of two 32-row tiles (stream order [q][t])
template <class SEQ, class NEXT, int POS, int NQ>
__device__ __forceinline__ void stream_mma2(WRing& ring, f32x16& acc0, f32x16& acc1, const float* b, bool last, rsrc_t wr,
                                            int lane16) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ring_prefetch<SEQ, NEXT>(ring, POS + 2 * q, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + 2 * q) % (RPF + 1)][i], b[q * 4 + i], acc0, 0, 0, 0);
    ring_prefetch<SEQ, NEXT>(ring, POS + 2 * q + 1, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + 2 * q + 1) % (RPF + 1)][i], b[q * 4 + i], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same for NV views at once: every weight group is fetched once and used by all of them, and the views' accumulator
// chains are independent (no dependent-MFMA issue delay).  (Letting VALU work cross the scheduling barrier, so that one
// view's activations could be scheduled beside another's MFMAs, measured 4-5 % slower than pinning everything.)
#define SURF_BLEND_SB() __builtin_amdgcn_sched_barrier(0)
template <class SEQ, class NEXT, int POS, int NQ, int NV>
__device__ __forceinline__ void stream_mma_v(WRing& ring, f32x16 (&acc)[NV], const float* const (&b)[NV], bool last, rsrc_t wr,
                                             int lane16) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ring_prefetch<SEQ, NEXT>(ring, POS + q, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < NV; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + q) % (RPF + 1)][i], b[u][q * 4 + i], acc[u], 0, 0, 0);
    SURF_BLEND_SB();
  }
}
template <class SEQ, class NEXT, int POS, int NQ, int NV>
__device__ __forceinline__ void stream_mma2_v(WRing& ring, f32x16 (&acc0)[NV], f32x16 (&acc1)[NV], const float* const (&b)[NV],
                                              bool last, rsrc_t wr, int lane16) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ring_prefetch<SEQ, NEXT>(ring, POS + 2 * q, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < NV; ++u)
        acc0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + 2 * q) % (RPF + 1)][i], b[u][q * 4 + i], acc0[u], 0, 0, 0);
    ring_prefetch<SEQ, NEXT>(ring, POS + 2 * q + 1, last, wr, lane16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < NV; ++u)
        acc1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(POS + 2 * q + 1) % (RPF + 1)][i], b[u][q * 4 + i], acc1[u], 0, 0, 0);
    SURF_BLEND_SB();
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  return v;
}

__device__ __forceinline__ f32x16 load_row16(rsrc_t wr, int h64, int off_floats) {
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 x = bload(wr, h64, off_floats * 4 + g * 16);
    v[4 * g + 0] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
  }
  return v;
}

template <int NS>
// Two wavefronts per SIMD only while the per-view state fits 256 registers (NS <= 2); with more source views one
// wavefront per SIMD and the whole register file beats two spilling ones (57.9 vs 67.1 ms at NS = 4).
__global__ __launch_bounds__(256, (NS <= 2 ? 2 : 1)) void blend_kernel(BlendArgs a) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t n_tiles = (a.n + TILE - 1) / TILE;
  const rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, PACKED_FLOATS * 4, 0x00020000);
  const int lane16 = lane * 16, h64 = h * 64;
  const float s_abs = a.w[SCAL_OFF + 0];

  // the two pyramid levels this half fetches
  const int lvA = h ? 2 : 0, lvB = h ? 3 : 1;
  const float* __restrict__ mapA = h ? a.feats[2] : a.feats[0];
  const float* __restrict__ mapB = h ? a.feats[3] : a.feats[1];
  const int HA = h ? a.hw[4] : a.hw[0], WA = h ? a.hw[5] : a.hw[1];
  const int HB = h ? a.hw[6] : a.hw[2], WB = h ? a.hw[7] : a.hw[3];
  const float scA = h ? 0.25f : 1.0f, scB = h ? 0.125f : 0.5f;
  (void)lvA; (void)lvB;

  for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    const int64_t slot = tile * TILE + j;
    const int64_t sc = slot < a.n ? slot : a.n - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot < a.n) && (!a.mask || a.mask[i] != 0);
    if (__ballot(active) == 0ull) continue;
    const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];

    WRing ring;
#pragma unroll
    for (int p = 0; p < RPF; ++p) ring.a[p] = bload(wr, lane16, SeqP1::off(p));

    float floc[NS][12];  // this half's channels of rgb_feat + direction feature (11 or 8 used)
    float rgb[NS][3];    // raw source colours (meaningful in half 0)
    float rd[NS][4];
    float mk[NS], ex[NS];

    float ax = a.cpos[0][0] - px, ay = a.cpos[0][1] - py, az = a.cpos[0][2] - pz;
    {
      float nn = sqrtf(ax * ax + ay * ay + az * az) + 1e-6f;
      ax /= nn; ay /= nn; az /= nn;
    }
    int nvalid = 0;
    // ------------------------------ pass 1a: projections, all texel fetches of all views issued together ------
    // (one wavefront per SIMD: nothing else hides the fetch latency, so it is exposed once, not once per view;
    //  with two wavefronts per SIMD (NS <= 2) the register budget still covers 24 fetches)
    //  with more than four source views the fetches go in groups of three: 36 fetches in flight fit the registers)
    constexpr int GV = NS <= 4 ? NS : 3;
    Tap4 qA[NS], qB[NS], qC[NS];
    bool okv[NS];
#pragma unroll
    for (int v0 = 0; v0 < NS; v0 += GV) {
#pragma unroll
    for (int v = v0; v < (v0 + GV < NS ? v0 + GV : NS); ++v) {
      const int cam = v + 1;
      // ray_diff (projector.py:485-498)
      float bx = a.cpos[cam][0] - px, by = a.cpos[cam][1] - py, bz = a.cpos[cam][2] - pz;
      float nn = sqrtf(bx * bx + by * by + bz * bz) + 1e-6f;
      bx /= nn; by /= nn; bz /= nn;
      float ddx = ax - bx, ddy = ay - by, ddz = az - bz;
      float dn = fmaxf(sqrtf(ddx * ddx + ddy * ddy + ddz * ddz), 1e-6f);
      rd[v][0] = ddx / dn; rd[v][1] = ddy / dn; rd[v][2] = ddz / dn;
      rd[v][3] = ax * bx + ay * by + az * bz;
      // projection (projector.py:527-539); level l uses intrinsics rows 0,1 x 0.5^l = exact scaling of u,v
      const float* M = a.w2c[cam];
      float X = M[0] * px + M[1] * py + M[2] * pz + M[3];
      float Y = M[4] * px + M[5] * py + M[6] * pz + M[7];
      float Z = M[8] * px + M[9] * py + M[10] * pz + M[11];
      const float* K = a.K[cam];
      float qx = K[0] * X + K[1] * Y + K[2] * Z;
      float qy = K[3] * X + K[4] * Y + K[5] * Z;
      float qz = K[6] * X + K[7] * Y + K[8] * Z;
      float u0 = qx / qz, v0 = qy / qz;
      bool ok = qz > 0.f;
      {
        float u = u0 * scA, vv = v0 * scA;
        ok = ok && (u >= 0.f) && (u < (float)WA) && (vv >= 0.f) && (vv < (float)HA);
        float nx = u / ((float)(WA - 1) / 2.0f) - 1.0f, ny = vv / ((float)(HA - 1) / 2.0f) - 1.0f;
        float gx = ((nx + 1.0f) * (float)WA - 1.0f) / 2.0f, gy = ((ny + 1.0f) * (float)HA - 1.0f) / 2.0f;
        tap_issue(qA[v], mapA + (int64_t)cam * HA * WA * 4, HA, WA, gx, gy);
        // half 1 re-reads its level-A taps instead of the image (same addresses: L1 hits), result unused
        tap_issue(qC[v], (h == 0 ? a.imgs : mapA) + (int64_t)cam * HA * WA * 4, HA, WA, gx, gy);
      }
      {
        float u = u0 * scB, vv = v0 * scB;
        ok = ok && (u >= 0.f) && (u < (float)WB) && (vv >= 0.f) && (vv < (float)HB);
        float nx = u / ((float)(WB - 1) / 2.0f) - 1.0f, ny = vv / ((float)(HB - 1) / 2.0f) - 1.0f;
        float gx = ((nx + 1.0f) * (float)WB - 1.0f) / 2.0f, gy = ((ny + 1.0f) * (float)HB - 1.0f) / 2.0f;
        tap_issue(qB[v], mapB + (int64_t)cam * HB * WB * 4, HB, WB, gx, gy);
      }
      okv[v] = ok;
    }
    // ------------------------------ pass 1b: direction feature per view ----------------------------------------
#pragma unroll
    for (int v = v0; v < (v0 + GV < NS ? v0 + GV : NS); ++v) {
      const f32x4 tA = tap_finish(qA[v]), tB = tap_finish(qB[v]);
      f32x4 tC = tap_finish(qC[v]);
      if (h != 0) tC = f32x4{0.f, 0.f, 0.f, 0.f};
      bool ok = okv[v];
      ok = ok && (__shfl_xor((int)ok, 32) != 0);  // AND over all four levels
      mk[v] = ok ? 1.f : 0.f;
      nvalid += ok ? 1 : 0;
      rgb[v][0] = tC[0]; rgb[v][1] = tC[1]; rgb[v][2] = tC[2];
      // local channel order: half 0 = [rgb, F0, F1], half 1 = [F2, F3, 0, 0, 0]
      float g[12];
      if (h == 0) {
        g[0] = tC[0]; g[1] = tC[1]; g[2] = tC[2];
        g[3] = tA[0]; g[4] = tA[1]; g[5] = tA[2]; g[6] = tA[3];
        g[7] = tB[0]; g[8] = tB[1]; g[9] = tB[2]; g[10] = tB[3];
      } else {
        g[0] = tA[0]; g[1] = tA[1]; g[2] = tA[2]; g[3] = tA[3];
        g[4] = tB[0]; g[5] = tB[1]; g[6] = tB[2]; g[7] = tB[3];
        g[8] = g[9] = g[10] = 0.f;
      }
      g[11] = 0.f;
      // direction feature ELU(L(ELU(L(ray_diff))))  4 -> 16 -> 19  (blending_network.py:72-74)
      float bin[4] = {h ? rd[v][1] : rd[v][0], h ? rd[v][3] : rd[v][2], 0.f, 0.f};
      // bias rows are loaded before the MFMAs they follow and added in the activation: their latency hides under
      // the matrix work and no accumulator waits for an initial value
      int h64p = h64;
      asm volatile("" : "+v"(h64p));  // see pass 2: keeps the per-view bias loads from being merged and kept live
      const f32x16 bias1 = load_row16(wr, h64p, BIAS_OFF + B_RD0 * 32);
      const f32x16 bias2 = load_row16(wr, h64p, BIAS_OFF + B_RD2 * 32);
      const bool lastv = (v == NS - 1);
      f32x16 acc1 = zero16();
      stream_mma<SeqP1, SeqBS, 0, 1>(ring, acc1, bin, lastv, wr, lane16);
      float h8[8];
      elu_rows<8>(acc1, bias1, h8);
      f32x16 acc2 = zero16();
      stream_mma<SeqP1, SeqBS, 1, 2>(ring, acc2, h8, lastv, wr, lane16);
      {
        // rows of half 1 beyond its 8 channels carry zero weights and zero bias: elu(0) = 0
        float d12[12];
        elu_rows<12>(acc2, bias2, d12);
#pragma unroll
        for (int r = 0; r < 11; ++r) floc[v][r] = g[r] + d12[r];
      }
      floc[v][11] = 0.f;
      ex[v] = expf(s_abs * (rd[v][3] - 1.0f));
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (a.n_valid && active && h == 0) a.n_valid[i] = (uint8_t)nvalid;

    // ------------------------------ pooling weights, weighted mean / variance (:76-86) ----------------------
    float emin = ex[0];
#pragma unroll
    for (int v = 1; v < NS; ++v) emin = fminf(emin, ex[v]);
    float wv[NS], wsum = 0.f;
#pragma unroll
    for (int v = 0; v < NS; ++v) { wv[v] = (ex[v] - emin) * mk[v]; wsum += wv[v]; }
#pragma unroll
    for (int v = 0; v < NS; ++v) wv[v] = wv[v] / (wsum + 1e-8f);
    float mv[24];  // B operands of the view-independent part: [mean(12) | var(12)]
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      float mean = 0.f;
#pragma unroll
      for (int v = 0; v < NS; ++v) mean += floc[v][c] * wv[v];
      float var = 0.f;
#pragma unroll
      for (int v = 0; v < NS; ++v) { float d = floc[v][c] - mean; var += wv[v] * (d * d); }
      mv[c] = mean;
      mv[12 + c] = var;
    }
    // view-independent part of base_fc.0 (bias added at the activation below)
    f32x16 G0a = zero16(), G0b = zero16();
    stream_mma2<SeqBS, SeqPV, 0, 6>(ring, G0a, G0b, mv, true, wr, lane16);

    // ------------------------------ pass 2: per-view chain, online softmax over views (:88-116) -------------
    const float b_vis = a.w[SCAL_OFF + 1], b_vis2 = a.w[SCAL_OFF + 2], b_rgb4 = a.w[SCAL_OFF + 3];
    float Mx = -INFINITY, Zs = 0.f, o_r = 0.f, o_g = 0.f, o_b = 0.f;
    // NV source views per pass over the weight stream (two at a time, a last single one when NS is odd)
    auto chain = [&](auto nv_tag, const int v0, const bool lastv) __attribute__((always_inline)) {
      constexpr int NV = decltype(nv_tag)::value;
      // Per-chain copy of the row offset that the compiler cannot see through: otherwise the (identical) bias-row loads
      // of all chains are merged into one set of 160 registers kept live across the whole pass.
      int h64v = h64;
      asm volatile("" : "+v"(h64v));
      // base_fc.0 (view part) + ELU : 57 -> 64
      const f32x16 bb0 = load_row16(wr, h64v, BIAS_OFF + B_B0_T0 * 32), bb1 = load_row16(wr, h64v, BIAS_OFF + B_B0_T1 * 32);
      f32x16 a64a[NV], a64b[NV];
      const float* bp[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) { a64a[u] = G0a; a64b[u] = G0b; bp[u] = floc[v0 + u]; }
      stream_mma2_v<SeqPV, SeqNone, 0, 3, NV>(ring, a64a, a64b, bp, lastv, wr, lane16);
      float h32[NV][32];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        elu_rows<16>(a64a[u], bb0, h32[u]);
        elu_rows<16>(a64b[u], bb1, h32[u] + 16);
        bp[u] = h32[u];
      }
      // base_fc.2 + ELU : 64 -> 32
      const f32x16 bx = load_row16(wr, h64v, BIAS_OFF + B_B2 * 32);
      f32x16 accx[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) accx[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 6, 8, NV>(ring, accx, bp, lastv, wr, lane16);
      float x[NV][16], xin[NV][16];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        elu_rows<16>(accx[u], bx, x[u]);
#pragma unroll
        for (int r = 0; r < 16; ++r) xin[u][r] = x[u][r] * wv[v0 + u];
        bp[u] = xin[u];
      }
      // vis_fc: 32 -> 32 (ELU) -> 33 (ELU)
      const f32x16 bt = load_row16(wr, h64v, BIAS_OFF + B_V0 * 32);
      f32x16 acct[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) acct[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 14, 4, NV>(ring, acct, bp, lastv, wr, lane16);
      float t16[NV][16];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        elu_rows<16>(acct[u], bt, t16[u]);
        bp[u] = t16[u];
      }
      const f32x16 br = load_row16(wr, h64v, BIAS_OFF + B_V2 * 32);
      const f32x16 dvis = load_row16(wr, h64v, DOT_OFF + D_VIS * 32);
      f32x16 accr[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) accr[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 18, 4, NV>(ring, accr, bp, lastv, wr, lane16);
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        float vraw = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) vraw = fmaf(dvis[r], t16[u][r], vraw);
        vraw += __shfl_xor(vraw, 32);
        const float vis = sigm(elu(vraw + b_vis)) * mk[v0 + u];
        float d16[16];
        elu_rows<16>(accr[u], br, d16);
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[u][r] = x[u][r] + d16[r]; xin[u][r] = x[u][r] * vis; }
        bp[u] = xin[u];
      }
      // vis_fc2: 32 -> 32 (ELU) -> 1 (sigmoid)
      const f32x16 bw = load_row16(wr, h64v, BIAS_OFF + B_W0 * 32);
      const f32x16 dvis2 = load_row16(wr, h64v, DOT_OFF + D_VIS2 * 32);
      f32x16 accw[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) accw[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 22, 4, NV>(ring, accw, bp, lastv, wr, lane16);
      // rgb_fc: [x(32), vis, ray_diff(4)] = 37 -> 16 (ELU) -> 8 (ELU) -> 1
      float rin[NV][20];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int v = v0 + u;
        float v2 = 0.f;
        float d16[16];
        elu_rows<16>(accw[u], bw, d16);
#pragma unroll
        for (int r = 0; r < 16; ++r) v2 = fmaf(dvis2[r], d16[r], v2);
        v2 += __shfl_xor(v2, 32);
        const float vis2 = sigm(v2 + b_vis2) * mk[v];
#pragma unroll
        for (int r = 0; r < 16; ++r) rin[u][r] = x[u][r];
        rin[u][16] = h ? rd[v][0] : vis2;
        rin[u][17] = h ? rd[v][2] : rd[v][1];
        rin[u][18] = h ? 0.f : rd[v][3];
        rin[u][19] = 0.f;
        bp[u] = rin[u];
      }
      const f32x16 b16 = load_row16(wr, h64v, BIAS_OFF + B_R0 * 32);
      f32x16 acc16[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) acc16[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 26, 5, NV>(ring, acc16, bp, lastv, wr, lane16);
      float r8[NV][8];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        elu_rows<8>(acc16[u], b16, r8[u]);
        bp[u] = r8[u];
      }
      const f32x16 b8 = load_row16(wr, h64v, BIAS_OFF + B_R2 * 32);
      const f32x16 drgb4 = load_row16(wr, h64v, DOT_OFF + D_RGB4 * 32);
      f32x16 acc8[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) acc8[u] = zero16();
      stream_mma_v<SeqPV, SeqNone, 31, 2, NV>(ring, acc8, bp, lastv, wr, lane16);
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int v = v0 + u;
        float rr = 0.f;
        float d4[4];
        elu_rows<4>(acc8[u], b8, d4);
#pragma unroll
        for (int r = 0; r < 4; ++r) rr = fmaf(drgb4[r], d4[r], rr);
        rr += __shfl_xor(rr, 32);
        rr += b_rgb4;
        if (mk[v] == 0.f) rr = -1e9f;
        const float Mn = fmaxf(Mx, rr);
        const float sc = expf(Mx - Mn);  // exp(-inf) = 0 on the first view
        const float e = expf(rr - Mn);
        Zs = Zs * sc + e;
        o_r = o_r * sc + e * rgb[v][0];
        o_g = o_g * sc + e * rgb[v][1];
        o_b = o_b * sc + e * rgb[v][2];
        Mx = Mn;
      }
    };
    // views per chain: all of them up to four (52.0 ms vs 54.9 one by one at NS = 4); beyond that the split that does not spill
    constexpr int VPC = NS <= 4 ? NS : (NS == 5 ? 4 : (NS == 6 ? 2 : 3));
#pragma unroll
    for (int v = 0; v + VPC <= NS; v += VPC) chain(std::integral_constant<int, VPC>{}, v, v + VPC == NS);
    if (NS % VPC != 0) chain(std::integral_constant<int, (NS % VPC) ? (NS % VPC) : 1>{}, NS - NS % VPC, true);
    if (active && h == 0) {
      a.color[i * 3 + 0] = o_r / Zs;
      a.color[i * 3 + 1] = o_g / Zs;
      a.color[i * 3 + 2] = o_b / Zs;
    }
  }
}

int grid_blocks(int64_t n) {
  int64_t tiles = (n + TILE - 1) / TILE;
  int64_t blocks = (tiles + 3) / 4;
  return (int)(blocks < 4096 ? blocks : 4096);
}

// ---- host packer --------------------------------------------------------------------------------------------
inline int hk(int tt, int r, int h) { return 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h; }
// local channel index (register r of half h) -> channel of the 19-vector, -1 = pad
inline int loc_ch(int r, int h) { return h == 0 ? (r < 11 ? r : -1) : (r < 8 ? 11 + r : -1); }

template <class RowF, class ColF>
void pack_mma(float* dst, int NQ, int NT, const float* W, int ldw, RowF row_of, ColF col_of) {
  for (int q = 0; q < NQ; ++q)
    for (int t = 0; t < NT; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
          const int step = 4 * q + i, h = lane >> 5, rho = lane & 31;
          const int row = row_of(t, rho), col = col_of(step, h);
          dst[((q * NT + t) * 64 + lane) * 4 + i] = (row >= 0 && col >= 0) ? W[row * ldw + col] : 0.f;
        }
}

template <class RowF>
void pack_rows(float* dst, const float* b, RowF feat_of) {  // [h][16] <- b[feat_of(r,h)]
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 16; ++r) {
      int f = feat_of(r, h);
      dst[h * 16 + r] = f >= 0 ? b[f] : 0.f;
    }
}

}  // namespace

extern "C" int surf_blend_raw_floats(void) { return RAW_FLOATS; }
extern "C" int surf_blend_packed_floats(void) { return PACKED_FLOATS; }

extern "C" int surf_blend_pack_weights(const float* raw, float* out) {
  if (!raw || !out) return SURF_E_ARG;
  for (int i = 0; i < PACKED_FLOATS; ++i) out[i] = 0.f;
  auto nat_row = [](int lim) { return [lim](int t, int rho) { int f = 32 * t + rho; return f < lim ? f : -1; }; };
  auto nat_col = [](int lim) { return [lim](int step, int h) { int f = hk(step / 16, step % 16, h); return f < lim ? f : -1; }; };
  // row map of the direction-feature output: D row rho -> (r, h_row) -> local channel
  auto dir_row = [](int t, int rho) { int hr = (rho >> 2) & 1, r = (rho & 3) | ((rho >> 3) << 2); return loc_ch(r, hr); };
  auto natf = [](int lim) { return [lim](int r, int h) { int f = hk(0, r, h); return f < lim ? f : -1; }; };

  // ray_dir_fc.0: 4 -> 16; steps: (rd0|rd1), (rd2|rd3)
  pack_mma(out + mma_off(L_RD0), 1, 1, raw + R_RD0_W, 4, nat_row(16),
           [](int step, int h) { return step < 2 ? 2 * step + h : -1; });
  pack_rows(out + BIAS_OFF + B_RD0 * 32, raw + R_RD0_B, natf(16));
  // ray_dir_fc.2: 16 -> 19, output rows in local-channel order
  pack_mma(out + mma_off(L_RD2), 2, 1, raw + R_RD2_W, 16, dir_row, nat_col(16));
  pack_rows(out + BIAS_OFF + B_RD2 * 32, raw + R_RD2_B, [](int r, int h) { return loc_ch(r, h); });
  // base_fc.0 shared part: [mean(12 steps) | var(12 steps)] -> 64
  pack_mma(out + mma_off(L_B0S), 6, 2, raw + R_B0_W, 57, nat_row(64), [](int step, int h) {
    int grp = step / 12, s = step % 12;
    int ch = s < 11 ? loc_ch(s, h) : -1;
    return ch >= 0 ? grp * DF + ch : -1;
  });
  pack_rows(out + BIAS_OFF + B_B0_T0 * 32, raw + R_B0_B, [](int r, int h) { return hk(0, r, h); });
  pack_rows(out + BIAS_OFF + B_B0_T1 * 32, raw + R_B0_B, [](int r, int h) { return hk(1, r, h); });
  // base_fc.0 view part: f (12 steps) -> 64
  pack_mma(out + mma_off(L_B0V), 3, 2, raw + R_B0_W, 57, nat_row(64), [](int step, int h) {
    int ch = step < 11 ? loc_ch(step, h) : -1;
    return ch >= 0 ? 2 * DF + ch : -1;
  });
  // base_fc.2: 64 -> 32
  pack_mma(out + mma_off(L_B2), 8, 1, raw + R_B2_W, 64, nat_row(32), nat_col(64));
  pack_rows(out + BIAS_OFF + B_B2 * 32, raw + R_B2_B, natf(32));
  // vis_fc.0: 32 -> 32 ; vis_fc.2 rows 0..31 (x_res) as MFMA, row 32 (vis) as a per-lane dot
  pack_mma(out + mma_off(L_V0), 4, 1, raw + R_V0_W, 32, nat_row(32), nat_col(32));
  pack_rows(out + BIAS_OFF + B_V0 * 32, raw + R_V0_B, natf(32));
  pack_mma(out + mma_off(L_V2), 4, 1, raw + R_V2_W, 32, nat_row(32), nat_col(32));
  pack_rows(out + BIAS_OFF + B_V2 * 32, raw + R_V2_B, natf(32));
  pack_rows(out + DOT_OFF + D_VIS * 32, raw + R_V2_W + 32 * 32, natf(32));
  // vis_fc2.0: 32 -> 32 ; vis_fc2.2: 32 -> 1 as a dot
  pack_mma(out + mma_off(L_W0), 4, 1, raw + R_W0_W, 32, nat_row(32), nat_col(32));
  pack_rows(out + BIAS_OFF + B_W0 * 32, raw + R_W0_B, natf(32));
  pack_rows(out + DOT_OFF + D_VIS2 * 32, raw + R_W2_W, natf(32));
  // rgb_fc.0: [x(32) | vis | rd(4)] -> 16 ; steps 16..18: (vis|rd0), (rd1|rd2), (rd3|-)
  pack_mma(out + mma_off(L_R0), 5, 1, raw + R_R0_W, 37, nat_row(16), [](int step, int h) {
    if (step < 16) return hk(0, step, h);
    if (step == 16) return h ? 33 : 32;
    if (step == 17) return h ? 35 : 34;
    if (step == 18) return h ? -1 : 36;
    return -1;
  });
  pack_rows(out + BIAS_OFF + B_R0 * 32, raw + R_R0_B, natf(16));
  // rgb_fc.2: 16 -> 8 ; rgb_fc.4: 8 -> 1 as a dot over registers 0..3 (feature 4 h + r)
  pack_mma(out + mma_off(L_R2), 2, 1, raw + R_R2_W, 16, nat_row(8), nat_col(16));
  pack_rows(out + BIAS_OFF + B_R2 * 32, raw + R_R2_B, natf(8));
  pack_rows(out + DOT_OFF + D_RGB4 * 32, raw + R_R4_W, [](int r, int h) { return r < 4 ? 4 * h + r : -1; });
  out[SCAL_OFF + 0] = fabsf(raw[R_S]);
  out[SCAL_OFF + 1] = raw[R_V2_B + 32];
  out[SCAL_OFF + 2] = raw[R_W2_B];
  out[SCAL_OFF + 3] = raw[R_R4_B];
  return 0;
}

extern "C" int surf_blend(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_feats,
                          const int* h_hw, int n_level, const float* imgs, int nv, const float* h_intrs,
                          const float* h_w2c, const float* h_c2w, const float* blend_w, float* color,
                          uint8_t* n_valid, void* stream) {
  if (!pts || !h_feats || !h_hw || !imgs || !h_intrs || !h_w2c || !h_c2w || !blend_w || !color) return SURF_E_ARG;
  if (n <= 0 || nv < 2) return SURF_E_ARG;
  if (n_level != 4 || nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;  // d_feature = 16 = 4 levels x 4 channels
  BlendArgs a;
  a.pts = pts; a.mask = mask; a.idx = idx; a.n = n; a.imgs = imgs; a.w = blend_w; a.color = color; a.n_valid = n_valid;
  for (int l = 0; l < 4; ++l) {
    if (!h_feats[l]) return SURF_E_ARG;
    a.feats[l] = h_feats[l];
    a.hw[2 * l] = h_hw[2 * l];
    a.hw[2 * l + 1] = h_hw[2 * l + 1];
  }
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int s = v < nv ? v : 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) a.K[v][r * 3 + c] = h_intrs[s * 16 + r * 4 + c];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) a.w2c[v][r * 4 + c] = h_w2c[s * 16 + r * 4 + c];
    for (int r = 0; r < 3; ++r) a.cpos[v][r] = h_c2w[s * 16 + r * 4 + 3];
  }
  dim3 grid(grid_blocks(n)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (nv - 1) {
    case 1: hipLaunchKernelGGL(blend_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(blend_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(blend_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(blend_kernel<4>, grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL(blend_kernel<5>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(blend_kernel<6>, grid, block, 0, st, a); break;
    case 7: hipLaunchKernelGGL(blend_kernel<7>, grid, block, 0, st, a); break;
    default: return SURF_E_LIMIT;
  }
  return surf_check_launch();
}
