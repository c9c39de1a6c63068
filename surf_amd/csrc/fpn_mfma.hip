// K1m: the wide layers (C_in >= 16, coarse levels) of the FPN feature extractor on the matrix cores - forward, input gradient and weight gradient.
// Same functions as conv3x3_kernel / deconv3x3_s2_kernel / wgrad_kernel of fpn.hip (FeatureNetwork.forward
// feature_network.py:158-178; Conv2d :6-25, Deconv2d :57-75, and their autograd under loss.backward(), runner.py:163).
//
// Why (round 6): the direct fp32 VALU convolutions run at 8 - 14 TFLOP/s with one thread per output pixel; irrelevant to
// inference (1.6 ms of 285) but 8.7 ms of the 76 ms training step (two FPN forwards - the matching features come from a frozen
// copy, surf.py:37 - plus input and weight gradients).  north_star names these blocks as MFMA work.
//
// Arithmetic: the bf16x3 scheme of the SDF / blend / sparse-convolution kernels - both operands split exactly into three bf16
// pieces, six v_mfma_f32_32x32x16_bf16 products accumulated in fp32: fp32-equivalent results (NPROD = 6) - or, under the bf16
// training policy (train_precision = bf16), both operands rounded to bf16 and ONE product (NPROD = 1).
//
// conv3x3_mfma_kernel (forward, and the input gradients through flipped / transposed kernels): output-stationary implicit GEMM.
//   One wavefront owns 32 consecutive output pixels and all C_out channels; for each of the 9 taps
//       D[c_out][pixel] += W_tap^T[c_out][c_in] * X[c_in][pixel + tap]
//   X rows come straight from the NHWC map (two 16-byte loads per 16-channel k-step and lane: the 3 x 3 neighbourhoods of a row of
//   pixels overlap in L1 / L2), are split in registers and are the B operand; W_tap (fp32, [tap][c_in][c_out] as the VALU kernels
//   take it) is split by the workgroup while it is staged into LDS in A-fragment order - no separate weight packing pass, the
//   training step changes the weights every iteration.  MODE: stride 1, stride 2, or the stride-2 transposed convolution
//   (output pixel (y, x) reads input ((y + 1 - ky) / 2, (x + 1 - kx) / 2) where both are even).
// wgrad_mfma_kernel: dW[tap][cb][cs] = sum over pixels of big[pixel * S + tap][cb] * small[pixel][cs] - a GEMM whose K dimension
//   is the pixel index.  One wavefront owns one tap and a group of WG_ROWS rows of the small map: both operands are read
//   TRANSPOSED (lane = channel, eight consecutive pixels of the row per lane: 128-byte coalesced segments across the lanes),
//   split in registers, and the 32 x 32 tiles of the tap's (cb, cs) slice accumulate over the rows; per-group partial sums go to
//   the workspace and are added in a fixed order by wgrad_finalize_kernel (fpn.hip): deterministic, no atomics.
//
// MFMA operand layout (32x32x16): A lane l: row l%32, k = 8(l/32)..+8;  B lane l: column l%32, k = 8(l/32)..+8;
// D register r of lane l: row (r&3) + 8(r>>2) + 4(l/32), column l%32.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { FM_S1 = 0, FM_S2 = 1, FM_UP = 2 };

struct ConvArgs {
  const float* in;
  const float* w;      // [9][CIN][COUT] fp32
  float* out;
  int N, H, W, Ho, Wo;
};

#define SURF_FM_MFMA(x, y, c) \
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)

// one pair (a, b) -> NP packed pieces
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&q)[3]) {
  if constexpr (NP == 1) {
    q[0] = surf_pack2_bf16(a, b);
  } else {
    surf_split3_bf16(a, b, q);
  }
}

template <int CIN, int COUT, int MODE, int NPROD>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
  constexpr int KS = CIN / 16, MT = (COUT + 31) / 32, NP = NPROD == 1 ? 1 : 3, FRAGS = KS * MT * NP;
  static_assert(CIN % 16 == 0 && COUT % 4 == 0, "channel counts");
  __shared__ u32x4 wlds[2][FRAGS * 64];                       // double-buffered W_tap pieces (64 -> 64: 2 x 24 KB)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int64_t total = (int64_t)a.N * a.Ho * a.Wo;
  const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool live = i < total;
  const int64_t ic = live ? i : total - 1;
  const int xo = (int)(ic % a.Wo), yo = (int)((ic / a.Wo) % a.Ho), n = (int)(ic / ((int64_t)a.Wo * a.Ho));
  const float* __restrict__ inn = a.in + (int64_t)n * a.H * a.W * CIN + 8 * h;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  auto stage = [&](int tap, int buf) {                        // split W_tap into A fragments: slot = (ks, m, lane')
#pragma unroll
    for (int slot = threadIdx.x; slot < KS * MT * 64; slot += 256) {
      const int sl = slot & 63, m = (slot >> 6) % MT, ks = (slot >> 6) / MT;
      const int co = 32 * m + (sl & 31), ci0 = 16 * ks + 8 * (sl >> 5);
      const float* __restrict__ ws = a.w + ((int64_t)tap * CIN + ci0) * COUT + co;
      u32x4 p[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float wa = co < COUT ? ws[(2 * j) * COUT] : 0.f, wb = co < COUT ? ws[(2 * j + 1) * COUT] : 0.f;
        uint32_t q[3];
        split_pair<NP>(wa, wb, q);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e][j] = q[e];
      }
#pragma unroll
      for (int e = 0; e < NP; ++e) wlds[buf][((ks * MT + m) * NP + e) * 64 + sl] = p[e];
    }
  };
  auto source = [&](int tap) -> int64_t {                     // float offset of the input pixel feeding tap, or -1
    const int ky = tap / 3, kx = tap % 3;
    int yi, xi;
    bool ok = live;
    if (MODE == FM_S1) {
      yi = yo + ky - 1; xi = xo + kx - 1;
    } else if (MODE == FM_S2) {
      yi = 2 * yo + ky - 1; xi = 2 * xo + kx - 1;
    } else {
      const int ty = yo + 1 - ky, tx = xo + 1 - kx;
      ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & 1) == 0;
      yi = ty >> 1; xi = tx >> 1;
    }
    ok = ok && yi >= 0 && yi < a.H && xi >= 0 && xi < a.W;
    return ok ? ((int64_t)yi * a.W + xi) * CIN : -1;
  };

  stage(0, 0);
  int64_t off = source(0);
  for (int tap = 0; tap < 9; ++tap) {
    f32x4 xv[KS][2];
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(inn + (off < 0 ? 0 : off));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (off >= 0) { xv[ks][0] = src[4 * ks]; xv[ks][1] = src[4 * ks + 1]; }
      else { xv[ks][0] = f32x4{0.f, 0.f, 0.f, 0.f}; xv[ks][1] = xv[ks][0]; }
    }
    const bool any = __ballot(off >= 0) != 0ull;
    const int64_t next_off = tap + 1 < 9 ? source(tap + 1) : -1;
    __syncthreads();                                          // W_tap landed in wlds[tap & 1]; the other buffer is free
    if (tap + 1 < 9) stage(tap + 1, (tap + 1) & 1);
    if (any) {
      const u32x4* __restrict__ wl = wlds[tap & 1] + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 b[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t q[3];
          split_pair<NP>(xv[ks][j >> 1][2 * (j & 1)], xv[ks][j >> 1][2 * (j & 1) + 1], q);
#pragma unroll
          for (int e = 0; e < NP; ++e) b[e][j] = q[e];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const u32x4* __restrict__ wf = wl + (ks * MT + m) * NP * 64;
          if constexpr (NPROD == 1) {
            const u32x4 a0 = wf[0];
            SURF_FM_MFMA(a0, b[0], acc[m]);
          } else {
            const u32x4 a0 = wf[0], a1 = wf[64], a2 = wf[128];
            SURF_FM_MFMA(a2, b[0], acc[m]);  // smallest terms first
            SURF_FM_MFMA(a0, b[2], acc[m]);
            SURF_FM_MFMA(a1, b[1], acc[m]);
            SURF_FM_MFMA(a1, b[0], acc[m]);
            SURF_FM_MFMA(a0, b[1], acc[m]);
            SURF_FM_MFMA(a0, b[0], acc[m]);
          }
        }
      }
    }
    off = next_off;
  }
  if (!live) return;
  // register 4 g + q of tile m = channel 32 m + 8 g + 4 h + q
  float* __restrict__ dst = a.out + i * COUT;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = 32 * m + 8 * g + 4 * h;
      if (co >= COUT) continue;
      const f32x4 v = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
      *reinterpret_cast<f32x4*>(dst + co) = v;
    }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------------
constexpr int WG_ROWS = 4;     // rows of the small map per wavefront task (one partial per (row group, tap))

struct WgmArgs {
  const float* big;     // (N, Hb, Wb, CB), Hb = Hs S
  const float* small;   // (N, Hs, Ws, CS)
  float* part;          // (N ceil(Hs / WG_ROWS), 9, CB, CS)
  int N, Hs, Ws, Hb, Wb;
};

template <int CB, int CS, int STRIDE, int NPROD>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgmArgs a) {
  constexpr int MB = (CB + 31) / 32, MS = (CS + 31) / 32, NP = NPROD == 1 ? 1 : 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
  const int groups = (a.Hs + WG_ROWS - 1) / WG_ROWS;
  const int64_t task = (int64_t)blockIdx.x * 4 + wave;            // (n, row group, tap), tap fastest: the 9 taps of a group share L2 lines
  if (task >= (int64_t)a.N * groups * 9) return;
  const int tap = (int)(task % 9), grp = (int)((task / 9) % groups), n = (int)(task / (9 * groups));
  const int ky = tap / 3, kx = tap % 3;
  f32x16 acc[MB][MS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][ms][r] = 0.f;

  for (int ys = grp * WG_ROWS; ys < min((grp + 1) * WG_ROWS, a.Hs); ++ys) {
    const int yb = ys * STRIDE + ky - 1;
    if (yb < 0 || yb >= a.Hb) continue;                            // wave-uniform: the whole row multiplies zero padding
    const float* __restrict__ brow = a.big + ((int64_t)n * a.Hb + yb) * a.Wb * CB;
    const float* __restrict__ srow = a.small + ((int64_t)n * a.Hs + ys) * a.Ws * CS;
    for (int x0 = 0; x0 < a.Ws; x0 += 16) {                        // k-step: 16 pixels of the row, this lane's eight: x0 + 8 h ..
      u32x4 pa[MB][3], pb[MS][3];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const int cs = 32 * ms + c;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int xs = x0 + 8 * h + t;
          v[t] = (cs < CS && xs < a.Ws) ? srow[(int64_t)xs * CS + cs] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t q[3];
          split_pair<NP>(v[2 * j], v[2 * j + 1], q);
#pragma unroll
          for (int e = 0; e < NP; ++e) pb[ms][e][j] = q[e];
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int cb = 32 * mb + c;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int xs = x0 + 8 * h + t;
          const int xb = xs * STRIDE + kx - 1;
          v[t] = (cb < CB && xs < a.Ws && xb >= 0 && xb < a.Wb) ? brow[(int64_t)xb * CB + cb] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t q[3];
          split_pair<NP>(v[2 * j], v[2 * j + 1], q);
#pragma unroll
          for (int e = 0; e < NP; ++e) pa[mb][e][j] = q[e];
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          if constexpr (NPROD == 1) {
            SURF_FM_MFMA(pa[mb][0], pb[ms][0], acc[mb][ms]);
          } else {
            SURF_FM_MFMA(pa[mb][2], pb[ms][0], acc[mb][ms]);
            SURF_FM_MFMA(pa[mb][0], pb[ms][2], acc[mb][ms]);
            SURF_FM_MFMA(pa[mb][1], pb[ms][1], acc[mb][ms]);
            SURF_FM_MFMA(pa[mb][1], pb[ms][0], acc[mb][ms]);
            SURF_FM_MFMA(pa[mb][0], pb[ms][1], acc[mb][ms]);
            SURF_FM_MFMA(pa[mb][0], pb[ms][0], acc[mb][ms]);
          }
        }
    }
  }
  // D register r of lane l: row (cb) 32 mb + (r & 3) + 8 (r >> 2) + 4 h, column (cs) 32 ms + c
  float* __restrict__ dst = a.part + (((int64_t)n * groups + grp) * 9 + tap) * CB * CS;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cb = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * h, cs = 32 * ms + c;
        if (cb < CB && cs < CS) dst[cb * CS + cs] = acc[mb][ms][r];
      }
}

}  // namespace

// ---- dispatch (called from fpn.hip's C-ABI entry points; 1 = launched, 0 = no matrix-core kernel for this shape) ---------------
#define FM_CONV(CI, CO, MD)                                                                                              \
  if (cin == CI && cout == CO && mode == MD) {                                                                          \
    if (bf16_operands)                                                                                                  \
      hipLaunchKernelGGL((conv3x3_mfma_kernel<CI, CO, MD, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);             \
    else                                                                                                                \
      hipLaunchKernelGGL((conv3x3_mfma_kernel<CI, CO, MD, 6>), dim3((unsigned)blocks), dim3(256), 0, st, a);             \
    return 1;                                                                                                           \
  }

// mode: 0 = stride 1, 1 = stride 2, 2 = transposed stride 2.  (H, W) = the INPUT map.
int surf_fpn_conv_mfma(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int mode, float* out,
                       int bf16_operands, hipStream_t st) {
  ConvArgs a;
  a.in = in; a.w = weight; a.out = out; a.N = N; a.H = H; a.W = W;
  a.Ho = mode == FM_S2 ? H / 2 : (mode == FM_UP ? 2 * H : H);
  a.Wo = mode == FM_S2 ? W / 2 : (mode == FM_UP ? 2 * W : W);
  const int64_t blocks = ((int64_t)N * a.Ho * a.Wo + 127) / 128;
  if (blocks > 0x7fffffff) return 0;
  // forward: encoder (stride 1 / 2), heads (-> 4), decoder (transposed).  Input gradients reuse them through flipped / transposed
  // kernels: stride 1 <- stride 1, transposed <- stride 2 (16 -> 32 is the gradient of the 32 -> 16 decoder layer), stride 2 <-
  // transposed.  C_in < 16 (the first level, the heads' input gradients) stays on the VALU kernels of fpn.hip.
  // Measured per call at the bench shape, 5 views (profiles/r06_fpn_ab.txt): 64 -> 64 305 -> 45 us, 32 -> 64 s2 156 -> 26,
  // 64 -> 32 transposed 120 -> 39, 32 -> 32 89 -> 36, 16 -> 32 s2 51 -> 26, 32 -> 16 transposed 66 -> 59.  NOT dispatched here
  // because the direct kernels win on the pixel-heavy fine levels, where a 32-row tile is mostly padding and the launch is bound by
  // its gathers: 16 -> 16 (47 vs 56 us), the 4-channel heads (16 -> 4: 28 vs 55, 32 -> 4: 28 vs 33, 64 -> 4: 20 vs 33) and the
  // 16 -> 8 transposed layer (41 vs 121).
  FM_CONV(32, 32, FM_S1) FM_CONV(64, 64, FM_S1)
  FM_CONV(16, 32, FM_S2) FM_CONV(32, 64, FM_S2)
  FM_CONV(64, 32, FM_UP) FM_CONV(32, 16, FM_UP)
  return 0;
}
#undef FM_CONV

int64_t surf_fpn_wgrad_mfma_chunks(int N, int Hs) { return (int64_t)N * ((Hs + WG_ROWS - 1) / WG_ROWS); }

#define FM_WGRAD(B, S, ST)                                                                                                \
  if (cb == B && cs == S && stride == ST) {                                                                              \
    if (bf16_operands)                                                                                                   \
      hipLaunchKernelGGL((wgrad_mfma_kernel<B, S, ST, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);                  \
    else                                                                                                                 \
      hipLaunchKernelGGL((wgrad_mfma_kernel<B, S, ST, 6>), dim3((unsigned)blocks), dim3(256), 0, st, a);                  \
    return 1;                                                                                                            \
  }

// part: (surf_fpn_wgrad_mfma_chunks(N, Hs), 9, cb, cs) floats; the caller sums the chunks (wgrad_finalize_kernel).
int surf_fpn_wgrad_mfma(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride, float* part,
                        int bf16_operands, int thin_too, hipStream_t st) {
  WgmArgs a;
  a.big = big; a.small = small; a.part = part; a.N = N; a.Hs = Hs; a.Ws = Ws; a.Hb = Hs * stride; a.Wb = Ws * stride;
  const int64_t tasks = surf_fpn_wgrad_mfma_chunks(N, Hs) * 9;
  const int64_t blocks = (tasks + 3) / 4;
  if (blocks > 0x7fffffff) return 0;
  // per call: 64 x 64 298 -> 77 us, 32 x 64 s2 250 -> 63, 16 x 32 s2 228 -> 91, 32 x 32 160 -> 82, 16 x 16 272 -> 213,
  // 64 x 4 110 -> 56; the VALU kernel keeps 16 x 4 (113 vs 208) and 32 x 4 (66 vs 80)
  FM_WGRAD(16, 16, 1) FM_WGRAD(32, 32, 1) FM_WGRAD(64, 64, 1) FM_WGRAD(64, 4, 1)
  FM_WGRAD(16, 32, 2) FM_WGRAD(32, 64, 2)
  if (thin_too) {      // the other pairs (A/B switch SURF_FPN_WGRAD_THIN_MFMA: the training step measured 0.5 ms slower with them)
    FM_WGRAD(8, 16, 2) FM_WGRAD(8, 8, 1) FM_WGRAD(8, 4, 1) FM_WGRAD(4, 8, 1) FM_WGRAD(16, 4, 1) FM_WGRAD(32, 4, 1)
  }
  return 0;
}
#undef FM_WGRAD
