// K1: FPN feature extractor -- 3x3 convolutions / stride-2 transposed convolutions in NHWC, InstanceNorm
// statistics, normalise + ReLU (+ skip add), 4-channel heads that write texel4 maps directly.
//
// Restates FeatureNetwork.forward  feature_network.py:158-178  (Conv2d :6-25, Deconv2d :57-75).
//
// The FPN is ~4.4 GFLOP per view with 8..64 channels.  This file holds the plain NHWC direct convolutions: one thread = one
// output pixel and all output channels in registers, 16-byte activation loads, wave-uniform weights through scalar loads - what
// the first level (3 / 8 channels, bound by activation traffic) and the heads' input gradients run on.  Round 6: the layers with
// C_in >= 16 (forward, input gradient, weight gradient) run on the matrix cores, fpn_mfma.hip; the entry points below dispatch
// (SURF_FPN_VALU=1 in the environment keeps everything on the kernels of this file: the A/B switch of the tests).
// InstanceNorm statistics are reduced deterministically in fp64 (per-block partials, then a serial finalise).
#include "common.h"

namespace {

// weights repacked by the host to [ky][kx][ci][co]
template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ W, int N, int H,
                                                      int Wd, int Ho, int Wo, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), n = (int)(i / ((int64_t)Wo * Ho));
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int yi = yo * STRIDE + ky - 1;
    if (yi < 0 || yi >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int xi = xo * STRIDE + kx - 1;
      if (xi < 0 || xi >= Wd) continue;
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(in + (((int64_t)n * H + yi) * Wd + xi) * CIN);
      const float* __restrict__ Wk = W + (ky * 3 + kx) * CIN * COUT;
      for (int c4 = 0; c4 < CIN / 4; ++c4) {
        const f32x4 xv = src[c4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* __restrict__ Wr = Wk + (c4 * 4 + q) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv[q], Wr[co], acc[co]);
        }
      }
    }
  }
  f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(out + i * COUT);
#pragma unroll
  for (int c4 = 0; c4 < COUT / 4; ++c4) {
    f32x4 v = {acc[c4 * 4], acc[c4 * 4 + 1], acc[c4 * 4 + 2], acc[c4 * 4 + 3]};
    dst[c4] = v;
  }
}

// ConvTranspose2d(k3, s2, p1, output_padding 1): out (2H, 2W); out[y][x] += in[(y+1-ky)/2][(x+1-kx)/2] * W[ky][kx]
// for even (y+1-ky), (x+1-kx).  Weights repacked to [ky][kx][ci][co] from (Cin, Cout, 3, 3).
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void deconv3x3_s2_kernel(const float* __restrict__ in, const float* __restrict__ W, int N,
                                                           int H, int Wd, float* __restrict__ out) {
  const int Ho = 2 * H, Wo = 2 * Wd;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), n = (int)(i / ((int64_t)Wo * Ho));
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = yo + 1 - ky;
    if (ty < 0 || (ty & 1)) continue;
    const int yi = ty >> 1;
    if (yi >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = xo + 1 - kx;
      if (tx < 0 || (tx & 1)) continue;
      const int xi = tx >> 1;
      if (xi >= Wd) continue;
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(in + (((int64_t)n * H + yi) * Wd + xi) * CIN);
      const float* __restrict__ Wk = W + (ky * 3 + kx) * CIN * COUT;
      for (int c4 = 0; c4 < CIN / 4; ++c4) {
        const f32x4 xv = src[c4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* __restrict__ Wr = Wk + (c4 * 4 + q) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv[q], Wr[co], acc[co]);
        }
      }
    }
  }
  f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(out + i * COUT);
#pragma unroll
  for (int c4 = 0; c4 < COUT / 4; ++c4) {
    f32x4 v = {acc[c4 * 4], acc[c4 * 4 + 1], acc[c4 * 4 + 2], acc[c4 * 4 + 3]};
    dst[c4] = v;
  }
}

// ---- InstanceNorm2d(affine=False, eps=1e-5): per-(n,c) mean and biased variance over H*W --------------------
constexpr int ST_PIX = 1024;  // pixels per partial block (round 5: 4096 left the finest level with 2 workgroups per CU)

// Round 5: thread (pixel group g = tid / C, channel c = tid % C) - the 256 lanes of a step read 256 CONSECUTIVE floats of the
// NHWC rows (the first version gave every thread a pixel and walked the channels in an outer loop: lanes 4 C bytes apart, every
// line fetched C times: 0.76 TB/s at the finest level, 2.1 ms per training step and a fifth of the FPN's forward).
__global__ __launch_bounds__(256) void inorm_partial_kernel(const float* __restrict__ x, int HW, int C, int nblk,
                                                            double* __restrict__ part /* (N, nblk, C, 2) */) {
  __shared__ double s_red[2][256];
  const int n = blockIdx.y, b = blockIdx.x;
  const int G = 256 / C, c = threadIdx.x % C, g = threadIdx.x / C;       // C in {4, 8, 16, 32, 64}: divides 256 (checked at launch)
  const int p0 = b * ST_PIX;
  const int p1 = min(p0 + ST_PIX, HW);
  // four independent accumulator pairs: the fp64 add / fma chains are what a thread waits for, not the loads
  double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
  const float* __restrict__ xn = x + (int64_t)n * HW * C;
  int p = p0 + g;
  for (; p + 3 * G < p1; p += 4 * G) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double v = (double)xn[(int64_t)(p + u * G) * C + c];
      sa[u] += v;
      sb[u] = fma(v, v, sb[u]);
    }
  }
  for (; p < p1; p += G) {
    const double v = (double)xn[(int64_t)p * C + c];
    sa[0] += v;
    sb[0] = fma(v, v, sb[0]);
  }
  s_red[0][threadIdx.x] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
  s_red[1][threadIdx.x] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0, a2 = 0.0;
    for (int k = 0; k < G; ++k) { a += s_red[0][k * C + threadIdx.x]; a2 += s_red[1][k * C + threadIdx.x]; }
    part[(((int64_t)n * nblk + b) * C + threadIdx.x) * 2 + 0] = a;
    part[(((int64_t)n * nblk + b) * C + threadIdx.x) * 2 + 1] = a2;
  }
}

// one wavefront per (n, c): lanes stride over the partial blocks, wave sum in a fixed order (deterministic)
__global__ __launch_bounds__(64) void inorm_finalize_kernel(const double* __restrict__ part, int N, int nblk, int C, int HW,
                                                            float* __restrict__ stats /* (N, C, 2): mean, rstd */) {
  const int t = blockIdx.x;
  if (t >= N * C) return;
  const int n = t / C, c = t % C;
  double s = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    s += part[(((int64_t)n * nblk + b) * C + c) * 2 + 0];
    s2 += part[(((int64_t)n * nblk + b) * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    s2 += __shfl_xor(s2, o);
  }
  if (threadIdx.x != 0) return;
  const double mean = s / HW;
  double var = s2 / HW - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[t * 2 + 0] = (float)mean;
  stats[t * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// y = relu((x - mean) * rstd) (+ skip); y may be x (in place) or another buffer (train mode keeps x for the backward)
__global__ __launch_bounds__(256) void inorm_relu_kernel(const float* x, float* y, const float* __restrict__ stats,
                                                         const float* __restrict__ skip, int N, int HW, int C) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one f32x4 per thread
  const int c4n = C / 4;
  if (t >= (int64_t)N * HW * c4n) return;
  const int c4 = (int)(t % c4n);
  const int n = (int)(t / ((int64_t)HW * c4n));
  f32x4 v = reinterpret_cast<const f32x4*>(x)[t];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c4 * 4 + q;
    const float mean = stats[(n * C + c) * 2 + 0], rstd = stats[(n * C + c) * 2 + 1];
    v[q] = fmaxf((v[q] - mean) * rstd, 0.f);
  }
  if (skip) v += reinterpret_cast<const f32x4*>(skip)[t];
  reinterpret_cast<f32x4*>(y)[t] = v;
}


// ---- weight gradient of the 3x3 convolutions (train mode) --------------------------------------------------------
// out[tap][cb][cs] = sum over (n, ys, xs) of big[n][ys S + ky - 1][xs S + kx - 1][cb] * small[n][ys][xs][cs] (zero padding).
//   Conv2d (stride S):      big = the layer input, small = d output  -> d W[ky][kx][ci][co]
//   ConvTranspose2d (S = 2): big = d output,       small = the layer input -> d W[ky][kx][co][ci] (the caller transposes)
// One workgroup = one tap x one chunk of small-map pixels: 64 or 256 pixels at a time are staged in LDS (the shifted big rows
// and the small rows), every thread owns 4 x 2 register blocks of the tap's slice (thin layers: one block on a 1/G share of
// the pixels); per-chunk partials, summed by wgrad_finalize_kernel (deterministic, no atomics).
constexpr int WG_CHUNK = 2048;

template <int CB, int CS, int STRIDE>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ big, const float* __restrict__ small, int N, int Hb,
                                                    int Wb, int Hs, int Ws, float* __restrict__ part) {
  constexpr int PIX = (CB + CS <= 48) ? 256 : 64;     // pixels staged per round (LDS: PIX (CB + CS + 6) floats)
  constexpr int NB = (CB / 4) * (CS / 2);             // 4 x 2 register blocks of the tap's (CB, CS) slice
  constexpr int PERB = NB >= 256 ? NB / 256 : 1;
  constexpr int G = NB >= 256 ? 1 : 256 / NB;         // pixel groups sharing one block (thin layers: every thread has work)
  constexpr int XS = CB + 4, DS = CS + 2, O = CB * CS;
  static_assert(NB >= 256 ? NB % 256 == 0 : 256 % NB == 0, "(CB / 4) (CS / 2) must divide or be a multiple of 256");
  __shared__ __attribute__((aligned(16))) float sb[PIX * XS];
  __shared__ __attribute__((aligned(16))) float ss[PIX * DS];
  __shared__ float red[NB >= 256 ? 1 : 256 * 8];
  const int tap = blockIdx.y, ky = tap / 3, kx = tap % 3;
  const int64_t total = (int64_t)N * Hs * Ws;
  const int64_t p_begin = (int64_t)blockIdx.x * WG_CHUNK;
  const int64_t p_end = p_begin + WG_CHUNK < total ? p_begin + WG_CHUNK : total;
  const int g = NB >= 256 ? 0 : threadIdx.x / NB;
  const int b0 = NB >= 256 ? threadIdx.x : threadIdx.x % NB;
  float acc[PERB][4][2];
#pragma unroll
  for (int e = 0; e < PERB; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[e][q][0] = acc[e][q][1] = 0.f;
  for (int64_t p0 = p_begin; p0 < p_end; p0 += PIX) {
    const int np = (int)(p_end - p0 < PIX ? p_end - p0 : PIX);
    __syncthreads();
    for (int e = threadIdx.x; e < PIX * (CB / 4); e += 256) {
      const int pl = e / (CB / 4), c4 = e % (CB / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pl < np) {
        const int64_t p = p0 + pl;
        const int xs = (int)(p % Ws), ys = (int)((p / Ws) % Hs), n = (int)(p / ((int64_t)Ws * Hs));
        const int yb = ys * STRIDE + ky - 1, xb = xs * STRIDE + kx - 1;
        if (yb >= 0 && yb < Hb && xb >= 0 && xb < Wb)
          v = *reinterpret_cast<const f32x4*>(big + (((int64_t)n * Hb + yb) * Wb + xb) * CB + c4 * 4);
      }
      *reinterpret_cast<f32x4*>(sb + pl * XS + c4 * 4) = v;
    }
    for (int e = threadIdx.x; e < PIX * CS; e += 256) {
      const int pl = e / CS, c = e % CS;
      ss[pl * DS + c] = pl < np ? small[(p0 + pl) * CS + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PERB; ++e) {
      const int blk = b0 + 256 * e;
      const int cb0 = (blk / (CS / 2)) * 4, cs0 = (blk % (CS / 2)) * 2;
#pragma unroll 4
      for (int pl = g; pl < PIX; pl += G) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(sb + pl * XS + cb0);
        const float d0 = ss[pl * DS + cs0], d1 = ss[pl * DS + cs0 + 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[e][q][0] = fmaf(xv[q], d0, acc[e][q][0]);
          acc[e][q][1] = fmaf(xv[q], d1, acc[e][q][1]);
        }
      }
    }
  }
  float* dst = part + ((int64_t)blockIdx.x * 9 + tap) * O;
  if (NB >= 256) {
#pragma unroll
    for (int e = 0; e < PERB; ++e) {
      const int blk = b0 + 256 * e;
      const int cb0 = (blk / (CS / 2)) * 4, cs0 = (blk % (CS / 2)) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dst[(cb0 + q) * CS + cs0] = acc[e][q][0];
        dst[(cb0 + q) * CS + cs0 + 1] = acc[e][q][1];
      }
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(q * 2 + 0) * 256 + threadIdx.x] = acc[0][q][0];
      red[(q * 2 + 1) * 256 + threadIdx.x] = acc[0][q][1];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NB * 8; o += 256) {
      const int ent = o / NB, blk = o % NB;
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[ent * 256 + q * NB + blk];
      dst[((blk / (CS / 2)) * 4 + ent / 2) * CS + (blk % (CS / 2)) * 2 + (ent & 1)] = t;
    }
  }
}

// Sum of the per-chunk partials (nchunk, total).  Round 5: a workgroup takes 64 consecutive outputs x 4 chunk groups (thread
// (group = tid / 64, output = tid % 64) adds the chunks b = group (mod 4): coalesced rows, four partial sums per output met in
// LDS in a fixed order: deterministic) - one thread per output walking all 1,125 chunks of the finest level alone took 80-290 us
// on one or two workgroups.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ part, int nchunk, int total, float* __restrict__ out) {
  __shared__ double red[4][64];
  const int il = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  double t0 = 0.0, t1 = 0.0;
  if (i < total) {
    int b = cg;
    for (; b + 4 < nchunk; b += 8) {
      t0 += (double)part[(int64_t)b * total + i];
      t1 += (double)part[(int64_t)(b + 4) * total + i];
    }
    if (b < nchunk) t0 += (double)part[(int64_t)b * total + i];
  }
  red[cg][il] = t0 + t1;
  __syncthreads();
  if (cg == 0 && i < total) out[i] = (float)((red[0][il] + red[1][il]) + (red[2][il] + red[3][il]));
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

#define CONV_CASE(CI, CO, ST)                                                                                          \
  if (cin == CI && cout == CO && stride == ST) {                                                                       \
    hipLaunchKernelGGL((conv3x3_kernel<CI, CO, ST>), grid1d((int64_t)N * Ho * Wo, 256), dim3(256), 0, st, in, weight, N, H, \
                       W, Ho, Wo, out);                                                                                 \
    return surf_check_launch();                                                                                        \
  }
#define DECONV_CASE(CI, CO)                                                                                             \
  if (cin == CI && cout == CO) {                                                                                       \
    hipLaunchKernelGGL((deconv3x3_s2_kernel<CI, CO>), grid1d((int64_t)N * 4 * H * W, 256), dim3(256), 0, st, in, weight, N, \
                       H, W, out);                                                                                      \
    return surf_check_launch();                                                                                        \
  }

// fpn_mfma.hip
int surf_fpn_conv_mfma(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int mode, float* out,
                       int bf16_operands, hipStream_t st);
int64_t surf_fpn_wgrad_mfma_chunks(int N, int Hs);
int surf_fpn_wgrad_mfma(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride, float* part,
                        int bf16_operands, int thin_too, hipStream_t st);
static bool fpn_valu_only() { return getenv("SURF_FPN_VALU") != nullptr; }

extern "C" int surf_conv3x3_p(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int stride,
                              float* out, int precision, void* stream) {
  if (!in || !weight || !out || N <= 0 || H <= 0 || W <= 0) return SURF_E_ARG;
  if ((stride != 1 && stride != 2) || (stride == 2 && ((H | W) & 1)) || precision < 0 || precision > 1) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
  if (!fpn_valu_only() && surf_fpn_conv_mfma(in, weight, N, H, W, cin, cout, stride == 2 ? 1 : 0, out, precision, st))
    return surf_check_launch();
  CONV_CASE(4, 8, 1) CONV_CASE(8, 8, 1) CONV_CASE(8, 16, 2) CONV_CASE(16, 16, 1) CONV_CASE(16, 32, 2) CONV_CASE(32, 32, 1)
  CONV_CASE(32, 64, 2) CONV_CASE(64, 64, 1) CONV_CASE(8, 4, 1) CONV_CASE(16, 4, 1) CONV_CASE(32, 4, 1) CONV_CASE(64, 4, 1)
  CONV_CASE(4, 16, 1) CONV_CASE(4, 32, 1) CONV_CASE(4, 64, 1) /* the heads' input gradients (train mode) */
  return SURF_E_LIMIT;
}

extern "C" int surf_conv3x3(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int stride,
                            float* out, void* stream) {
  return surf_conv3x3_p(in, weight, N, H, W, cin, cout, stride, out, 0, stream);
}

extern "C" int surf_deconv3x3_s2_p(const float* in, const float* weight, int N, int H, int W, int cin, int cout, float* out,
                                   int precision, void* stream) {
  if (!in || !weight || !out || N <= 0 || H <= 0 || W <= 0 || precision < 0 || precision > 1) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!fpn_valu_only() && surf_fpn_conv_mfma(in, weight, N, H, W, cin, cout, 2, out, precision, st)) return surf_check_launch();
  DECONV_CASE(64, 32) DECONV_CASE(32, 16) DECONV_CASE(16, 8)
  return SURF_E_LIMIT;
}

extern "C" int surf_deconv3x3_s2(const float* in, const float* weight, int N, int H, int W, int cin, int cout, float* out,
                                 void* stream) {
  return surf_deconv3x3_s2_p(in, weight, N, H, W, cin, cout, out, 0, stream);
}

extern "C" int64_t surf_inorm_workspace_doubles(int N, int H, int W, int C) {
  const int nblk = (H * W + ST_PIX - 1) / ST_PIX;
  return (int64_t)N * nblk * C * 2;
}

extern "C" int surf_inorm_relu_out(const float* x, int N, int H, int W, int C, const float* skip, double* workspace, float* stats,
                                   float* out, void* stream) {
  if (!x || !out || !workspace || !stats || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || C > 64 || 256 % C) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int nblk = (HW + ST_PIX - 1) / ST_PIX;
  hipLaunchKernelGGL(inorm_partial_kernel, dim3(nblk, N), dim3(256), 0, st, x, HW, C, nblk, workspace);
  hipLaunchKernelGGL(inorm_finalize_kernel, dim3(N * C), dim3(64), 0, st, workspace, N, nblk, C, HW, stats);
  hipLaunchKernelGGL(inorm_relu_kernel, grid1d((int64_t)N * HW * (C / 4), 256), dim3(256), 0, st, x, out, stats, skip, N, HW, C);
  return surf_check_launch();
}

extern "C" int surf_inorm_relu(float* x, int N, int H, int W, int C, const float* skip, double* workspace, float* stats,
                               void* stream) {
  return surf_inorm_relu_out(x, N, H, W, C, skip, workspace, stats, x, stream);
}

extern "C" int64_t surf_conv3x3_wgrad_workspace_floats(int N, int Hs, int Ws, int cb, int cs) {
  const int64_t nchunk = ((int64_t)N * Hs * Ws + WG_CHUNK - 1) / WG_CHUNK;       // VALU kernel: 2,048-pixel chunks
  const int64_t nchunk_m = surf_fpn_wgrad_mfma_chunks(N, Hs);                      // matrix-core kernel: groups of 4 rows
  return (nchunk > nchunk_m ? nchunk : nchunk_m) * 9 * cb * cs;
}

extern "C" int surf_conv3x3_wgrad_p(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride,
                                    float* workspace, float* out, int precision, void* stream) {
  if (!big || !small || !workspace || !out || N <= 0 || Hs <= 0 || Ws <= 0 || precision < 0 || precision > 1) return SURF_E_ARG;
  if (stride != 1 && stride != 2) return SURF_E_ARG;
  const int O = cb * cs;
  hipStream_t st = (hipStream_t)stream;
  if (!fpn_valu_only() &&
      surf_fpn_wgrad_mfma(big, small, N, Hs, Ws, cb, cs, stride, workspace, precision, getenv("SURF_FPN_WGRAD_THIN_MFMA") != nullptr, st)) {
    hipLaunchKernelGGL(wgrad_finalize_kernel, grid1d(9 * O, 64), dim3(256), 0, st, workspace, (int)surf_fpn_wgrad_mfma_chunks(N, Hs),
                       9 * O, out);
    return surf_check_launch();
  }
  const int nchunk = (int)(((int64_t)N * Hs * Ws + WG_CHUNK - 1) / WG_CHUNK);
  const int Hb = Hs * stride, Wb = Ws * stride;
  bool done = false;
#define WGRAD_CASE(B, S, ST)                                                                                                 \
  if (!done && cb == B && cs == S && stride == ST) {                                                                        \
    hipLaunchKernelGGL((wgrad_kernel<B, S, ST>), dim3(nchunk, 9), dim3(256), 0, st, big, small, N, Hb, Wb, Hs, Ws, workspace); \
    done = true;                                                                                                            \
  }
  // Conv2d stride 1: (Cin, Cout) of the encoder and the heads; stride 2: the down convolutions, and the transposed ones as (Cout, Cin)
  WGRAD_CASE(4, 8, 1) WGRAD_CASE(8, 8, 1) WGRAD_CASE(16, 16, 1) WGRAD_CASE(32, 32, 1) WGRAD_CASE(64, 64, 1)
  WGRAD_CASE(8, 4, 1) WGRAD_CASE(16, 4, 1) WGRAD_CASE(32, 4, 1) WGRAD_CASE(64, 4, 1)
  WGRAD_CASE(8, 16, 2) WGRAD_CASE(16, 32, 2) WGRAD_CASE(32, 64, 2)
#undef WGRAD_CASE
  if (!done) return SURF_E_LIMIT;
  hipLaunchKernelGGL(wgrad_finalize_kernel, grid1d(9 * O, 64), dim3(256), 0, st, workspace, nchunk, 9 * O, out);
  return surf_check_launch();
}

extern "C" int surf_conv3x3_wgrad(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride,
                                  float* workspace, float* out, void* stream) {
  return surf_conv3x3_wgrad_p(big, small, N, Hs, Ws, cb, cs, stride, workspace, out, 0, stream);
}
