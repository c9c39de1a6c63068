// K1: FPN feature extractor -- 3x3 convolutions / stride-2 transposed convolutions in NHWC, InstanceNorm
// statistics, normalise + ReLU (+ skip add), 4-channel heads that write texel4 maps directly.
//
// Restates FeatureNetwork.forward  feature_network.py:158-178  (Conv2d :6-25, Deconv2d :57-75).
//
// The FPN is ~4.4 GFLOP per view with 8..64 channels: far too thin for MFMA tiles to matter and bound by
// activation traffic, so the kernels are plain NHWC direct convolutions: one thread = one output pixel and all
// output channels in registers, 16-byte activation loads, wave-uniform weights through scalar loads.
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
constexpr int ST_PIX = 4096;  // pixels per partial block

__global__ __launch_bounds__(256) void inorm_partial_kernel(const float* __restrict__ x, int HW, int C, int nblk,
                                                            double* __restrict__ part /* (N, nblk, C, 2) */) {
  extern __shared__ double s_red[];  // (256/64) * C * 2
  const int n = blockIdx.y, b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = b * ST_PIX;
  const int p1 = min(p0 + ST_PIX, HW);
  for (int c = 0; c < C; ++c) {
    double s = 0.0, s2 = 0.0;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
      const double v = (double)x[((int64_t)n * HW + p) * C + c];
      s += v;
      s2 += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) {
      s_red[(wave * C + c) * 2 + 0] = s;
      s_red[(wave * C + c) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, s2 = 0.0;
    for (int w = 0; w < 4; ++w) {
      s += s_red[(w * C + c) * 2 + 0];
      s2 += s_red[(w * C + c) * 2 + 1];
    }
    part[(((int64_t)n * nblk + b) * C + c) * 2 + 0] = s;
    part[(((int64_t)n * nblk + b) * C + c) * 2 + 1] = s2;
  }
}

__global__ void inorm_finalize_kernel(const double* __restrict__ part, int N, int nblk, int C, int HW,
                                      float* __restrict__ stats /* (N, C, 2): mean, rstd */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  const int n = t / C, c = t % C;
  double s = 0.0, s2 = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += part[(((int64_t)n * nblk + b) * C + c) * 2 + 0];
    s2 += part[(((int64_t)n * nblk + b) * C + c) * 2 + 1];
  }
  const double mean = s / HW;
  double var = s2 / HW - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[t * 2 + 0] = (float)mean;
  stats[t * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// y = relu((x - mean) * rstd) (+ skip), in place on x
__global__ __launch_bounds__(256) void inorm_relu_kernel(float* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ skip, int N, int HW, int C) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one f32x4 per thread
  const int c4n = C / 4;
  if (t >= (int64_t)N * HW * c4n) return;
  const int c4 = (int)(t % c4n);
  const int n = (int)(t / ((int64_t)HW * c4n));
  f32x4 v = reinterpret_cast<f32x4*>(x)[t];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c4 * 4 + q;
    const float mean = stats[(n * C + c) * 2 + 0], rstd = stats[(n * C + c) * 2 + 1];
    v[q] = fmaxf((v[q] - mean) * rstd, 0.f);
  }
  if (skip) v += reinterpret_cast<const f32x4*>(skip)[t];
  reinterpret_cast<f32x4*>(x)[t] = v;
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

extern "C" int surf_conv3x3(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int stride,
                            float* out, void* stream) {
  if (!in || !weight || !out || N <= 0 || H <= 0 || W <= 0) return SURF_E_ARG;
  if (stride == 2 && ((H | W) & 1)) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
  CONV_CASE(4, 8, 1) CONV_CASE(8, 8, 1) CONV_CASE(8, 16, 2) CONV_CASE(16, 16, 1) CONV_CASE(16, 32, 2) CONV_CASE(32, 32, 1)
  CONV_CASE(32, 64, 2) CONV_CASE(64, 64, 1) CONV_CASE(8, 4, 1) CONV_CASE(16, 4, 1) CONV_CASE(32, 4, 1) CONV_CASE(64, 4, 1)
  return SURF_E_LIMIT;
}

extern "C" int surf_deconv3x3_s2(const float* in, const float* weight, int N, int H, int W, int cin, int cout, float* out,
                                 void* stream) {
  if (!in || !weight || !out || N <= 0 || H <= 0 || W <= 0) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  DECONV_CASE(64, 32) DECONV_CASE(32, 16) DECONV_CASE(16, 8)
  return SURF_E_LIMIT;
}

extern "C" int64_t surf_inorm_workspace_doubles(int N, int H, int W, int C) {
  const int nblk = (H * W + ST_PIX - 1) / ST_PIX;
  return (int64_t)N * nblk * C * 2;
}

extern "C" int surf_inorm_relu(float* x, int N, int H, int W, int C, const float* skip, double* workspace, float* stats,
                               void* stream) {
  if (!x || !workspace || !stats || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || C > 64) return SURF_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int nblk = (HW + ST_PIX - 1) / ST_PIX;
  hipLaunchKernelGGL(inorm_partial_kernel, dim3(nblk, N), dim3(256), 4 * C * 2 * sizeof(double), st, x, HW, C, nblk, workspace);
  hipLaunchKernelGGL(inorm_finalize_kernel, grid1d(N * C, 64), dim3(64), 0, st, workspace, N, nblk, C, HW, stats);
  hipLaunchKernelGGL(inorm_relu_kernel, grid1d((int64_t)N * HW * (C / 4), 256), dim3(256), 0, st, x, stats, skip, N, HW, C);
  return surf_check_launch();
}
