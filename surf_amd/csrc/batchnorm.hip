// BatchNorm over the voxel rows of a sparse tensor with BATCH statistics (train mode of spnn.BatchNorm = nn.BatchNorm1d on
// the feature rows, reg_network.py:14-15,28-29; eval mode is folded into the convolution epilogues, spconv.hip).
//   surf_bn_train_affine: per-channel mean / biased variance of x (n, C) -> scale = gamma / sqrt(var + eps),
//                         shift = beta - mean scale; running_mean / running_var updated as torch does (momentum, unbiased var)
//   surf_bn_relu_apply:   out = relu(x scale + shift) (+ skip)
// Statistics are reduced deterministically in fp64: per-workgroup partials, then a serial finalise (as the FPN's
// InstanceNorm, fpn.hip).  Byte-bound: x is read twice, written once.
#include "common.h"

namespace {

constexpr int BN_BLOCKS = 1024;   // (round 5: 512 left the 5 M-row layers at 1.2 TB/s)

template <int C>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
  constexpr int G = 256 / C;                       // row groups per workgroup
  __shared__ double sh[2][256];
  const int c = threadIdx.x % C, g = threadIdx.x / C;
  double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;            // two independent fp64 chains per thread
  const int64_t step = (int64_t)gridDim.x * G;
  int64_t r = (int64_t)blockIdx.x * G + g;
  for (; r + step < n; r += 2 * step) {
    const double v = (double)x[r * C + c], w = (double)x[(r + step) * C + c];
    s1 += v;
    s2 = fma(v, v, s2);
    t1 += w;
    t2 = fma(w, w, t2);
  }
  if (r < n) {
    const double v = (double)x[r * C + c];
    s1 += v;
    s2 = fma(v, v, s2);
  }
  s1 += t1;
  s2 += t2;
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < G; ++k) { a += sh[0][k * C + threadIdx.x]; b += sh[1][k * C + threadIdx.x]; }
    part[((int64_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// Sum of the per-block partials (blocks, 2, C) of one channel.
// One CHANNEL per workgroup (round 5, second pass: a single workgroup walking blocks x C partials was a chain of L2 round trips -
// 12-13 us per layer at 1,024 partial blocks, 80 such launches a training step): thread t takes the blocks b = t (mod 256), a
// wavefront sum by xor shuffles and the four wavefronts' sums in LDS, all in a fixed order (deterministic).  True for thread 0.
__device__ __forceinline__ bool reduce_partials_channel(const double* __restrict__ part, int blocks, int C, int c, double& s1,
                                                        double& s2) {
  __shared__ double red[2][4];
  double a = 0.0, b2 = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) {
    a += part[((int64_t)b * 2 + 0) * C + c];
    b2 += part[((int64_t)b * 2 + 1) * C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b2 += __shfl_xor(b2, o);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b2; }
  __syncthreads();
  s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  return threadIdx.x == 0;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int blocks, int C, int64_t n, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ batch_stats) {
  double s1, s2;
  const int c = blockIdx.x;
  if (!reduce_partials_channel(part, blocks, C, c, s1, s2)) return;
  const double mean = s1 / (double)n;
  double var = s2 / (double)n - mean * mean;      // biased (what the normalisation uses)
  if (var < 0.0) var = 0.0;
  const float sc = gamma[c] / sqrtf((float)var + eps);
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (batch_stats) {
    batch_stats[c] = (float)mean;
    batch_stats[C + c] = 1.0f / sqrtf((float)var + eps);
  }
  if (running_mean) {
    const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// bf16 (RNE) of a float as its 16 bits: the row shadows the bf16 training policy gathers from (round 6)
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ x, int64_t n4, uint16_t* __restrict__ out16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  reinterpret_cast<u16x4*>(out16)[i] = u16x4{bf16_bits(v[0]), bf16_bits(v[1]), bf16_bits(v[2]), bf16_bits(v[3])};
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t n4, int C4, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ skip,
                                                       float* __restrict__ out, uint16_t* __restrict__ out16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 4-channel group
  if (i >= n4) return;
  const int c = (int)(i % C4) * 4;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
  f32x4 y;
#pragma unroll
  for (int q = 0; q < 4; ++q) y[q] = fmaxf(v[q] * sc[q] + sh[q], 0.f);
  if (skip) y += reinterpret_cast<const f32x4*>(skip)[i];
  reinterpret_cast<f32x4*>(out)[i] = y;
  if (out16) reinterpret_cast<u16x4*>(out16)[i] = u16x4{bf16_bits(y[0]), bf16_bits(y[1]), bf16_bits(y[2]), bf16_bits(y[3])};
}

// ---- backward of y = relu(xhat gamma + beta) (+ skip), xhat = (x - mean) invstd -------------------------------------------
//   zbar = dy [relu_out > 0];  dbeta = sum zbar;  dgamma = sum zbar xhat
//   train (batch statistics):  dx = gamma invstd (zbar - mean(zbar) - xhat mean(zbar xhat));   eval: dx = gamma invstd zbar
template <int C>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             double* __restrict__ part) {
  constexpr int G = 256 / C;
  __shared__ double sh[2][256];
  const int c = threadIdx.x % C, g = threadIdx.x / C;
  const float mu = mean[c], is = invstd[c], sc = scale[c], sh0 = shift[c];
  double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;            // two independent fp64 chains per thread
  const int64_t step = (int64_t)gridDim.x * G;
  int64_t r = (int64_t)blockIdx.x * G + g;
  for (; r + step < n; r += 2 * step) {
    const float xv = x[r * C + c], xw = x[(r + step) * C + c];
    const float zb = xv * sc + sh0 > 0.f ? dy[r * C + c] : 0.f;      // the forward's own pre-activation (bn_apply_kernel)
    const float zc = xw * sc + sh0 > 0.f ? dy[(r + step) * C + c] : 0.f;
    s1 += (double)zb;
    s2 = fma((double)zb, (double)((xv - mu) * is), s2);
    t1 += (double)zc;
    t2 = fma((double)zc, (double)((xw - mu) * is), t2);
  }
  if (r < n) {
    const float xv = x[r * C + c];
    const float zb = xv * sc + sh0 > 0.f ? dy[r * C + c] : 0.f;
    s1 += (double)zb;
    s2 = fma((double)zb, (double)((xv - mu) * is), s2);
  }
  s1 += t1;
  s2 += t2;
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < G; ++k) { a += sh[0][k * C + threadIdx.x]; b += sh[1][k * C + threadIdx.x]; }
    part[((int64_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int blocks, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  double s1, s2;
  const int c = blockIdx.x;
  if (!reduce_partials_channel(part, blocks, C, c, s1, s2)) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n, int C,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, int train,
                                                           float* __restrict__ dx, uint16_t* __restrict__ dx16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  const int c = (int)(i % C);
  const float xv = x[i];
  const float zb = xv * scale[c] + shift[c] > 0.f ? dy[i] : 0.f;
  const float xh = (xv - mean[c]) * invstd[c];
  float v = zb;
  if (train) v = zb - dbeta[c] / (float)n - xh * (dgamma[c] / (float)n);
  const float o = scale[c] * v;                                       // gamma invstd = the forward's scale
  dx[i] = o;
  if (dx16) dx16[i] = bf16_bits(o);
}

// ---- InstanceNorm + ReLU backward for ALL views of an FPN layer in three launches (round 5) ---------------------------------
// InstanceNorm = the BatchNorm above over one view's pixels without affine parameters (scale = rstd, shift = -mean rstd).  Until
// round 5 the FPN backward called surf_bn_relu_backward once per view: 11 layers x 5 views x (3 kernels + ~8 torch helpers that
// sliced the statistics and copied the result back) = ~600 launches of a few microseconds per training step.  Here blockIdx.y is
// the view; stats (N, C, 2) = mean | rstd as surf_inorm_relu wrote them.
template <int C>
__global__ __launch_bounds__(256) void inorm_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n,
                                                                const float* __restrict__ stats, double* __restrict__ part) {
  constexpr int G = 256 / C;
  __shared__ double sh[2][256];
  const int c = threadIdx.x % C, g = threadIdx.x / C, v = blockIdx.y;
  const float mu = stats[((int64_t)v * C + c) * 2], is = stats[((int64_t)v * C + c) * 2 + 1];
  const float* __restrict__ xv = x + (int64_t)v * n * C;
  const float* __restrict__ dv = dy + (int64_t)v * n * C;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * G + g; r < n; r += (int64_t)gridDim.x * G) {
    const float xh = (xv[r * C + c] - mu) * is;
    const float zb = xh > 0.f ? dv[r * C + c] : 0.f;                 // relu(xhat) > 0: the forward's own pre-activation
    s1 += (double)zb;
    s2 += (double)zb * (double)xh;
  }
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < G; ++k) { a += sh[0][k * C + threadIdx.x]; b += sh[1][k * C + threadIdx.x]; }
    double* pv = part + (int64_t)v * gridDim.x * 2 * C;
    pv[((int64_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    pv[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

__global__ __launch_bounds__(256) void inorm_bwd_finalize_kernel(const double* __restrict__ part, int blocks, int C,
                                                                 float* __restrict__ sums /* (N, 2, C) */) {
  double s1, s2;
  const int v = blockIdx.x, c = blockIdx.y;
  if (!reduce_partials_channel(part + (int64_t)v * blocks * 2 * C, blocks, C, c, s1, s2)) return;
  sums[((int64_t)v * 2 + 0) * C + c] = (float)s1;
  sums[((int64_t)v * 2 + 1) * C + c] = (float)s2;
}

__global__ __launch_bounds__(256) void inorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n, int C,
                                                              int N, const float* __restrict__ stats, const float* __restrict__ sums,
                                                              float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * n * C) return;
  const int c = (int)(i % C), v = (int)(i / (n * C));
  const float mu = stats[((int64_t)v * C + c) * 2], is = stats[((int64_t)v * C + c) * 2 + 1];
  const float xh = (x[i] - mu) * is;
  const float zb = xh > 0.f ? dy[i] : 0.f;
  dx[i] = is * (zb - sums[((int64_t)v * 2 + 0) * C + c] / (float)n - xh * (sums[((int64_t)v * 2 + 1) * C + c] / (float)n));
}

}  // namespace

extern "C" int64_t surf_inorm_backward_workspace_bytes(int N, int channels) {
  return (int64_t)N * BN_BLOCKS * 2 * channels * sizeof(double) + (int64_t)N * 2 * channels * sizeof(float);
}

extern "C" int surf_inorm_relu_backward(const float* x, const float* dy, int N, int64_t hw, int channels, const float* stats,
                                        void* workspace, float* dx, void* stream) {
  if (!x || !dy || !stats || !workspace || !dx || N <= 0 || hw <= 0) return SURF_E_ARG;
  if (N > 65535) return SURF_E_LIMIT;
  const int64_t want = (hw * channels + 255) / 256;
  const int blocks = (int)(want < BN_BLOCKS ? want : BN_BLOCKS);
  double* part = (double*)workspace;
  float* sums = (float*)((char*)workspace + (int64_t)N * BN_BLOCKS * 2 * channels * sizeof(double));
  hipStream_t s = (hipStream_t)stream;
  switch (channels) {
    case 8: hipLaunchKernelGGL(inorm_bwd_partial_kernel<8>, dim3(blocks, N), dim3(256), 0, s, x, dy, hw, stats, part); break;
    case 16: hipLaunchKernelGGL(inorm_bwd_partial_kernel<16>, dim3(blocks, N), dim3(256), 0, s, x, dy, hw, stats, part); break;
    case 32: hipLaunchKernelGGL(inorm_bwd_partial_kernel<32>, dim3(blocks, N), dim3(256), 0, s, x, dy, hw, stats, part); break;
    case 64: hipLaunchKernelGGL(inorm_bwd_partial_kernel<64>, dim3(blocks, N), dim3(256), 0, s, x, dy, hw, stats, part); break;
    default: return SURF_E_LIMIT;
  }
  hipLaunchKernelGGL(inorm_bwd_finalize_kernel, dim3(N, channels), dim3(256), 0, s, part, blocks, channels, sums);
  const int64_t total = (int64_t)N * hw * channels;
  hipLaunchKernelGGL(inorm_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, dy, hw, channels, N, stats, sums, dx);
  return surf_check_launch();
}

extern "C" int surf_bn_relu_backward16(const float* x, const float* dy, int64_t n, int channels, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, int train, void* workspace,
                                       float* dgamma, float* dbeta, float* dx, uint16_t* dx16, void* stream) {
  if (!x || !dy || !scale || !shift || !mean || !invstd || !workspace || !dgamma || !dbeta || !dx || n <= 0) return SURF_E_ARG;
  const int64_t want = (n * channels + 255) / 256;
  const int blocks = (int)(want < BN_BLOCKS ? want : BN_BLOCKS);
  double* part = (double*)workspace;
  hipStream_t s = (hipStream_t)stream;
  switch (channels) {
    case 8: hipLaunchKernelGGL(bn_bwd_partial_kernel<8>, dim3(blocks), dim3(256), 0, s, x, dy, n, scale, shift, mean, invstd, part); break;
    case 16: hipLaunchKernelGGL(bn_bwd_partial_kernel<16>, dim3(blocks), dim3(256), 0, s, x, dy, n, scale, shift, mean, invstd, part); break;
    case 32: hipLaunchKernelGGL(bn_bwd_partial_kernel<32>, dim3(blocks), dim3(256), 0, s, x, dy, n, scale, shift, mean, invstd, part); break;
    case 64: hipLaunchKernelGGL(bn_bwd_partial_kernel<64>, dim3(blocks), dim3(256), 0, s, x, dy, n, scale, shift, mean, invstd, part); break;
    default: return SURF_E_LIMIT;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(channels), dim3(256), 0, s, part, blocks, channels, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n * channels + 255) / 256)), dim3(256), 0, s, x, dy, n, channels,
                     scale, shift, mean, invstd, dgamma, dbeta, train, dx, dx16);
  return surf_check_launch();
}

extern "C" int surf_bn_relu_backward(const float* x, const float* dy, int64_t n, int channels, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, int train, void* workspace,
                                     float* dgamma, float* dbeta, float* dx, void* stream) {
  return surf_bn_relu_backward16(x, dy, n, channels, scale, shift, mean, invstd, train, workspace, dgamma, dbeta, dx, nullptr, stream);
}

extern "C" int64_t surf_bn_workspace_bytes(int channels) { return (int64_t)BN_BLOCKS * 2 * channels * sizeof(double); }

extern "C" int surf_bn_train_affine(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                    float* batch_stats, void* workspace, void* stream) {
  if (!x || !gamma || !beta || !scale || !shift || !workspace || n <= 0) return SURF_E_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SURF_E_ARG;
  const int64_t want = (n * channels + 255) / 256;
  const int blocks = (int)(want < BN_BLOCKS ? want : BN_BLOCKS);
  double* part = (double*)workspace;
  hipStream_t s = (hipStream_t)stream;
  switch (channels) {
    case 8: hipLaunchKernelGGL(bn_partial_kernel<8>, dim3(blocks), dim3(256), 0, s, x, n, part); break;
    case 16: hipLaunchKernelGGL(bn_partial_kernel<16>, dim3(blocks), dim3(256), 0, s, x, n, part); break;
    case 32: hipLaunchKernelGGL(bn_partial_kernel<32>, dim3(blocks), dim3(256), 0, s, x, n, part); break;
    case 64: hipLaunchKernelGGL(bn_partial_kernel<64>, dim3(blocks), dim3(256), 0, s, x, n, part); break;
    default: return SURF_E_LIMIT;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(channels), dim3(256), 0, s, part, blocks, channels, n, gamma, beta, eps, momentum,
                     running_mean, running_var, scale, shift, batch_stats);
  return surf_check_launch();
}

extern "C" int surf_bn_relu_apply16(const float* x, int64_t n, int channels, const float* scale, const float* shift,
                                    const float* skip, float* out, uint16_t* out16, void* stream) {
  if (!x || !scale || !shift || !out || n <= 0 || channels < 4 || channels % 4) return SURF_E_ARG;
  const int64_t n4 = n * (channels / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4, channels / 4,
                     scale, shift, skip, out, out16);
  return surf_check_launch();
}

extern "C" int surf_bn_relu_apply(const float* x, int64_t n, int channels, const float* scale, const float* shift,
                                  const float* skip, float* out, void* stream) {
  return surf_bn_relu_apply16(x, n, channels, scale, shift, skip, out, nullptr, stream);
}

extern "C" int surf_rows_to_bf16(const float* x, int64_t n_floats, uint16_t* out16, void* stream) {
  if (!x || !out16 || n_floats <= 0 || (n_floats & 3)) return SURF_E_ARG;
  const int64_t n4 = n_floats / 4;
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4, out16);
  return surf_check_launch();
}
