// Tall-skinny reduction  out[m][n] = sum_r A[r][m] X[r][n]  (+ optionally column N: sum_r A[r][m])  for the weight / bias
// gradients of the backward kernels: r runs over 10^5..10^6 (sample, view) rows, M and N are layer widths (1..160).
// rocBLAS' tiles do not fit that shape (10.7 of 24.8 ms of a finetune step went into these reductions); here every workgroup
// sweeps a slab of rows through LDS and keeps a 16 x 16 grid of TM x TN register micro-tiles, slabs are summed by a second
// tiny pass (deterministic: no atomics).  Plain fp32 FMAs.
#include "common.h"

namespace {

constexpr int RT = 32;     // rows per LDS tile

template <int TM, int TN>
__global__ __launch_bounds__(256) void colgram_kernel(const float* __restrict__ A, int ldA, int M, const float* __restrict__ X,
                                                      int ldX, int N, int64_t rows, int64_t rows_per_block, int with_sum,
                                                      float* __restrict__ partial) {
  __shared__ float As[RT][16 * TM], Xs[RT][16 * TN];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int NX = N + (with_sum ? 1 : 0);
  float acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t rb = r0; rb < r1; rb += RT) {
    // cooperative loads, column index fastest; LDS layout [row][i * 16 + t]: element i of thread t
    for (int e = threadIdx.x; e < RT * 16 * TM; e += 256) {
      const int r = e / (16 * TM), c = e % (16 * TM);       // c = logical column m
      const int64_t rr = rb + r;
      const float v = (rr < r1 && c < M) ? A[rr * ldA + c] : 0.f;
      As[r][(c % TM) * 16 + c / TM] = v;
    }
    for (int e = threadIdx.x; e < RT * 16 * TN; e += 256) {
      const int r = e / (16 * TN), c = e % (16 * TN);
      const int64_t rr = rb + r;
      float v = 0.f;
      if (rr < r1) v = c < N ? X[rr * ldX + c] : (c == N && with_sum ? 1.0f : 0.f);
      Xs[r][(c % TN) * 16 + c / TN] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < RT; ++r) {
      float a[TM], x[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[r][i * 16 + ty];
#pragma unroll
      for (int j = 0; j < TN; ++j) x[j] = Xs[r][j * 16 + tx];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(a[i], x[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* __restrict__ out = partial + (int64_t)blockIdx.x * M * NX;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = ty * TM + i, n = tx * TN + j;
      if (m < M && n < NX) out[m * NX + n] = acc[i][j];
    }
}

// Sum of the slabs' partials: 64 output elements per workgroup, the slabs split over its four wavefronts with four independent
// accumulators each (the first version walked ~1000 slabs serially per element: 217 us per call, 7 ms of a training step).
// Fixed summation order: deterministic.
__global__ __launch_bounds__(256) void colgram_reduce_kernel(const float* __restrict__ partial, int blocks, int64_t count,
                                                             float* __restrict__ out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < count) {
    int b = w;
    for (; b + 12 < blocks; b += 16) {
      s0 += partial[(int64_t)b * count + e];
      s1 += partial[(int64_t)(b + 4) * count + e];
      s2 += partial[(int64_t)(b + 8) * count + e];
      s3 += partial[(int64_t)(b + 12) * count + e];
    }
    for (; b < blocks; b += 4) s0 += partial[(int64_t)b * count + e];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && e < count) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[e] = accumulate ? out[e] + t : t;
  }
}

// slab height: ~1024 workgroups per call (a training batch has 6e4..1e6 rows: 1024-row slabs left most CUs idle)
__host__ inline int64_t slab_rows(int64_t rows) {
  int64_t r = (rows + 1023) / 1024;
  r = (r + RT - 1) / RT * RT;
  return r < 2 * RT ? 2 * RT : r;
}

}  // namespace

extern "C" int64_t surf_colgram_workspace_floats(int64_t rows, int M, int N) {
  const int64_t rpb = slab_rows(rows), blocks = (rows + rpb - 1) / rpb;
  return blocks * M * (N + 1);
}

// out (M, N + with_sum) = (accumulate ? out : 0) + A[:, :M]^T [X[:, :N] | 1]
extern "C" int surf_colgram(const float* A, int ldA, int M, const float* X, int ldX, int N, int64_t rows, int with_sum,
                            int accumulate, float* workspace, float* out, void* stream) {
  if (!A || !X || !workspace || !out || rows <= 0 || M < 1 || N < 1 || ldA < M || ldX < N) return SURF_E_ARG;
  const int NX = N + (with_sum ? 1 : 0);
  const int64_t ROWS_PER_BLOCK = slab_rows(rows);
  const int64_t blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (blocks > 0x7fffffff) return SURF_E_LIMIT;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 64 && NX <= 64)
    hipLaunchKernelGGL((colgram_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, M, X, ldX, N, rows, ROWS_PER_BLOCK, with_sum, workspace);
  else if (M <= 128 && NX <= 160)
    hipLaunchKernelGGL((colgram_kernel<8, 10>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, M, X, ldX, N, rows, ROWS_PER_BLOCK, with_sum, workspace);
  else
    return SURF_E_LIMIT;
  const int64_t count = (int64_t)M * NX;
  hipLaunchKernelGGL(colgram_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, st, workspace, (int)blocks, count, out, accumulate);
  return surf_check_launch();
}
