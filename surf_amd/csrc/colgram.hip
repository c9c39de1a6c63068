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

// ---- the same reduction on the matrix cores (round 3) -----------------------------------------------------------------------
// out = A^T X is a GEMM whose contraction runs over the ROWS: per 16 rows one v_mfma_f32_32x32x16_bf16 k-step per (32-column
// tile of A) x (32-column tile of X).  Both operands have the same fragment shape - lane l holds column l & 31 of rows
// 8 (l >> 5) .. + 7 of the step - so a fragment is eight coalesced dword loads (128 contiguous bytes per row and half-wave) and
// needs no transposition.  NP = 3: every fp32 value is split exactly into three bf16 pieces and six products are accumulated in
// fp32 (fp32-equivalent, like the SDF / blend kernels); NP = 1 (train.precision = bf16): one rounded bf16 piece, one product.
// A workgroup sweeps a slab of rows; wave w owns A tile w % MT for the row share w / MT of the slab and all X tiles; partials
// go to the workspace and are summed by colgram_reduce_kernel (deterministic).  The VALU kernel above stays for short inputs.
typedef __bf16 cg_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t cg_u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
struct CgFrag { cg_u32x4 p[NP]; };

// raw loads of one fragment (eight rows of this lane's column) and their conversion are separate steps, so that the loads of
// k-step i + 1 are in flight while k-step i is split and multiplied (a wave runs only ~15 k-steps on a 60 K-row batch: without
// the overlap every step exposed a full memory latency)
__device__ __forceinline__ void cg_load_raw(const float* __restrict__ base, int ld, int64_t row0, int64_t row_end, int col, int ncols,
                                            bool ones_col, int h, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t r = row0 + 8 * h + j;
    float x = 0.f;
    if (r < row_end) {
      if (col < ncols) x = base[r * ld + col];
      else if (ones_col && col == ncols) x = 1.0f;
    }
    v[j] = x;
  }
}
template <int NP>
__device__ __forceinline__ void cg_make_frag(const float (&v)[8], CgFrag<NP>& f) {
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    if (NP == 3) {
      uint32_t p[3];
      surf_split3_bf16(v[2 * pr], v[2 * pr + 1], p);
#pragma unroll
      for (int k = 0; k < 3; ++k) f.p[k % NP][pr] = p[k];
    } else {
      f.p[0][pr] = surf_pack2_bf16(v[2 * pr], v[2 * pr + 1]);
    }
  }
}

template <int NP, int NT>
__global__ __launch_bounds__(256) void colgram_mfma_kernel(const float* __restrict__ A, int ldA, int M, const float* __restrict__ X,
                                                           int ldX, int N, int64_t rows, int64_t rows_per_split, int MT, int with_sum,
                                                           float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int nsplit = 4 / MT;
  const int mt = wave % MT, split = wave / MT;
  if (split >= nsplit) return;                                  // MT = 3: the fourth wave has no tile
  const int NX = N + (with_sum ? 1 : 0);
  const int64_t part = (int64_t)blockIdx.x * nsplit + split;
  const int64_t r0 = part * rows_per_split;
  const int64_t r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float va[8], vx[NT][8];
  cg_load_raw(A, ldA, r0, r1, 32 * mt + c, M, false, h, va);
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (32 * t < NX) cg_load_raw(X, ldX, r0, r1, 32 * t + c, N, with_sum != 0, h, vx[t]);
  for (int64_t rb = r0; rb < r1; rb += 16) {
    CgFrag<NP> fa, fx[NT];
    cg_make_frag<NP>(va, fa);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (32 * t < NX) cg_make_frag<NP>(vx[t], fx[t]);
    if (rb + 16 < r1) {                                         // next step's rows: issued before this step's products
      cg_load_raw(A, ldA, rb + 16, r1, 32 * mt + c, M, false, h, va);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (32 * t < NX) cg_load_raw(X, ldX, rb + 16, r1, 32 * t + c, N, with_sum != 0, h, vx[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (32 * t < NX) {                                        // (uniform)
#define CG_MF(x, y) \
  acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cg_bf16x8, fa.p[x]), __builtin_bit_cast(cg_bf16x8, fx[t].p[y]), acc[t], 0, 0, 0)
        if (NP == 3) {
          CG_MF(2 % NP, 0); CG_MF(0, 2 % NP); CG_MF(1 % NP, 1 % NP); CG_MF(1 % NP, 0); CG_MF(0, 1 % NP); CG_MF(0, 0);
        } else {
          CG_MF(0, 0);
        }
#undef CG_MF
      }
    }
  }
  // accumulator register r of lane (h, c): row m = (r & 3) + 8 (r >> 2) + 4 h of the A tile, column c of the X tile
  float* __restrict__ out = partial + part * (int64_t)M * NX;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = 32 * t + c;
    if (n < NX) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) out[(int64_t)m * NX + n] = acc[t][r];
      }
    }
  }
}

constexpr int64_t CG_MFMA_MIN_ROWS = 4096;    // below this the VALU kernel's launch is as fast
struct CgPlan { int MT, nsplit, blocks; int64_t rows_per_split; };
__host__ inline CgPlan cg_plan(int64_t rows, int M) {
  CgPlan p;
  p.MT = (M + 31) / 32;
  p.nsplit = 4 / p.MT;
  int64_t per = (rows + 256 * p.nsplit - 1) / (256 * p.nsplit);  // <= 256 workgroups
  per = (per + 15) / 16 * 16;
  if (per < 64) per = 64;
  p.rows_per_split = per;
  p.blocks = (int)((rows + per * p.nsplit - 1) / (per * p.nsplit));
  return p;
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
  int64_t need = blocks * M * (N + 1);
  // the MFMA plan is also taken for SHORT inputs when precision == 1 (surf_colgram_p), and its partial count
  // nsplit * ceil(rows / (per * nsplit)) can exceed the VALU slab count there: size for whichever is larger, always
  if (rows > 0 && M <= 128) {
    const CgPlan p = cg_plan(rows, M);
    const int64_t need2 = (int64_t)p.blocks * p.nsplit * M * (N + 1);
    if (need2 > need) need = need2;
  }
  return need;
}

// out (M, N + with_sum) = (accumulate ? out : 0) + A[:, :M]^T [X[:, :N] | 1];  precision 0: fp32-equivalent (VALU for short inputs,
// exact three-way bf16 split on the matrix cores otherwise), 1: operands rounded to bf16, fp32 accumulate (train.precision = bf16)
extern "C" int surf_colgram_p(const float* A, int ldA, int M, const float* X, int ldX, int N, int64_t rows, int with_sum,
                              int accumulate, int precision, float* workspace, float* out, void* stream) {
  if (!A || !X || !workspace || !out || rows <= 0 || M < 1 || N < 1 || ldA < M || ldX < N) return SURF_E_ARG;
  if (precision != 0 && precision != 1) return SURF_E_ARG;
  const int NX = N + (with_sum ? 1 : 0);
  if (M > 128 || NX > 160) return SURF_E_LIMIT;
  hipStream_t st = (hipStream_t)stream;
  const int64_t count = (int64_t)M * NX;
  int64_t blocks;
  if (rows >= CG_MFMA_MIN_ROWS || precision == 1) {
    const CgPlan p = cg_plan(rows, M);
    blocks = (int64_t)p.blocks * p.nsplit;
    const int NT = (NX + 31) / 32;
#define CG_LAUNCH(NP, NTV)                                                                                                   \
  hipLaunchKernelGGL((colgram_mfma_kernel<NP, NTV>), dim3((unsigned)p.blocks), dim3(256), 0, st, A, ldA, M, X, ldX, N, rows, \
                     p.rows_per_split, p.MT, with_sum, workspace)
    if (precision == 0) {
      if (NT <= 2) CG_LAUNCH(3, 2); else if (NT <= 3) CG_LAUNCH(3, 3); else CG_LAUNCH(3, 5);
    } else {
      if (NT <= 2) CG_LAUNCH(1, 2); else if (NT <= 3) CG_LAUNCH(1, 3); else CG_LAUNCH(1, 5);
    }
#undef CG_LAUNCH
  } else {
    const int64_t ROWS_PER_BLOCK = slab_rows(rows);
    blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (blocks > 0x7fffffff) return SURF_E_LIMIT;
    if (M <= 64 && NX <= 64)
      hipLaunchKernelGGL((colgram_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, M, X, ldX, N, rows, ROWS_PER_BLOCK, with_sum, workspace);
    else
      hipLaunchKernelGGL((colgram_kernel<8, 10>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, M, X, ldX, N, rows, ROWS_PER_BLOCK, with_sum, workspace);
  }
  hipLaunchKernelGGL(colgram_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, st, workspace, (int)blocks, count, out, accumulate);
  return surf_check_launch();
}

extern "C" int surf_colgram(const float* A, int ldA, int M, const float* X, int ldX, int N, int64_t rows, int with_sum,
                            int accumulate, float* workspace, float* out, void* stream) {
  return surf_colgram_p(A, ldA, M, X, ldX, N, rows, with_sum, accumulate, 0, workspace, out, stream);
}
