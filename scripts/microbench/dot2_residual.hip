// Is `v_dot2c_f32_bf16` usable for the residuals of the exact three-way bf16 operand split?
//   residual_lo = a - float(p.lo) = dot2c(acc = a, p, (-1, 0)),  residual_hi = b - float(p.hi) = dot2c(acc = b, p, (0, -1))
// with p = v_cvt_pk_bf16_f32(a, b): ONE VALU instruction per residual instead of two (v_lshlrev / v_and + v_sub).
// Checks bit-exactness against the shift-and-subtract form over random and edge-case inputs and times both forms
// (N_ITER dependent chains per lane).   hipcc --offload-arch=gfx950 -O3 dot2_residual.hip -o dot2_residual && ./dot2_residual
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  asm volatile("" : "+v"(u));
  return u;
}
__device__ __forceinline__ void split_shift(float a, float b, uint32_t (&p)[3]) {
  p[0] = pack2(a, b);
  const float ra = a - __builtin_bit_cast(float, p[0] << 16), rb = b - __builtin_bit_cast(float, p[0] & 0xffff0000u);
  p[1] = pack2(ra, rb);
  p[2] = pack2(ra - __builtin_bit_cast(float, p[1] << 16), rb - __builtin_bit_cast(float, p[1] & 0xffff0000u));
}
__device__ __forceinline__ void split_dot2(float a, float b, uint32_t (&p)[3]) {
  // the packed constants (-1, 0) / (0, -1) in registers the compiler cannot see through: written as bf16x2 literals hipcc
  // turns (-1, 0) into the inline constant -1.0, which the hardware expands to 0xBF800000 = (0, -1) (first run of this file:
  // every residual_lo came out as a - piece.hi)
  uint32_t lo_u = 0x0000bf80u, hi_u = 0xbf800000u;
  asm volatile("" : "+v"(lo_u), "+v"(hi_u));
  const bf16x2 lo = __builtin_bit_cast(bf16x2, lo_u), hi = __builtin_bit_cast(bf16x2, hi_u);
  p[0] = pack2(a, b);
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[0]), lo, a, false);
  const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[0]), hi, b, false);
  p[1] = pack2(ra, rb);
  const float ra2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[1]), lo, ra, false);
  const float rb2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[1]), hi, rb, false);
  p[2] = pack2(ra2, rb2);
}

__global__ void check(const float* x, int n, uint32_t* out_shift, uint32_t* out_dot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  uint32_t p[3], q[3];
  split_shift(x[2 * i], x[2 * i + 1], p);
  split_dot2(x[2 * i], x[2 * i + 1], q);
  for (int k = 0; k < 3; ++k) { out_shift[3 * i + k] = p[k]; out_dot[3 * i + k] = q[k]; }
}

template <int MODE>
__global__ void timing(float* io, int iters) {
  float a = io[threadIdx.x], b = io[threadIdx.x + 64];
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t p[3];
    if (MODE == 0) split_shift(a, b, p); else split_dot2(a, b, p);
    acc ^= p[0] ^ p[1] ^ p[2];
    a = a * 1.0001f + __builtin_bit_cast(float, (acc & 0xff) | 0x3f800000u) * 1e-6f;
    b = b * 0.9999f + 1e-6f;
  }
  io[threadIdx.x] = a + b + (float)acc;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    if (i % 5 == 0) {  // full random bit patterns (incl. denormals, huge), minus NaN / Inf
      if (((u >> 23) & 0xff) == 0xff) u &= 0x7f7fffffu;
      memcpy(&h[i], &u, 4);
    } else {           // the kernels' range: |x| in [2^-20, 2^8)
      const float m = 1.0f + (float)(u & 0x7fffff) / 8388608.0f;
      h[i] = ldexpf(m, (int)((u >> 23) % 28) - 20) * ((u >> 31) ? -1.f : 1.f);
    }
  }
  h[0] = 0.f; h[1] = -0.f; h[2] = 1.0f; h[3] = 0.99609375f; h[4] = 1.00390625f; h[5] = 3.0e38f; h[6] = 1e-38f; h[7] = 1e-45f;
  float* dx; uint32_t *ds, *dd;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, (n / 2) * 12); hipMalloc(&dd, (n / 2) * 12);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  check<<<(n / 2 + 255) / 256, 256>>>(dx, n, ds, dd);
  std::vector<uint32_t> hs(n / 2 * 3), hd(n / 2 * 3);
  hipMemcpy(hs.data(), ds, hs.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0, inexact = 0;
  for (int i = 0; i < n / 2; ++i) {
    bool differ = false;
    for (int k = 0; k < 3; ++k) differ |= hs[3 * i + k] != hd[3 * i + k];
    if (differ) {
      ++bad;
      const float a = h[2 * i], b = h[2 * i + 1];
      if (fabsf(a) > 1e-30f && fabsf(b) > 1e-30f && fabsf(a) < 1e30f && fabsf(b) < 1e30f) {
        if (bad_normal < 5) printf("differs: %a %a  shift %08x %08x %08x  dot %08x %08x %08x\n", a, b, hs[3 * i], hs[3 * i + 1], hs[3 * i + 2], hd[3 * i], hd[3 * i + 1], hd[3 * i + 2]);
        ++bad_normal;
      }
    }
    // exactness of the dot2 split itself: the three pieces must sum back to the input (in double)
    for (int half = 0; half < 2; ++half) {
      const float v = h[2 * i + half];
      if (!(fabsf(v) > 1e-30f && fabsf(v) < 1e30f)) continue;
      double s = 0;
      for (int k = 0; k < 3; ++k) {
        uint32_t w = half ? (hd[3 * i + k] & 0xffff0000u) : (hd[3 * i + k] << 16);
        float f; memcpy(&f, &w, 4);
        s += (double)f;
      }
      if (s != (double)v) ++inexact;
    }
  }
  printf("pairs %d: pieces differing from the shift form: %ld (of which with both inputs in [1e-30, 1e30]: %ld); inputs not reproduced exactly by the dot2 pieces: %ld\n",
         n / 2, bad, bad_normal, inexact);
  float* dio; hipMalloc(&dio, 128 * 4);
  hipMemcpy(dio, h.data() + 8, 128 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) timing<0><<<1, 64>>>(dio, iters); else timing<1><<<1, 64>>>(dio, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ns per pair split (one wave, dependent loop incl. ~4 loop-carried VALU)\n", mode ? "dot2c" : "shift", ms * 1e6 / iters);
  }
  return bad_normal || inexact;
}
