// Microbenchmark (gfx950), round 4: can the matrix pipe form the RESIDUALS of the exact three-way bf16 split?
// The blend kernel turns every activation tile (32 rows x 32 samples, 16 values a lane) into B fragments
// a = p0 + p1 + p2 (bf16 pieces, RNE).  On the VALU that is, per pair of values: 3 v_cvt_pk_bf16_f32 + 2 x (2 expands + 2
// subtracts) = 11 instructions, and the kernel is VALU-issue bound.  The residual r = a - p0 is an accumulator update
//     D = C - I * P0     (C = the fp32 tile, P0 = the packed first pieces AS the B operand of its own k-step, I = a constant
//                         A operand with a single -1 per row: "row i of the tile is k-slot k of this step")
// i.e. one v_mfma_f32_32x32x16_bf16 per k-step (16 rows) instead of 16 VALU instructions a lane.  This program checks that the
// matrix pipe's result is BIT-IDENTICAL to the VALU subtraction (the difference is exactly representable; the question is
// whether the pipe's adder aligns / truncates), over random magnitudes, zeros, negative values, values near the bf16 rounding
// boundaries and tiny / huge exponents, and times both forms.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_residual.hip -o /tmp/mfma_residual && /tmp/mfma_residual
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  asm volatile("" : "+v"(u));
  return u;
}
__device__ __forceinline__ float lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// A operand of k-step s: lane (row i = lane & 31, k group g = lane >> 5) holds A[i][8g .. 8g+7]; -1 where row i is the tile row
// of k-slot (g, i'): rho = (i' & 3) + 8 (2 s + (i' >> 2)) + 4 g   (accumulator register 8 s + i' of lane half g)
__device__ __forceinline__ u32x4 ident_frag(int lane, int s) {
  const int i = lane & 31, g = lane >> 5;
  u32x4 f = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int ip = 0; ip < 8; ++ip) {
    const int rho = (ip & 3) + 8 * (2 * s + (ip >> 2)) + 4 * g;
    if (rho == i) f[ip >> 1] |= (ip & 1) ? 0xBF800000u : 0x0000BF80u;  // bf16(-1.0) = 0xBF80
  }
  return f;
}

__global__ void check(const float* in, uint32_t* out_valu, uint32_t* out_mfma, int n_tiles) {
  const int lane = threadIdx.x & 63;
  const u32x4 I0 = ident_frag(lane, 0), I1 = ident_frag(lane, 1);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    f32x16 a;
    for (int r = 0; r < 16; ++r) a[r] = in[((int64_t)t * 16 + r) * 64 + lane];
    // VALU form
    uint32_t pv[3][8];
    for (int pr = 0; pr < 8; ++pr) {
      const float x = a[2 * pr], y = a[2 * pr + 1];
      pv[0][pr] = pack2(x, y);
      const float rx = x - lo(pv[0][pr]), ry = y - hi(pv[0][pr]);
      pv[1][pr] = pack2(rx, ry);
      const float rx2 = rx - lo(pv[1][pr]), ry2 = ry - hi(pv[1][pr]);
      pv[2][pr] = pack2(rx2, ry2);
    }
    // matrix-pipe form
    uint32_t pm[3][8];
    f32x16 r = a;
    for (int lvl = 0; lvl < 3; ++lvl) {
      for (int pr = 0; pr < 8; ++pr) pm[lvl][pr] = pack2(r[2 * pr], r[2 * pr + 1]);
      if (lvl == 2) break;
      const u32x4 f0 = {pm[lvl][0], pm[lvl][1], pm[lvl][2], pm[lvl][3]}, f1 = {pm[lvl][4], pm[lvl][5], pm[lvl][6], pm[lvl][7]};
      r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, I0), __builtin_bit_cast(bf16x8, f0), r, 0, 0, 0);
      r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, I1), __builtin_bit_cast(bf16x8, f1), r, 0, 0, 0);
    }
    for (int lvl = 0; lvl < 3; ++lvl)
      for (int pr = 0; pr < 8; ++pr) {
        out_valu[(((int64_t)t * 3 + lvl) * 8 + pr) * 64 + lane] = pv[lvl][pr];
        out_mfma[(((int64_t)t * 3 + lvl) * 8 + pr) * 64 + lane] = pm[lvl][pr];
      }
  }
}

// timing: NT tiles per iteration, split on the VALU (MODE 0) or with matrix-pipe residuals (MODE 1), each followed by the six
// products of ONE 16-k step per fragment (a stand-in consumer), two waves per SIMD as in the blend kernel
template <int MODE>
__global__ __launch_bounds__(512, 1) void timing(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const u32x4 I0 = ident_frag(lane, 0), I1 = ident_frag(lane, 1);
  f32x16 a, acc;
  for (int r = 0; r < 16; ++r) { a[r] = 1.0f + 0.001f * (lane + r); acc[r] = 0.f; }
  u32x4 w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int it = 0; it < iters; ++it) {
    uint32_t p[3][8];
    if (MODE == 0) {
      for (int pr = 0; pr < 8; ++pr) {
        const float x = a[2 * pr], y = a[2 * pr + 1];
        p[0][pr] = pack2(x, y);
        const float rx = x - lo(p[0][pr]), ry = y - hi(p[0][pr]);
        p[1][pr] = pack2(rx, ry);
        p[2][pr] = pack2(rx - lo(p[1][pr]), ry - hi(p[1][pr]));
      }
    } else {
      f32x16 r = a;
      for (int lvl = 0; lvl < 3; ++lvl) {
        for (int pr = 0; pr < 8; ++pr) p[lvl][pr] = pack2(r[2 * pr], r[2 * pr + 1]);
        if (lvl == 2) break;
        const u32x4 f0 = {p[lvl][0], p[lvl][1], p[lvl][2], p[lvl][3]}, f1 = {p[lvl][4], p[lvl][5], p[lvl][6], p[lvl][7]};
        r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, I0), __builtin_bit_cast(bf16x8, f0), r, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, I1), __builtin_bit_cast(bf16x8, f1), r, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < 2; ++s)
      for (int pc = 0; pc < 3; ++pc) {
        const u32x4 f = {p[pc][4 * s], p[pc][4 * s + 1], p[pc][4 * s + 2], p[pc][4 * s + 3]};
        for (int q = 0; q < (pc == 0 ? 3 : pc == 1 ? 2 : 1); ++q)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f), acc, 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    for (int r = 0; r < 16; ++r) a[r] = acc[r] * 1e-3f + 1.0f + 0.001f * r;   // next tile depends on this one (a layer chain)
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + a[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int n_tiles = 4096;
  const size_t n = (size_t)n_tiles * 16 * 64;
  std::vector<float> h(n);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (size_t i = 0; i < n; ++i) {
    const uint64_t r = rnd();
    const int kind = (int)(r % 11);
    uint32_t bits;
    if (kind == 0) bits = 0u;                                            // +0
    else if (kind == 1) bits = (uint32_t)(r >> 32);                        // any bit pattern (filtered below)
    else if (kind == 2) bits = ((uint32_t)(r >> 32) & 0xffff0000u) | 0x8000u;   // exactly on a bf16 tie
    else if (kind == 3) bits = ((uint32_t)(r >> 32) & 0xffff0000u) | 0x7fffu;   // just below a tie
    else if (kind == 4) bits = ((uint32_t)(r >> 32) & 0xffff0000u) | 0x8001u;   // just above a tie
    else if (kind == 5) bits = ((uint32_t)(r >> 32) & 0x807fffffu) | (1u << 23);        // smallest normal exponent
    else if (kind == 6) bits = ((uint32_t)(r >> 32) & 0x807fffffu) | (0xf0u << 23);     // huge (2^113; a first piece that ROUNDS UP TO INF, |a| > 3.39e38, would poison its column: 0 x inf)
    else { float f = ((int64_t)(r >> 20) % 2000001 - 1000000) * 1e-5f * ((kind & 1) ? 1.f : 37.f); memcpy(&bits, &f, 4); }
    const uint32_t e = (bits >> 23) & 0xff;
    if (e == 0xff) bits &= 0x807fffffu;                                    // no inf / nan
    if (e == 0 && (bits & 0x7fffffu)) bits &= 0x80000000u;                  // no denormal INPUTS (activations are not)
    memcpy(&h[i], &bits, 4);
  }
  float* d_in;
  uint32_t *d_v, *d_m;
  const size_t n_out = (size_t)n_tiles * 3 * 8 * 64;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_v, n_out * 4); hipMalloc(&d_m, n_out * 4);
  hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(256), dim3(64), 0, 0, d_in, d_v, d_m, n_tiles);
  std::vector<uint32_t> v(n_out), m(n_out);
  hipMemcpy(v.data(), d_v, n_out * 4, hipMemcpyDeviceToHost);
  hipMemcpy(m.data(), d_m, n_out * 4, hipMemcpyDeviceToHost);
  size_t bad[3] = {0, 0, 0}, shown = 0, nan_like = 0;
  for (size_t i = 0; i < n_out; ++i)
    if (v[i] != m[i]) {
      const int lvl = (int)((i / (8 * 64)) % 3);
      ++bad[lvl];
      if ((m[i] & 0x7f800000u) == 0x7f800000u || (m[i] & 0x7f80u) == 0x7f80u) ++nan_like;
      if (shown++ < 8) printf("  mismatch lvl %d: valu %08x mfma %08x\n", lvl, v[i], m[i]);
    }
  printf("pieces compared: %zu per level; mismatches p0 %zu, p1 %zu, p2 %zu (of which inf / nan pieces: %zu)\n", n_out / 3, bad[0], bad[1],
         bad[2], nan_like);
  float* d_o;
  hipMalloc(&d_o, 256 * 512 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    if (mode == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(512), 0, 0, d_o, 100); else hipLaunchKernelGGL(timing<1>, dim3(256), dim3(512), 0, 0, d_o, 100);
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(512), 0, 0, d_o, iters); else hipLaunchKernelGGL(timing<1>, dim3(256), dim3(512), 0, 0, d_o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f ns per tile per wave pair (two waves per SIMD; 16 values a lane split + 12 product MFMAs)\n",
           mode == 0 ? "VALU residuals" : "MFMA residuals", ms * 1e6f / iters);
  }
  return (bad[0] | bad[1] | bad[2]) ? 1 : 0;
}
