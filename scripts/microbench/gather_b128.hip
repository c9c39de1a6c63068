// Microbenchmark (gfx950), round 4: what does a wavefront pay for 16-byte GATHERS (every lane its own address: the bilinear
// taps of the blend kernel's pass 1)?  A wavefront issues NB batches of 12 global_load_dwordx4 whose lane addresses are
// (a) random over a 48 MB map set, (b) a smooth walk (neighbouring lanes 1-2 texels apart, as consecutive samples of a ray),
// (c) identical for all lanes, consumes each batch before the next (DEPTH = 1) or keeps DEPTH batches in flight.
// Printed: shader clocks per batch of 12 gathers as seen by the wavefront, for 1 / 2 / 4 / 8 wavefronts per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ map, uint32_t n_texels, int mode, int nb, unsigned long long* clk, float* sink) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t walk = (uint32_t)((uint64_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2654435761u % n_texels);
  f32x4 acc = {0, 0, 0, 0};
  f32x4 buf[DEPTH][12];
  auto issue = [&](int d, int b) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      uint32_t idx;
      if (mode == 0) { s = s * 1664525u + 1013904223u; idx = (s >> 4) % n_texels; }
      else if (mode == 1) idx = (walk + (uint32_t)(b * 12 + t) * 977u + (uint32_t)lane * 2u + (uint32_t)(t & 1) + (uint32_t)((t >> 1) & 1) * 800u) % n_texels;
      else idx = (walk + (uint32_t)(b * 12 + t) * 977u) % n_texels;
      buf[d][t] = map[idx];
    }
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  for (int b = 0; b < nb; ++b) {
    if (b + DEPTH - 1 < nb) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        if ((b + DEPTH - 1) % DEPTH == d) issue(d, b + DEPTH - 1);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (b % DEPTH == d) {
#pragma unroll
        for (int t = 0; t < 12; ++t) acc += buf[d][t];
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) clk[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  const uint32_t n_texels = 3 * 1024 * 1024;   // 48 MB of 16-byte texels
  f32x4* map; unsigned long long* clk; float* sink;
  (void)hipMalloc(&map, (size_t)n_texels * 16); (void)hipMemset(map, 0, (size_t)n_texels * 16);
  (void)hipMalloc(&clk, 256 * 8 * 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  const char* modes[3] = {"random", "ray-like walk", "uniform"};
  for (int mode = 0; mode < 3; ++mode)
    for (int wpc : {1, 4, 8}) {
      for (int depth : {1, 3}) {
        const int nb = 64;
        for (int rep = 0; rep < 2; ++rep) {
          if (depth == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * wpc), 0, 0, map, n_texels, mode, nb, clk, sink);
          else hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * wpc), 0, 0, map, n_texels, mode, nb, clk, sink);
          (void)hipDeviceSynchronize();
        }
        unsigned long long h[256 * 8];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double sum = 0; int cnt = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < wpc; ++w) { sum += (double)h[b * 8 + w]; ++cnt; }
        printf("%-14s %d wave(s)/CU, %d batch(es) in flight: %7.0f clocks per batch of 12 gathers (%5.0f per gather)\n", modes[mode], wpc, depth,
               sum / cnt / nb, sum / cnt / nb / 12);
      }
    }
  return 0;
}
