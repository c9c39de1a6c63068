// Microbenchmark (gfx950), round 5: what a no-return global_atomic_add_f32 wave-instruction costs as a function of the SHAPE of its
// 64 lane addresses - the question behind costvol_bwd (8 segments of 32 B in 8 rows per instruction) and ptloss_bwd (64 lanes at a
// 16-byte stride: one channel of 64 adjacent texel4 entries).  Every wavefront issues ITER instructions of one shape into a 64 MB
// table at pseudo-random row bases; 256 CUs x 8 wavefronts.  Reported: ns per wave-instruction per CU and GB/s of added bytes.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/atomic_shapes.hip -o /tmp/atomic_shapes && /tmp/atomic_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// SHAPE 0: 64 contiguous dwords (256 B).  1: stride 16 B (one channel of 64 texel4 entries: 1 KB span).  2: eight octets, each 32
// contiguous bytes, in eight random rows.  3: 64 lanes in 64 random rows.  4: sixteen quads of 16 contiguous bytes in 16 random rows.
// 5: like 2 but only ONE octet active (7/8 of the lanes masked off).  6: like 1 but only every fourth lane active (16 lanes).
// TYPE 0: float add (the original question).  1: uint32 add.  2: uint64 add (dword index halved: the same byte span) - round 5,
// second pass: ds_add_f32 turned out 16x slower than the integer LDS atomics; is the memory side the same?
template <int SHAPE, int TYPE = 0>
__global__ __launch_bounds__(512) void k(float* tab, uint32_t n_rows, int iters) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 512 + threadIdx.x) / 64 * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    const uint32_t r = rnd(s);                                  // wave-uniform (same seed in all lanes)
    uint32_t row = r % n_rows;
    size_t off;
    bool on = true;
    if (SHAPE == 0) off = (size_t)row * 256 + lane;
    else if (SHAPE == 1 || SHAPE == 6) { off = (size_t)row * 256 + lane * 4; on = SHAPE == 1 || (lane & 3) == 0; }
    else if (SHAPE == 2 || SHAPE == 5) { const uint32_t rr = (row + (lane >> 3) * 7919u) % n_rows; off = (size_t)rr * 256 + (lane & 7); on = SHAPE == 2 || lane < 8; }
    else if (SHAPE == 3) { const uint32_t rr = (row + lane * 7919u) % n_rows; off = (size_t)rr * 256 + (lane & 3); }
    else { const uint32_t rr = (row + (lane >> 2) * 7919u) % n_rows; off = (size_t)rr * 256 + (lane & 3); }
    if (on) {
      if (TYPE == 0) atomicAdd(tab + off, 1.0f);
      else if (TYPE == 1) atomicAdd(reinterpret_cast<unsigned*>(tab) + off, 3u);
      else atomicAdd(reinterpret_cast<unsigned long long*>(tab) + (off >> 1), 3ull);
    }
  }
}

template <int SHAPE, int TYPE = 0>
void run(float* tab, uint32_t n_rows, const char* what, int lanes_on) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, TYPE>), dim3(256), dim3(512), 0, 0, tab, n_rows, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, TYPE>), dim3(256), dim3(512), 0, 0, tab, n_rows, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * iters;
  printf("%-62s %8.1f ns per wave-instruction per CU   %7.1f GB/s of added bytes\n", what, ms * 1e6 / instr_per_cu,
         256.0 * 8 * iters * lanes_on * 4 / (ms * 1e-3) / 1e9);
}

int main() {
  float* tab;
  const uint32_t n_rows = 65536;   // x 1 KB = 64 MB
  (void)hipMalloc(&tab, (size_t)n_rows * 1024);
  (void)hipMemset(tab, 0, (size_t)n_rows * 1024);
  run<0>(tab, n_rows, "0: 64 contiguous dwords (256 B)", 64);
  run<1>(tab, n_rows, "1: 64 lanes at a 16-byte stride (1 KB span)", 64);
  run<6>(tab, n_rows, "6: the same span, every fourth lane only (16 lanes)", 16);
  run<2>(tab, n_rows, "2: eight octets of 32 contiguous bytes in eight rows", 64);
  run<5>(tab, n_rows, "5: ONE octet of 32 bytes (56 lanes masked off)", 8);
  run<4>(tab, n_rows, "4: sixteen quads of 16 contiguous bytes in sixteen rows", 64);
  run<3>(tab, n_rows, "3: 64 lanes in 64 rows", 64);
  run<0, 1>(tab, n_rows, "0 / uint32: 64 contiguous dwords", 64);
  run<2, 1>(tab, n_rows, "2 / uint32: eight octets of 32 bytes in eight rows", 64);
  run<3, 1>(tab, n_rows, "3 / uint32: 64 lanes in 64 rows", 64);
  run<2, 2>(tab, n_rows, "2 / uint64: eight octets (4 qwords each) in eight rows", 64);
  run<3, 2>(tab, n_rows, "3 / uint64: 64 lanes in 64 rows", 64);
  return 0;
}
