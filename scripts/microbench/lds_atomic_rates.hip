// Microbenchmark (gfx950), round 5: what an LDS atomic wave-instruction costs by type - the question behind costvol_tile_kernel,
// whose 1.6 G ds_add_f32 lane-operations per training step take 7.2 ms (~150 clocks per 64-lane instruction).
// Every wavefront issues ITER no-return atomics into a 32 KB LDS array at pseudo-random word addresses (distinct per lane in the
// "spread" variants: lane l touches words = l (mod 64), i.e. one lane per bank; "random": any word).  256 CUs x 8 wavefronts.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/lds_atomic_rates.hip -o /tmp/lds_atomic_rates && /tmp/lds_atomic_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// KIND 0: float add.  1: uint32 add.  2: uint64 add.  3: plain (non-atomic) float read-modify-write.  4: float max (ds_max_f32)
template <int KIND, bool SPREAD>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ unsigned long long lds64[4096];           // 32 KB
  float* f = reinterpret_cast<float*>(lds64);
  unsigned* u = reinterpret_cast<unsigned*>(lds64);
  for (int e = threadIdx.x; e < 4096; e += 512) lds64[e] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    const uint32_t r = rnd(s);
    if (KIND == 2) {
      const int w = SPREAD ? (int)((r % 64) * 64 + lane) % 4096 : (int)(r % 4096);
      atomicAdd(&lds64[w], (unsigned long long)(r | 1u));
    } else {
      const int w = SPREAD ? (int)((r % 128) * 64 + lane) : (int)(r % 8192);
      if (KIND == 0) atomicAdd(&f[w], 1.0f);
      else if (KIND == 1) atomicAdd(&u[w], r | 1u);
      else if (KIND == 3) f[w] += 1.0f;
      else atomicMax(reinterpret_cast<int*>(&u[w]), (int)(r >> 1));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[0] + (float)u[1];
}

template <int KIND, bool SPREAD>
void run(float* out, const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, SPREAD>), dim3(256), dim3(512), 0, 0, out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, SPREAD>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.1f ns per wave-instruction per CU\n", what, ms * 1e6 / (8.0 * iters));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 1024 * 4);
  run<0, true>(out, "ds_add_f32, one lane per bank");
  run<0, false>(out, "ds_add_f32, random words");
  run<1, true>(out, "ds_add_u32, one lane per bank");
  run<1, false>(out, "ds_add_u32, random words");
  run<2, true>(out, "ds_add_u64, spread");
  run<2, false>(out, "ds_add_u64, random words");
  run<4, false>(out, "ds_max_i32, random words");
  run<3, true>(out, "plain float read-modify-write, one lane per bank");
  run<3, false>(out, "plain float read-modify-write, random words");
  return 0;
}
