// Microbenchmark (gfx950): which counter tracks an LDS-DMA (buffer_load ... lds)?  One wavefront per CU issues a DMA of a
// fresh 1 KB block (cold in L2: stride 1 MB over a 1 GB buffer), then waits with (0) nothing, (1) s_waitcnt lgkmcnt(0),
// (2) s_waitcnt vmcnt(0), (3) a ds_read of OTHER LDS + the compiler-visible lgkmcnt(0).  Cycles per iteration tell whether
// the wait covers the whole memory round trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int MODE>
__global__ __launch_bounds__(64) void k(const char* src, unsigned long long* out, int iters, int stride) {
  __shared__ __attribute__((aligned(16))) char lds[8192];
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const int lane16 = threadIdx.x * 16;
  float acc = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int off = ((blockIdx.x * iters + it) * stride) & 0x3fffffff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (it & 3) * 1024), 16, lane16, off, 0, 0);
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 3) {
      float v;
      asm volatile("ds_read_b32 %0, %1 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane16) : "memory");
      acc += v;
    }
    if (MODE == 0 && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep the queue bounded
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.f) out[0] = 0;
}

int main() {
  char* src; unsigned long long* out;
  hipMalloc(&src, 1u << 30); hipMemset(src, 1, 1u << 30);
  hipMalloc(&out, 256 * 8);
  const int iters = 256;
  unsigned long long h[256];
  const char* names[4] = {"no wait (vmcnt(8) every 8)", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)", "ds_read + lgkmcnt(0)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, src, out, iters, 1 << 20);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, src, out, iters, 1 << 20);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64), 0, 0, src, out, iters, 1 << 20);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64), 0, 0, src, out, iters, 1 << 20);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-30s %8.0f clk / iteration\n", names[mode], s / 256 / iters);
  }
  return 0;
}
