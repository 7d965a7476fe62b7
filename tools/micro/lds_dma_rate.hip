// Micro-benchmark (GPU box): how fast does a CU take LDS-DMA instructions (global_load_lds_dwordx4, 1 KB per wave
// instruction) from an L2-resident matrix, by access shape?  One 512-thread workgroup per CU, every wave issues
// instructions back to back (vmcnt-limited), rows of 1536 B (dim 768, 16-bit):
//   shape 0: 16 rows x  64 B per instruction (a 32-dim K slice: half a 128-B line per row)
//   shape 1:  8 rows x 128 B per instruction (a 64-dim K slice: whole lines)
//   shape 2:  4 rows x 256 B
// Prints bytes / clock / CU.  hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_rate.hip -o /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int SHAPE>
__global__ void __launch_bounds__(512) k(const char* src, int rows_total, int iters, uint64_t* cycles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int ROWB = 1536;
  constexpr int LPR = SHAPE == 0 ? 4 : SHAPE == 1 ? 8 : 16;   // lanes per row
  constexpr int RPI = 64 / LPR;                               // rows per instruction
  const uint32_t loff = (uint32_t)((lane / LPR) * ROWB + (lane % LPR) * 16);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  // each workgroup walks its own 576-row window (a tile's slice image), slice after slice, like the kernel does
  const char* base = src + (size_t)(((blockIdx.x & 7) * 600) % (rows_total - 600)) * ROWB;  // an XCD's workgroups share a window: L2 hits
  for (int it = 0; it < iters; ++it) {
    const int col = (it % (ROWB / (LPR * 16))) * (LPR * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * RPI;   // 32 instructions per iteration per workgroup = 32 KB
      __builtin_amdgcn_global_load_lds(GPTR(base + (size_t)row * ROWB + col + loff), LPTR(lds + ((it & 3) * 32 + wave * 4 + j) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int rows = 544 * 384, iters = 4000;
  char* src; uint64_t* cyc;
  hipMalloc(&src, (size_t)rows * 1536);
  hipMemset(src, 1, (size_t)rows * 1536);
  hipMalloc(&cyc, 256 * 8);
  std::vector<uint64_t> h(256);
  for (int shape = 0; shape < 3; ++shape) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 128 * 1024, 0, src, rows, iters, cyc);
      if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 128 * 1024, 0, src, rows, iters, cyc);
      if (shape == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 128 * 1024, 0, src, rows, iters, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto c : h) avg += c; avg /= 256;
      const double bytes = (double)iters * 32 * 1024;
      printf("shape %d: %.3f ms  %.1f cycles per instruction per CU  %.1f B/clk/CU  %.2f TB/s chip\n", shape, ms, avg / (iters * 32.0), bytes / avg, bytes * 256 / ms / 1e9);
    }
  }
  return 0;
}
