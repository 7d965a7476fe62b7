// Check (GPU box): s_atomic_add (scalar memory atomic, counted in lgkmcnt, not vmcnt) on gfx950: does it work, and how long
// does a returning one take while 2048 waves use it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(int* ctr, int nq, int iters, unsigned long long* cyc, int* seen) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int* q = ctr + 32 * ((blockIdx.x * 4 + wave) % nq);   // one counter per 128 B
  unsigned long long t = 0;
  int last = -1;
  for (int i = 0; i < iters; ++i) {
    int v = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(q) : "memory");
    t += __builtin_amdgcn_s_memtime() - t0;
    if (v <= last) seen[0] = 1;   // values a wave sees must increase
    last = v;
    __builtin_amdgcn_s_sleep(64);
  }
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t;
}
int main() {
  int* ctr; unsigned long long* cyc; int* seen;
  hipMalloc(&ctr, 64 * 128); hipMalloc(&cyc, 2048 * 8); hipMalloc(&seen, 4);
  for (int nq : {1, 8, 64}) {
    hipMemset(ctr, 0, 64 * 128); hipMemset(seen, 0, 4);
    const int iters = 8;
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, ctr, nq, iters, cyc, seen);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2048); std::vector<int> c(64 * 32); int s;
    hipMemcpy(h.data(), cyc, 2048 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), ctr, 64 * 128, hipMemcpyDeviceToHost);
    hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
    long long tot = 0; for (int i = 0; i < nq; ++i) tot += c[32 * i];
    double avg = 0; for (auto v : h) avg += v; avg /= 2048.0 * iters;
    printf("queues %2d: counter total %lld (expect %d), monotonic per wave: %s, %.0f cycles per returning s_atomic_add\n", nq, tot, 2048 * iters, s ? "NO" : "yes", avg);
  }
  return 0;
}
