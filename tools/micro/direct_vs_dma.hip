// Micro-benchmark (GPU box): the fp32 streaming kernel's inner cycle with the doc tile fetched (A) by LDS-DMA into a per-wave
// 16 KiB ring slot and read back with ds_read_b128 -- what k_maxsim_stream does -- against (B) plain non-temporal
// global_load_dwordx4 straight into the MFMA operand registers (no LDS at all), three 16-row blocks of registers in rotation.
// Both run the same matrix work per tile (32 rows x 32 query tokens x 128 dims on v_mfma_f32_16x16x4_f32 = 128 MFMAs, query in
// registers) and a token max; rows are contiguous (no doc descriptors, no packing), so only the DIFFERENCE between A and B means
// anything: does taking LDS out of the path (write + read of every byte, the wave held on each LDS-DMA instruction) buy time
// under the same matrix load?   hipcc --offload-arch=gfx950 -O3 tools/micro/direct_vs_dma.hip -o tools/micro/direct_vs_dma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef float f4 __attribute__((ext_vector_type(4)));

// -DPRIO=1: the contraction at raised wave priority, as k_maxsim_stream runs it -- the matrix pipe then finishes one wave's tile
// before the other wave's instead of interleaving them 1 : 1, so that one wave of a SIMD fetches while the other contracts
#ifndef PRIO
#define PRIO 0
#endif
__device__ __forceinline__ void block_mfma(const f4 (&d)[8], const float (&q)[2][32], f4 (&acc)[2]) {
  if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[i][e], q[0][4 * i + e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[i][e], q[1][4 * i + e], acc[1], 0, 0, 0);
    }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ float block_max(f4 (&acc)[2]) {
  float m = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3])) + fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3]));
  acc[0] = f4{0, 0, 0, 0};
  acc[1] = f4{0, 0, 0, 0};
  return m;
}

__device__ __forceinline__ void load_q(const float* qsrc, int lane, float (&q)[2][32]) {
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s = 0; s < 32; ++s) q[b][s] = qsrc[(b * 32 + s) * 64 + lane];
}

// (A) one 16 KiB ring slot per wave: wait, read the tile into registers, request the next, 128 MFMAs
template <bool MFMA>
__global__ void __launch_bounds__(256, 2) k_dma(const char* __restrict__ buf, const float* __restrict__ qsrc, int tiles_per_wave, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const wlds = lds + wave * 16384;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const char* const base = buf + wid * (int64_t)tiles_per_wave * 16384 + lane * 16;
  float q[2][32];
  load_q(qsrc, lane, q);
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) __builtin_amdgcn_global_load_lds(GPTR(base + (int64_t)t * 16384 + i * 1024), LPTR(wlds + i * 1024), 16, 0, 2);
  };
  f4 acc[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  float total = 0.f;
  issue(0);
  for (int t = 0; t < tiles_per_wave; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f4 d[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) d[b][i] = *(const f4*)(wlds + (b * 8 + i) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < tiles_per_wave) issue(t + 1);
    if (MFMA) {
      block_mfma(d[0], q, acc);
      total += block_max(acc);
      block_mfma(d[1], q, acc);
      total += block_max(acc);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) total += d[0][i][0] + d[1][i][3];
    }
  }
  if (total == 123.456f) out[0] = total;
}

// (B) no LDS: three 16-row blocks of operand registers in rotation, two in flight while the third is contracted.
// lane (n = lane & 15, kq = lane >> 4) loads the 16 bytes at row n, byte 64 i + 16 kq: the operand layout the kernel reads from LDS
template <bool MFMA>
__global__ void __launch_bounds__(256, 2) k_direct(const char* __restrict__ buf, const float* __restrict__ qsrc, int tiles_per_wave, float* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const char* const base = buf + wid * (int64_t)tiles_per_wave * 16384 + (lane & 15) * 512 + (lane >> 4) * 16;
  float q[2][32];
  load_q(qsrc, lane, q);
  const int nblk = tiles_per_wave * 2;     // 16-row blocks of 8 KiB
  auto load = [&](f4 (&d)[8], int h) {
    const f4* p = (const f4*)(base + (int64_t)(h < nblk ? h : nblk - 1) * 8192);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_nontemporal_load(p + 4 * i);
  };
  f4 acc[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  float total = 0.f;
  f4 A[8], B[8], C[8];
  load(A, 0);
  load(B, 1);
  load(C, 2);
  auto use = [&](f4 (&d)[8]) {
    if (MFMA) {
      block_mfma(d, q, acc);
      total += block_max(acc);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) total += d[i][0] + d[i][3];
    }
  };
  for (int h = 0; h < nblk; h += 3) {
    use(A);
    __builtin_amdgcn_sched_barrier(0);
    load(A, h + 3);
    __builtin_amdgcn_sched_barrier(0);
    if (h + 1 < nblk) use(B);
    __builtin_amdgcn_sched_barrier(0);
    load(B, h + 4);
    __builtin_amdgcn_sched_barrier(0);
    if (h + 2 < nblk) use(C);
    __builtin_amdgcn_sched_barrier(0);
    load(C, h + 5);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (total == 123.456f) out[0] = total;
}

// (B2) as (B) with TWO whole tiles of registers in rotation (4 x 16-row blocks: one tile in flight while the other is contracted,
// and the next requested block by block as its registers free up) -- 32 KiB per wave in flight at the peak, twice the LDS ring's
template <bool MFMA>
__global__ void __launch_bounds__(256, 2) k_direct2(const char* __restrict__ buf, const float* __restrict__ qsrc, int tiles_per_wave, float* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const char* const base = buf + wid * (int64_t)tiles_per_wave * 16384 + (lane & 15) * 512 + (lane >> 4) * 16;
  float q[2][32];
  load_q(qsrc, lane, q);
  const int nblk = tiles_per_wave * 2;
  auto load = [&](f4 (&d)[8], int h) {
    const f4* p = (const f4*)(base + (int64_t)(h < nblk ? h : nblk - 1) * 8192);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_nontemporal_load(p + 4 * i);
  };
  f4 acc[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  float total = 0.f;
  f4 A[8], B[8], C[8], D[8];
  load(A, 0);
  load(B, 1);
  load(C, 2);
  load(D, 3);
  auto use = [&](f4 (&d)[8]) {
    if (MFMA) {
      block_mfma(d, q, acc);
      total += block_max(acc);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) total += d[i][0] + d[i][3];
    }
  };
#define STEP(X, k)                                \
  if (h + k < nblk) use(X);                       \
  __builtin_amdgcn_sched_barrier(0);              \
  load(X, h + 4 + k);                             \
  __builtin_amdgcn_sched_barrier(0);
  for (int h = 0; h < nblk; h += 4) {
    STEP(A, 0) STEP(B, 1) STEP(C, 2) STEP(D, 3)
  }
#undef STEP
  if (total == 123.456f) out[0] = total;
}

// operands with live bits (zeros would cost the matrix pipe far less power than real embeddings do)
__global__ void k_fill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((int)(x & 0xffff) - 32768) * (0.09f / 32768.f);
  }
}

template <typename K>
double run(K kern, int ldsb, const char* buf, const float* q, float* out, int wgs, int tiles) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int r = 0; r < 25; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), ldsb, 0, buf, q, tiles, out);   // ~100 ms: off the clock ramp
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), ldsb, 0, buf, q, tiles, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)wgs * 4 * tiles * 16384 * 10 / (ms * 1e-3) / 1e9;
}

int main() {
  const int wgs = 256 * 2 * 8, tiles = 90;     // 8 rounds of two workgroups per CU; 90 tiles = 16 docs of 180 tokens per wave
  const size_t nbytes = (size_t)wgs * 4 * tiles * 16384 + (1 << 20);
  char* buf = nullptr;
  float *q = nullptr, *out = nullptr;
  if (hipMalloc(&buf, nbytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  (void)hipMalloc(&q, 64 * 64 * 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float*)buf, nbytes / 4);
  hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, q, (size_t)64 * 64);
  (void)hipMalloc(&out, 64);
  printf("%.1f GB per launch, %d workgroups x 4 waves, %d tiles of 16 KiB per wave\n", (double)wgs * 4 * tiles * 16384 / 1e9, wgs, tiles);
  for (int rep = 0; rep < 2; ++rep) {
    printf("LDS-DMA ring + ds_read:   with MFMAs %.0f GB/s   fetch only %.0f GB/s\n", run(k_dma<true>, 65536, buf, q, out, wgs, tiles),
           run(k_dma<false>, 65536, buf, q, out, wgs, tiles));
    fflush(stdout);
    printf("direct to registers:      with MFMAs %.0f GB/s   fetch only %.0f GB/s\n", run(k_direct<true>, 0, buf, q, out, wgs, tiles),
           run(k_direct<false>, 0, buf, q, out, wgs, tiles));
    fflush(stdout);
    printf("direct, two tiles of registers: with MFMAs %.0f GB/s   fetch only %.0f GB/s\n", run(k_direct2<true>, 0, buf, q, out, wgs, tiles),
           run(k_direct2<false>, 0, buf, q, out, wgs, tiles));
    fflush(stdout);
  }
  (void)hipFree(buf);
  return 0;
}
