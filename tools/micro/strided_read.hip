// Micro-benchmark (GPU box): what HBM read rate does the LDS-query kernel's FETCH PATTERN reach, by piece size?
// k_maxsim_stream_bigh streams a doc of wide rows (dim 768 fp16: 1536 B per row) one 128-dim block at a time: a sub-tile is
// 32 rows x 256 B -- 256-byte pieces at a 1536-byte stride -- and the same rows are visited again for each of the 6 blocks.
// A timing build without its MFMAs runs no faster (dep768 / C5: 0.80-0.81 of 8 TB/s), while the same ring shapes reach
// 0.84-0.86 on whole 256-byte rows back to back.  This program replays only the fetch, with the piece size as a parameter:
//   every wave owns consecutive TILES of RPS rows x ROWB bytes (RPS = 8192 / PIECE rows: an 8 KiB sub-tile per sweep) and
//   sweeps each tile ROWB / PIECE times, PIECE bytes of every row per sweep, through non-temporal LDS-DMA into its one
//   8 KiB ring slot (wait, re-issue: one sub-tile in flight per wave, as the kernel's 8 x 1 ring); nothing is consumed.
//   WAVES waves per workgroup, one workgroup per CU (LDS padded to 160 KiB like the query image does).
// Prints GB/s per (ROWB, PIECE, WAVES).  hipcc --offload-arch=gfx950 -O3 tools/micro/strided_read.hip -o tools/micro/strided_read
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int PIECE>
__global__ void __launch_bounds__(1024) k_strided(const char* __restrict__ buf, int rowb, int tiles_per_wave, int ring_off) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  constexpr int RPS = 8192 / PIECE;        // rows per sub-tile
  constexpr int LPR = PIECE / 16;          // lanes per row piece (PIECE <= 1024) ...
  constexpr int LPRC = LPR > 64 ? 64 : LPR;
  constexpr int RPI = 64 / LPRC;           // rows per DMA instruction (>= 1)
  constexpr int IPR = LPR > 64 ? LPR / 64 : 1;  // instructions per row piece (PIECE = 2048: 2)
  char* const wlds = lds + ring_off + wave * 8192;
  const int sweeps = rowb / PIECE;
  const int64_t wid = (int64_t)blockIdx.x * nwaves + wave;
  const char* const base = buf + wid * (int64_t)tiles_per_wave * RPS * rowb;
  const uint32_t loff = (uint32_t)((lane / LPRC) * rowb + (lane % LPRC) * 16);
  for (int t = 0; t < tiles_per_wave; ++t) {
    const char* const tile = base + (int64_t)t * RPS * rowb;
    for (int s = 0; s < sweeps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {       // 8 instructions = 8 KiB
        const int row = (i / IPR) * RPI;  // first row of this instruction
        const char* g = tile + (int64_t)row * rowb + s * PIECE + (i % IPR) * 1024 + loff;
        __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(wlds + i * 1024), 16, 0, 2);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The kernel's real cycle: a slot is READ (ds_read_b128 of the whole sub-tile into registers, s_waitcnt lgkmcnt(0)) before it is
// re-issued.  NSLOT slots of SLOTB bytes per wave (8 KiB in all): 1 x 8 KiB (32 rows x 256 B: what the kernel has beside a 96 KiB
// query image) against 2 x 4 KiB (32 rows x 128 B: a 64-dim sub-tile) -- with two slots one is always in flight.
template <int NSLOT>
__global__ void __launch_bounds__(1024) k_ring(const char* __restrict__ buf, int rowb, int tiles_per_wave, int ring_off, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int SLOTB = 8192 / NSLOT, PIECE = SLOTB / 32, NI = SLOTB / 1024;   // 32 rows per sub-tile
  constexpr int LPR = PIECE / 16, RPI = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  char* const wlds = lds + ring_off + wave * 8192;
  const int sweeps = rowb / PIECE;
  const int64_t wid = (int64_t)blockIdx.x * nwaves + wave;
  const char* const base = buf + wid * (int64_t)tiles_per_wave * 32 * rowb;
  const uint32_t loff = (uint32_t)((lane / LPR) * rowb + (lane % LPR) * 16);
  const int total = tiles_per_wave * sweeps;
  auto issue = [&](int u, int slot) {
    const char* const tile = base + (int64_t)(u / sweeps) * 32 * rowb + (u % sweeps) * PIECE;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(tile + (int64_t)(i * RPI) * rowb + loff), LPTR(wlds + slot * SLOTB + i * 1024), 16, 0, 2);
  };
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) issue(j, j);
  for (int u = 0; u < total; ++u) {
    const int slot = u % NSLOT;
    if (NSLOT == 2 && u + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 a[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = *(const f4*)(wlds + slot * SLOTB + i * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (u + NSLOT < total) issue(u + NSLOT, slot);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc += a[i][0];
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int NSLOT>
double run_ring(const char* buf, size_t nbytes, int rowb, int waves, int ring_pad, float* sink) {
  const int wgs = 256 * 8;
  const size_t tile_bytes = (size_t)32 * rowb;
  int tiles = (int)(nbytes / ((size_t)wgs * waves * tile_bytes));
  if (tiles > 64) tiles = 64;
  const size_t total = (size_t)wgs * waves * tiles * tile_bytes;
  const int ldsb = ring_pad + waves * 8192;
  (void)hipFuncSetAttribute((const void*)k_ring<NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ring<NSLOT>, dim3(wgs), dim3(waves * 64), ldsb, 0, buf, rowb, tiles, ring_pad, sink);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_ring<NSLOT>, dim3(wgs), dim3(waves * 64), ldsb, 0, buf, rowb, tiles, ring_pad, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)total * 3 / (ms * 1e-3) / 1e9;
}

template <int PIECE>
double run(const char* buf, size_t nbytes, int rowb, int waves, int ring_pad) {
  constexpr int RPS = 8192 / PIECE;
  const int wgs = 256 * 8;                                   // 8 rounds of one workgroup per CU
  const size_t tile_bytes = (size_t)RPS * rowb;
  int tiles = (int)(nbytes / ((size_t)wgs * waves * tile_bytes));
  if (tiles > 64) tiles = 64;
  const size_t total = (size_t)wgs * waves * tiles * tile_bytes;
  const int ldsb = ring_pad + waves * 8192;
  hipFuncSetAttribute((const void*)k_strided<PIECE>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_strided<PIECE>, dim3(wgs), dim3(waves * 64), ldsb, 0, buf, rowb, tiles, ring_pad);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_strided<PIECE>, dim3(wgs), dim3(waves * 64), ldsb, 0, buf, rowb, tiles, ring_pad);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)total * 3 / (ms * 1e-3) / 1e9;
}

int main() {
  const size_t nbytes = (size_t)24 << 30;
  char* buf = nullptr;
  if (hipMalloc(&buf, nbytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, nbytes);
  for (int waves : {8, 12}) {
    const int pad = 160 * 1024 - waves * 8192 - (waves == 8 ? 0 : 0);   // one workgroup per CU
    for (int rowb : {1536, 2048}) {
      printf("rows of %d B, %d waves x 8 KiB per CU:", rowb, waves);
      printf("  piece 256: %.0f GB/s", run<256>(buf, nbytes, rowb, waves, pad));
      printf("  512: %.0f", run<512>(buf, nbytes, rowb, waves, pad));
      if (rowb % 1024 == 0) printf("  1024: %.0f", run<1024>(buf, nbytes, rowb, waves, pad));
      if (rowb == 2048) printf("  2048 (contiguous): %.0f", run<2048>(buf, nbytes, rowb, waves, pad));
      printf("\n");
      fflush(stdout);
    }
  }
  float* sink = nullptr;
  (void)hipMalloc(&sink, 64);
  for (int waves : {8, 12}) {
    const int pad = 160 * 1024 - waves * 8192;
    printf("rows of 1536 B, %d waves, slots READ before re-issue:  1 x 8 KiB (256-B pieces): %.0f GB/s   2 x 4 KiB (128-B pieces): %.0f GB/s\n",
           waves, run_ring<1>(buf, nbytes, 1536, waves, pad, sink), run_ring<2>(buf, nbytes, 1536, waves, pad, sink));
    fflush(stdout);
  }
  (void)hipFree(buf);
  return 0;
}
