// maxsim_probe.h -- the read ceiling of THIS box, measured with the rerank kernels' own fetch path and nothing else:
// every wave streams a contiguous piece of a buffer through non-temporal LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// instruction) into a private LDS ring and consumes nothing.  SURVEY 8(d): "measure achievable with a copy/read microbench
// on the box" -- bench.py divides the kernels' algorithmic bytes by this rate next to the 8 TB/s spec peak.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// TILE bytes per ring slot (8 or 16 KiB), NT slots per wave, 4 waves per workgroup; a wave reads `per_wave` bytes
// (a multiple of TILE) starting at (global wave id) * per_wave.
// The same rings fed with SCATTERED pieces: every `gran` bytes (a power of two >= 1 KiB: one doc of `gran` / row-bytes
// tokens) come from a hashed position of the buffer -- the access pattern of a rerank over short docs (C4: 4 KiB docs).
template <int TILE, int NT>
static __global__ void __launch_bounds__(256) k_read_probe_scatter(const char* __restrict__ buf, int64_t per_wave, int64_t nbuf_gran,
                                                                   int gran) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  constexpr int NDMA = TILE / 1024;
  char* const wlds = lds + wave * (NT * TILE);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int ntile = (int)(per_wave / TILE);
  const int per_gran = gran / 1024;  // DMA instructions per granule
  // (wave-uniform address arithmetic: nbuf_gran and per_gran are powers of two; the piece index lives in scalar registers)
  const uint32_t gmask = (uint32_t)(nbuf_gran - 1);
  const int gshift = __builtin_ctz((unsigned)per_gran), lgran = __builtin_ctz((unsigned)gran);
  const uint32_t piece0 = (uint32_t)(wid * ntile * NDMA);
  auto src_of = [&](int tile, int i) -> const char* {  // instruction i of tile `tile`: 1 KiB inside a hashed granule
    const uint32_t piece = piece0 + (uint32_t)(tile * NDMA + i);  // global 1 KiB piece index
    uint32_t x = (piece >> gshift) * 0x9E3779B9u;
    x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
    const uint32_t where = (uint32_t)uni((int)(x & gmask));
    return buf + ((uint64_t)where << lgran) + (uint64_t)((piece & (uint32_t)(per_gran - 1)) * 1024u) + lane * 16;
  };
  int issued = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (issued < ntile) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(src_of(issued, i)), LPTR(wlds + j * TILE + i * 1024), 16, 0, 2);
      ++issued;
    }
  int buf_i = 0;
  for (int done = 0; done < ntile; ++done) {
    if (issued - done == NT) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
    if (issued < ntile) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(src_of(issued, i)), LPTR(wlds + buf_i * TILE + i * 1024), 16, 0, 2);
      ++issued;
    }
    buf_i = (buf_i + 1 == NT) ? 0 : buf_i + 1;
  }
  wait_vmcnt<0>();
}

template <int TILE, int NT>
static __global__ void __launch_bounds__(256) k_read_probe(const char* __restrict__ buf, int64_t per_wave) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  constexpr int NDMA = TILE / 1024;
  char* const wlds = lds + wave * (NT * TILE);
  const char* src = buf + ((int64_t)blockIdx.x * 4 + wave) * per_wave + lane * 16;
  const int ntile = (int)(per_wave / TILE);
  int issued = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (issued < ntile) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(src + (int64_t)issued * TILE + i * 1024), LPTR(wlds + j * TILE + i * 1024), 16, 0, 2);
      ++issued;
    }
  int buf_i = 0;
  for (int done = 0; done < ntile; ++done) {
    if (issued - done == NT) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
    if (issued < ntile) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(src + (int64_t)issued * TILE + i * 1024), LPTR(wlds + buf_i * TILE + i * 1024), 16, 0, 2);
      ++issued;
    }
    buf_i = (buf_i + 1 == NT) ? 0 : buf_i + 1;
  }
  wait_vmcnt<0>();
}

}  // namespace maxsim
