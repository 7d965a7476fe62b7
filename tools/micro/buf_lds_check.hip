// Check (GPU box): does buffer_load_dwordx4 ... lds reach LDS addresses past 64 KB, and does the range check count soffset?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void __launch_bounds__(64) k(const uint32_t* src, int nbytes, uint32_t* out, uint32_t lds_off, int soff) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef __attribute__((address_space(3))) char* lp;
  const lp l3 = (lp)((__attribute__((address_space(3))) void*)lds);
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) ((uint32_t*)lds)[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(l3 + lds_off), 16, (int)(threadIdx.x * 16), soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((uint32_t*)(lds + lds_off))[i];
  if (threadIdx.x == 0) out[256] = ((uint32_t*)lds)[(lds_off & 0xffff) / 4];  // where a 16-bit wrap would land
}
int main() {
  std::vector<uint32_t> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 1000 + i;
  uint32_t *src, *out;
  hipMalloc(&src, 4096 * 4); hipMalloc(&out, 257 * 4);
  hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  std::vector<uint32_t> o(257);
  struct { uint32_t off; int nbytes; int soff; } cases[] = {{1024, 16384, 0}, {100 * 1024, 16384, 0}, {150 * 1024, 16384, 0}, {1024, 512, 0}, {1024, 2048, 1024}, {1024, 1536, 1024}};
  for (auto c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, c.nbytes, out, c.off, c.soff);
    hipMemcpy(o.data(), out, 257 * 4, hipMemcpyDeviceToHost);
    printf("lds_off %6u num_records %5d soffset %4d: word0 %u word1 %u word127 %u word128 %u word255 %u | at 16-bit wrap: %x\n", c.off, c.nbytes, c.soff, o[0], o[1], o[127], o[128], o[255], o[256]);
  }
  return 0;
}
