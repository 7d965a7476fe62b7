// rank_forward_lat.cpp -- the online call (maxsim_rank_forward: 1 query x n candidates -> top-k) driven from C++, no Python:
// what one call costs end to end from a host thread, and where the time goes.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/micro/rank_forward_lat.cpp -o tools/micro/rank_forward_lat \
//         -Lcolbert_amd -lmaxsim -Wl,-rpath,$PWD/colbert_amd        (or link a diagnostic build: -Ltools/ab -l:diag.so)
//   DT=fp32|fp16  NDOCS=1000000  LD=180  N=1000  K=100  CALLS=2000
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <random>
#include <vector>

#include "maxsim.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e, hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill_index(uint32_t* p, int64_t nwords, uint32_t seed, int fp16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    if (fp16) {  // two halves in [-0.09, 0.09]: exponent 0x2c.., random mantissa and sign
      p[i] = (x & 0x83ff83ffu) | 0x2c002c00u;
    } else {     // a float in +-[0.0625, 0.125)
      p[i] = (x & 0x807fffffu) | 0x3d800000u;
    }
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int envi(const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; }
static void report(const char* what, std::vector<double> v) {
  std::sort(v.begin(), v.end());
  printf("%-64s median %7.2f us   p10 %7.2f   p90 %7.2f   min %7.2f\n", what, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v[0]);
}

int main() {
  const char* dts = getenv("DT") ? getenv("DT") : "fp32";
  const int fp16 = strcmp(dts, "fp16") == 0;
  const int64_t ndocs = envi("NDOCS", 1000000);
  const int ld = envi("LD", 180), h = 128, lq = 32, n = envi("N", 1000), k = envi("K", 100), calls = envi("CALLS", 2000);
  const int64_t ntok = ndocs * ld, bytes = ntok * h * (fp16 ? 2 : 4);
  void* index;
  CK(hipMalloc(&index, bytes));
  hipLaunchKernelGGL(k_fill_index, dim3(4096), dim3(256), 0, 0, (uint32_t*)index, bytes / 4, 12345u, fp16);
  std::vector<int64_t> offs(ndocs);
  std::vector<int32_t> lens(ndocs, ld);
  for (int64_t i = 0; i < ndocs; ++i) offs[i] = i * ld;
  int64_t* doffs; int32_t* dlens; void* table;
  CK(hipMalloc((void**)&doffs, ndocs * 8)); CK(hipMalloc((void**)&dlens, ndocs * 4)); CK(hipMalloc(&table, maxsim_doc_table_bytes(ndocs)));
  CK(hipMemcpy(doffs, offs.data(), ndocs * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlens, lens.data(), ndocs * 4, hipMemcpyHostToDevice));
  if (maxsim_build_doc_table(doffs, dlens, nullptr, ndocs, table, nullptr) != 0) return 2;
  std::vector<float> q(lq * h);
  std::mt19937_64 rng(7);
  for (auto& x : q) x = (float)((int)(rng() % 2001) - 1000) / 11000.0f;
  float* dq; CK(hipMalloc((void**)&dq, q.size() * 4)); CK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  maxsim_index_view iv{};
  iv.index = index; iv.index_dtype = fp16 ? MAXSIM_F16 : MAXSIM_F32; iv.h = h; iv.n_tokens = ntok; iv.tok_offsets = doffs;
  iv.doclens = dlens; iv.pad_len = nullptr; iv.n_docs = ndocs; iv.doc_table = table;
  int64_t* pin; CK(hipHostMalloc((void**)&pin, 16384 * 8, hipHostMallocDefault));
  char* out = (char*)maxsim_host_alloc_coherent(16384 * 12 + 64);
  if (!out) return 3;
  memset(out, 0, 16384 * 12 + 64);
  int64_t* op = (int64_t*)out; float* os = (float*)(out + 16384 * 8); uint32_t* flag = (uint32_t*)(out + 16384 * 12);
  void* ws; const int64_t wsb = maxsim_rank_forward_workspace_bytes(16384);
  CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  std::vector<std::vector<int64_t>> lists(64, std::vector<int64_t>(n));
  for (auto& l : lists) for (auto& p : l) p = (int64_t)(rng() % (uint64_t)ndocs);
  std::vector<double> tot, tsync, tasync;
  for (int i = 0; i < calls + 50; ++i) {
    const auto& l = lists[i % lists.size()];
    const double t0 = now_us();
    memcpy(pin, l.data(), n * 8);
    const int rc = maxsim_rank_forward(&iv, dq, MAXSIM_F32, lq, pin, n, k, ws, op, os, flag, 1, st);
    const double t1 = now_us();
    if (rc != 0) { printf("maxsim_rank_forward -> %d\n", rc); return 4; }
    if (i >= 50) tot.push_back(t1 - t0);
  }
  for (int i = 0; i < calls / 2 + 50; ++i) {  // the same call waited for with the runtime instead of the polled word
    const auto& l = lists[i % lists.size()];
    const double t0 = now_us();
    memcpy(pin, l.data(), n * 8);
    const int rc = maxsim_rank_forward(&iv, dq, MAXSIM_F32, lq, pin, n, k, ws, op, os, nullptr, 1, st);
    const double t1 = now_us();
    if (rc != 0) return 5;
    if (i >= 50) tsync.push_back(t1 - t0);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < calls / 2 + 50; ++i) {  // GPU span: the launch(es) of one call between two events, nothing else in flight
    const auto& l = lists[i % lists.size()];
    memcpy(pin, l.data(), n * 8);
    CK(hipEventRecord(e0, st));
    const int rc = maxsim_rank_forward(&iv, dq, MAXSIM_F32, lq, pin, n, k, ws, op, os, flag, 0, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    if (rc != 0) return 6;
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 50) tasync.push_back(ms * 1e3);
  }
  printf("%s index of %lld docs x %d tokens, 1 query x %d candidates -> top-%d, %d calls; MAXSIM_FUSED=%s\n", dts, (long long)ndocs, ld, n, k, calls,
         getenv("MAXSIM_FUSED") ? getenv("MAXSIM_FUSED") : "(default)");
  report("memcpy to pinned + maxsim_rank_forward(sync, polled word)", tot);
  report("the same, waited for with hipStreamSynchronize", tsync);
  report("GPU span of one call's launches (events)", tasync);
  return 0;
}
