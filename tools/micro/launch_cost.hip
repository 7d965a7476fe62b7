// launch_cost.hip -- what the HIP launch path costs on this box, as seen by a host thread that waits by polling a
// host-coherent word the (last) kernel stores to: the floor under any "one small call" latency (rank_forward).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o tools/micro/launch_cost && tools/micro/launch_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_empty(int) {}
__global__ void k_flag(uint32_t* flag, uint32_t ticket) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// reads n int64 from (pinned host) memory first, like the rerank kernel's candidate list
__global__ void k_read_flag(const int64_t* src, int n, int64_t* sink, uint32_t* flag, uint32_t ticket) {
  int64_t v = 0;
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) v += src[i];
  if (v == 0x7fffffffffffLL) sink[0] = v;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void report(const char* what, std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  printf("%-78s median %6.2f us  min %6.2f  p90 %6.2f\n", what, v[v.size() / 2], v[0], v[v.size() * 9 / 10]);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t* flag;
  CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent));
  *flag = 0;
  int64_t* pin;
  CK(hipHostMalloc((void**)&pin, 8000, hipHostMallocDefault));
  for (int i = 0; i < 1000; ++i) pin[i] = i;
  int64_t* dsink;
  CK(hipMalloc((void**)&dsink, 64));
  volatile uint32_t* f = flag;
  uint32_t ticket = 0;
  const int N = 400;
  std::vector<double> v;
  auto wait = [&](uint32_t t) { while (*f != t) __builtin_ia32_pause(); };
  for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, ++ticket); wait(ticket); }

  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, 0); v.push_back(now_us() - t0); CK(hipStreamSynchronize(st)); }
  report("host cost of one hipLaunchKernelGGL (empty kernel, idle stream)", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, ++ticket); wait(ticket); v.push_back(now_us() - t0); }
  report("launch -> host sees the kernel's flag (1 kernel, polled)", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, ++ticket); CK(hipStreamSynchronize(st)); v.push_back(now_us() - t0); }
  report("launch -> hipStreamSynchronize returns (1 kernel)", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_empty, dim3(250), dim3(256), 0, st, 0); hipLaunchKernelGGL(k_flag, dim3(63), dim3(256), 0, st, flag, ++ticket); wait(ticket); v.push_back(now_us() - t0); }
  report("2 dependent kernels (250 + 63 workgroups) -> flag polled", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_read_flag, dim3(250), dim3(256), 0, st, pin, 1000, dsink, flag, ++ticket); wait(ticket); v.push_back(now_us() - t0); }
  report("1 kernel that first reads 1000 int64 from pinned host memory -> flag polled", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_read_flag, dim3(250), dim3(256), 0, st, (const int64_t*)dsink, 8, dsink, flag, ++ticket); wait(ticket); v.push_back(now_us() - t0); }
  report("the same kernel reading device memory instead -> flag polled", v);

  // a pre-instantiated graph of the two kernels, ticket read from memory so the graph replays unchanged
  // (here: the flag kernel takes the ticket by value, so re-instantiate per ticket is avoided by always using ticket 7
  //  and resetting the flag on the host before each launch)
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_empty, dim3(250), dim3(256), 0, st, 0);
  hipLaunchKernelGGL(k_flag, dim3(63), dim3(256), 0, st, flag, 7u);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) { *f = 0; CK(hipGraphLaunch(ge, st)); wait(7u); }
  v.clear();
  for (int i = 0; i < N; ++i) { *f = 0; double t0 = now_us(); CK(hipGraphLaunch(ge, st)); wait(7u); v.push_back(now_us() - t0); }
  report("hipGraphLaunch of the same 2 kernels -> flag polled", v);
  v.clear();
  for (int i = 0; i < N; ++i) { *f = 0; double t0 = now_us(); CK(hipGraphLaunch(ge, st)); v.push_back(now_us() - t0); wait(7u); }
  report("host cost of that hipGraphLaunch", v);
  v.clear();
  for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(k_empty, dim3(250), dim3(256), 0, st, 0); double t1 = now_us(); hipLaunchKernelGGL(k_flag, dim3(63), dim3(256), 0, st, flag, ++ticket); double t2 = now_us(); wait(ticket); v.push_back(t2 - t1); (void)t0; }
  report("host cost of the SECOND hipLaunchKernelGGL of a pair", v);
  return 0;
}
