// Feasibility / latency probe for stream-side conditional hand-offs (hipStreamWaitValue32, Beta API):
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_wait.hip -o /tmp/stream_wait && /tmp/stream_wait
// A: one stream: k_set(flag) ; wait(flag) ; k_nop      -- cost of a wait that is already satisfied
// B: two streams: S: k_set(need) ; wait(go) ; k_nop      I: wait(need) ; k_nop x n ; k_set(go)  -- a hand-off and back
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_set(unsigned* p, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_nop(unsigned* p) { if (p && threadIdx.x == 9999) *p = 0; }
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  unsigned *need = nullptr, *go = nullptr, *plain = nullptr;
  CK(hipExtMallocWithFlags((void**)&need, 8, hipMallocSignalMemory));
  CK(hipExtMallocWithFlags((void**)&go, 8, hipMallocSignalMemory));
  CK(hipMalloc((void**)&plain, 8));
  CK(hipMemset(need, 0, 8)); CK(hipMemset(go, 0, 8)); CK(hipMemset(plain, 0, 8));
  hipStream_t S, I;
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&I, hipStreamNonBlocking));
  const int N = 200;
  for (int variant = 0; variant < 4; variant++) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= N; i++) {
      const unsigned t = (unsigned)(variant * 1000 + i);
      if (variant == 0) {  // baseline: three kernels back to back
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, S, need, t);
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
      } else if (variant == 1) {  // A
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, S, need, t);
        CK(hipStreamWaitValue32(S, need, t, hipStreamWaitValueGte));
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
      } else if (variant == 2) {  // B, other stream idles through 3 nops
        CK(hipStreamWaitValue32(I, need, t, hipStreamWaitValueGte));
        for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, I, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, I, go, t);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, S, need, t);
        CK(hipStreamWaitValue32(S, go, t, hipStreamWaitValueGte));
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
      } else {  // C: the decision kernel itself releases S (go written by S's own kernel), I still runs its chain
        CK(hipStreamWaitValue32(I, need, t, hipStreamWaitValueGte));
        for (int k = 0; k < 12; k++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, I, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, S, go, t);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, S, need, t);
        CK(hipStreamWaitValue32(S, go, t, hipStreamWaitValueGte));
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, S, (unsigned*)nullptr);
      }
    }
    CK(hipStreamSynchronize(S));
    CK(hipStreamSynchronize(I));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("variant %d: %.2f us per iteration\n", variant, us / N);
  }
  // plain device memory as the wait target?
  hipError_t e = hipStreamWaitValue32(S, plain, 0, hipStreamWaitValueGte);
  printf("wait on plain hipMalloc memory: %s\n", hipGetErrorString(e));
  CK(hipStreamSynchronize(S));
  printf("done\n");
  return 0;
}
