// Does hipExtAnyOrderLaunch overlap consecutive kernels of ONE stream on gfx950?  P -> A (1 WG, long) -> B (flag) -> C
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/anyorder.hip -o scripts/ubench/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_p(int* buf, int n, int v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] = v; }
__global__ void k_spin(const int* in, int* out, long cycles, int tag) {
  long t0 = __builtin_readcyclecounter();
  int guard = 0;
  while (__builtin_readcyclecounter() - t0 < cycles && ++guard < (1 << 24)) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = in[blockIdx.x] + tag;
}
__global__ void k_c(const int* a, const int* b, int* out) { out[0] = a[0] * 1000 + b[0]; }
int main() {
  int *p, *a, *b, *c;
  hipMalloc(&p, 4096 * 4); hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&c, 64);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, st);
      for (int it = 0; it < 10; ++it) {
        hipLaunchKernelGGL(k_p, dim3(4), dim3(1024), 0, st, p, 4096, it + 1);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, (const int*)p, a, 100000L, 10);      // ~50 us of 100 MHz ticks?
        hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0,
                              (const int*)p, b, 60000L, 20);
        hipLaunchKernelGGL(k_c, dim3(1), dim3(1), 0, st, (const int*)a, (const int*)b, c);
      }
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int h; hipMemcpy(&h, c, 4, hipMemcpyDeviceToHost);
      printf("flags=%d rep=%d: %.1f us per iteration, c=%d (want %d)\n", flags, rep, ms * 100.f, h, (10 + 10) * 1000 + 10 + 20);
    }
  }
  return 0;
}
