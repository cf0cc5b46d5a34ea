// Can a sweep read its rows with direct 2-byte gathers (lane i takes compact positions i, i+64, ...) instead of
// staging the whole row in LDS?  R rows x D bf16, C = D/2 random sorted columns; every wave handles rows w, w+W, ...
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/gather_read.hip -o scripts/ubench/gather_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int NPL = 28;
template <int PREFETCH>
__global__ __launch_bounds__(256) void k_gather(const uint16_t* __restrict__ x, int R, int D, const int* __restrict__ cols,
                                                int C, float* __restrict__ out, int rows_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int coff[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { const int p = i * 64 + lane; coff[i] = cols[p < C ? p : C - 1]; }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  float acc = 0.f;
  for (int r = r0 + wave; r < r1; r += 4) {
    const uint16_t* row = x + size_t(r) * D;
    uint32_t v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = row[coff[i]];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float f = __uint_as_float(v[i] << 16); s = fmaf(f, f, s); }
    acc += s;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;       // keep the loads alive
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}
int main() {
  const int F = 128, N = 196, D = 3584, R = F * N, C = D / 2;
  uint16_t* x; hipMalloc(&x, size_t(R) * D * 2); hipMemset(x, 0x3c, size_t(R) * D * 2);
  std::vector<int> all(D); for (int i = 0; i < D; ++i) all[i] = i;
  srand(1); std::random_shuffle(all.begin(), all.end()); std::vector<int> cols(all.begin(), all.begin() + C); std::sort(cols.begin(), cols.end());
  int* dc; hipMalloc(&dc, C * 4); hipMemcpy(dc, cols.data(), C * 4, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rpw : {25, 49, 98}) {
    const int grid = (R + rpw - 1) / rpw;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, x, R, D, dc, C, out, rpw);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("direct 2-byte gathers, %d rows per workgroup (%d workgroups): %.1f us per sweep = %.2f TB/s of row bytes\n", rpw, grid, ms * 100.f,
                           double(R) * D * 2 / (ms * 1e-4) / 1e12 / 1e0 * 1e-0);
    }
  }
  return 0;
}
