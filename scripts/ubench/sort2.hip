// Micro-benchmark of the std::sort replay (introsort2) and of the small partition rounds, in cycles (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define VC2_SEL2_DEBUG 1
#include "../../vidcom2_amd/csrc/vc2_select2.h"
using namespace vc2;

template <int NW, int SOLO, int COOP>
__global__ __launch_bounds__(64 * NW) void k_sort(const uint32_t* gw, int n, int reps, unsigned long long* out, int* order, int ta, int tb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Sel2<uint32_t> S = sel2_carve<uint32_t>(smem, n + 1);
  unsigned char* p = smem + (sel2_bytes(n + 1, 4) + 15) / 16 * 16;
  SortScratch2 Q = sort2_carve(p, NW);
  int* ord = reinterpret_cast<int*>(p + (sort2_bytes(n + 1, 16) + 15) / 16 * 16);
  const int tid = threadIdx.x;
  unsigned long long acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = tid; i < n; i += 64 * NW) S.w[i] = gw[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    introsort2<uint32_t, NW, SOLO, COOP>(S, Q, n, [&](int r, int i) { ord[r] = i; }, tid, ta, tb);
    const unsigned long long t1 = __builtin_readcyclecounter();
    acc += t1 - t0;
    __syncthreads();
  }
  if (tid == 0) out[0] = acc / reps;
  for (int i = tid; i < n; i += 64 * NW) order[i] = ord[i];
}
template <int EQ>
__global__ __launch_bounds__(64) void k_small(const uint32_t* gw, int n, int lo, int hi, int reps, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Sel2<uint32_t> S = sel2_carve<uint32_t>(smem, n);
  const int tid = threadIdx.x;
  unsigned long long acc = 0; int cut = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = tid; i < n; i += 64) S.w[i] = gw[i];
    wave_lds_order();
    const unsigned long long t0 = __builtin_readcyclecounter();
    cut = sel2_partition_t<uint32_t, 1, EQ>(S, lo, hi, S.la, S.lb, tid);
    const unsigned long long t1 = __builtin_readcyclecounter();
    acc += t1 - t0;
  }
  if (tid == 0) { out[0] = acc / reps; out[1] = cut; }
}
int main() {
  const int n = 1791;
  std::vector<uint32_t> w(4096);
  srand(1);
  for (int i = 0; i < 4096; ++i) { uint32_t key = (rand() % 400); w[i] = (key << 13) | i; }
  uint32_t* d; unsigned long long* dout; int* dord;
  hipMalloc(&d, 4096 * 4); hipMalloc(&dout, 64); hipMalloc(&dord, 4096 * 4);
  hipMemcpy(d, w.data(), 4096 * 4, hipMemcpyHostToDevice);
  unsigned long long h[2];
  size_t smem = sel2_bytes(n + 1, 4) + sort2_bytes(n + 1, 16) + (n + 1) * 4 + 256;
  auto run = [&](int nw, int part, int parts) {
    unsigned long long z[128] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(vc2::g_sel2_dbg), z, sizeof(z));
    const int ta = part, tb = 31 - __builtin_clz(parts);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    hipEventRecord(e0, 0);
    if (nw == 16) hipLaunchKernelGGL((k_sort<16, 2, 2>), dim3(1), dim3(1024), smem, 0, d, n, 20, dout, dord, ta, tb);
    else hipLaunchKernelGGL((k_sort<4, 4, 4>), dim3(1), dim3(256), smem, 0, d, n, 20, dout, dord, ta, tb);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
    unsigned long long dbg[128]; hipMemcpyFromSymbol(dbg, HIP_SYMBOL(vc2::g_sel2_dbg), sizeof(dbg));
    unsigned long long pc = 0, pn = 0; for (int w = 0; w < 16; ++w) { pc += dbg[16 + 2 * w]; pn += dbg[17 + 2 * w]; }
    printf("introsort2 NW=%2d slice %d/%d: %6llu cycles per sort (wall %.1f us/rep) | init %llu phaseA %llu phaseB %llu final %llu | coop %llu (%llu cyc) wave0 own %llu (%llu cyc)\n",
           nw, part, parts, h[0], ms * 1000 / 20, dbg[1]-dbg[0], dbg[2]-dbg[1], dbg[3]-dbg[2], dbg[5]-dbg[3], dbg[7], dbg[8], dbg[9], dbg[10]);
  };
  run(16, 0, 1); run(16, 0, 1); run(16, 1, 4); run(16, 3, 8); run(4, 0, 1); run(4, 1, 4); run(4, 3, 8); run(4, 7, 16);
  hipLaunchKernelGGL((k_small<0>), dim3(1), dim3(64), sel2_bytes(4096, 4), 0, d, 4096, 0, 40, 50, dout);
  hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost); printf("partition E=1  len 40 : %llu cycles (cut %llu)\n", h[0], h[1]);
  hipLaunchKernelGGL((k_small<0>), dim3(1), dim3(64), sel2_bytes(4096, 4), 0, d, 4096, 100, 120, 50, dout);
  hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost); printf("partition E=1  len 20 : %llu cycles (cut %llu)\n", h[0], h[1]);
  hipLaunchKernelGGL((k_small<1>), dim3(1), dim3(64), sel2_bytes(4096, 4), 0, d, 4096, 0, 200, 50, dout);
  hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost); printf("partition E=4  len 200: %llu cycles (cut %llu)\n", h[0], h[1]);
  return 0;
}
