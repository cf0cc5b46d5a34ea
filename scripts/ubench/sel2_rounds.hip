// Micro-benchmark of one partition round of vc2_select2.h: ticks (s_memtime) per call for NW x EQ x range length,
// plus the primitives it is made of.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o sel2_rounds sel2_rounds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../vidcom2_amd/csrc/vc2_select2.h"
using namespace vc2;

template <typename W, int NW, int EQ>
__global__ __launch_bounds__(256) void k_round(const W* gw, int n, int lo, int hi, int reps, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Sel2<W> S = sel2_carve<W>(smem, 4096);
  const int tid = threadIdx.x;
  unsigned long long acc = 0;
  int cut = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = tid; i < n; i += 256) S.w[i] = gw[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (NW > 1 || tid < 64) cut = sel2_partition<W, NW>(S, lo, hi, S.la, S.lb, S.stage, tid);
    const unsigned long long t1 = __builtin_readcyclecounter();
    acc += t1 - t0;
    if (r == 0 && tid == 0) out[2] = t1 - t0;
    __syncthreads();
  }
  if (tid == 0) { out[0] = acc / reps; out[1] = cut; }
}

__global__ void k_prims(unsigned long long* out, uint32_t* sink) {
  __shared__ uint32_t lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = i * 7 + 1;
  __syncthreads();
  uint32_t v = tid;
  unsigned long long t0, t1;
  // 1: 32 dependent wave scans
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) v = wave_incl_scan_u32(v) & 0xFFFF;
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = (t1 - t0) / 32;
  // 2: 32 dependent wave mins
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) v = wave_min_bcast_u32(v + i) + tid;
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[1] = (t1 - t0) / 32;
  // 3: 32 dependent LDS reads (pointer chase)
  t0 = __builtin_readcyclecounter();
  uint32_t a = v & 4095;
  for (int i = 0; i < 32; ++i) a = lds[a] & 4095;
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[2] = (t1 - t0) / 32;
  // 4: 32 barriers
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) { __syncthreads(); }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[3] = (t1 - t0) / 32;
  // 5: 256 dependent v_add (VALU latency)
  t0 = __builtin_readcyclecounter();
  uint32_t b = a;
#pragma unroll
  for (int i = 0; i < 256; ++i) b = b * 3 + i;
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[4] = (t1 - t0);
  // 6: LDS write then dependent read (same wave)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) { lds[(tid + i) & 4095] = b; b = lds[(tid * 5 + i) & 4095] + 1; }
  t1 = __builtin_readcyclecounter();
  if (tid == 0) out[5] = (t1 - t0) / 32;
  sink[tid] = v + a + b;
}

template <typename W, int NW, int EQ>
void run(const char* name, const W* dw, int n, int lo, int hi, unsigned long long* dout) {
  unsigned long long h[3];
  hipLaunchKernelGGL((k_round<W, NW, EQ>), dim3(1), dim3(256), sel2_bytes(4096, sizeof(W)) + 64, 0, dw, n, lo, hi, 20, dout);
  hipError_t e = hipGetLastError();
  hipMemcpy(h, dout, 24, hipMemcpyDeviceToHost);
  printf("%-10s NW=%d len=%5d : avg %7llu ticks, first (cold) call %7llu  (cut %llu) %s\n", name, NW, hi - lo, h[0], h[2], h[1], e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
  const int n = 4096;
  std::vector<uint32_t> w32(n);
  std::vector<uint64_t> w64(n);
  srand(1);
  for (int i = 0; i < n; ++i) { uint32_t key = (rand() % 800); w32[i] = (key << 13) | i; w64[i] = (uint64_t(key) << 32) | i; }
  uint32_t* d32; uint64_t* d64; unsigned long long* dout; uint32_t* sink;
  hipMalloc(&d32, n * 4); hipMalloc(&d64, n * 8); hipMalloc(&dout, 64); hipMalloc(&sink, 4096);
  hipMemcpy(d32, w32.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d64, w64.data(), n * 8, hipMemcpyHostToDevice);
  hipSetDevice(0);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint32_t, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_round<uint64_t, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  for (int rep = 0; rep < 2; ++rep) {
    run<uint32_t, 1, 1>("u32", d32, n, 0, 16, dout);
    run<uint32_t, 1, 1>("u32", d32, n, 0, 64, dout);
    run<uint32_t, 1, 1>("u32", d32, n, 0, 200, dout);
    run<uint32_t, 1, 1>("u32", d32, n, 3, 256, dout);
    run<uint32_t, 1, 2>("u32", d32, n, 0, 500, dout);
    run<uint32_t, 1, 4>("u32", d32, n, 0, 1000, dout);
    run<uint32_t, 1, 8>("u32", d32, n, 0, 2000, dout);
    run<uint32_t, 4, 1>("u32", d32, n, 0, 1000, dout);
    run<uint32_t, 4, 2>("u32", d32, n, 0, 2000, dout);
    run<uint32_t, 4, 4>("u32", d32, n, 0, 3584, dout);
    run<uint32_t, 4, 8>("u32", d32, n, 0, 4000, dout);
    run<uint64_t, 4, 8>("u64", d64, n, 0, 3584, dout);
  }
  unsigned long long h[8];
  hipLaunchKernelGGL(k_prims, dim3(1), dim3(256), 0, 0, dout, sink);
  hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost);
  printf("prims (ticks): scan %llu  min %llu  lds_read %llu  barrier(4 waves) %llu  256 dependent mad %llu  lds wr+rd %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  // tick calibration: a timed empty-ish kernel of known tick count
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_round<uint32_t, 4, 4>), dim3(1), dim3(256), sel2_bytes(4096, 4) + 64, 0, d32, n, 0, 3584, 2000, dout);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
  printf("calibration: 2000 reps of NW=4 EQ=4 len 3584: %.1f us wall per rep (incl. reload), %llu ticks per partition\n", ms * 1000 / 2000, h[0]);
  return 0;
}
