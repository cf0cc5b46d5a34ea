// Does the ACCESS PATTERN of sweep 1 cost bandwidth?  Same bytes (128x196x3584 bf16, or a larger clip with argv[1] frames),
// 16 B / lane, 8 loads in flight per lane, three ways to cut the tensor:
//   contiguous   grid-stride over the flat tensor (stream_read.hip)
//   slab         sweep 1 today: workgroup = (1 KB column slab, row group), 4 waves, wave w takes rows r0 + w, + 4, ...
//   rows         workgroup = 14 waves = 7 slabs x 2 row phases: a workgroup reads whole 7 KB rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
__global__ __launch_bounds__(256) void k_contig(const uint4* __restrict__ x, size_t n16, unsigned* out) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  unsigned acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { uint4 v = x[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// rows of `pitch16` uint4; block (bx, gy): slab bx (64 lanes x 16 B), rows [gy * rpg, (gy + 1) * rpg), wave w: r0 + w + 4 i
__global__ __launch_bounds__(256) void k_slab(const uint4* __restrict__ x, int pitch16, int rpg, int R, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rpg, r1 = min(R, r0 + rpg);
  const uint4* base = x + size_t(blockIdx.x) * 64 + lane;
  unsigned acc = 0;
  for (int r = r0 + wave; r < r1; r += 4 * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int rr = r + 4 * u; v[u] = rr < r1 ? base[size_t(rr) * pitch16] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// block gy: rows [gy * rpg, ...), NS slabs x NP row phases waves: wave = phase * NS + slab
template <int NS, int NP>
__global__ __launch_bounds__(NS * NP * 64) void k_rows(const uint4* __restrict__ x, int pitch16, int rpg, int R, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slab = wave % NS, ph = wave / NS;
  const int r0 = blockIdx.x * rpg, r1 = min(R, r0 + rpg);
  const uint4* base = x + size_t(slab) * 64 + lane;
  unsigned acc = 0;
  for (int r = r0 + ph; r < r1; r += NP * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int rr = r + NP * u; v[u] = rr < r1 ? base[size_t(rr) * pitch16] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <typename L> void timeit(const char* name, size_t bytes, L launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.1f us per pass, %.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
  const int F = argc > 1 ? atoi(argv[1]) : 128, N = 196, D = 3584;
  const int R = F * N, pitch16 = D * 2 / 16;
  const size_t bytes = size_t(R) * D * 2, n16 = bytes / 16;
  uint4* x; unsigned* out; hipMalloc(&x, bytes); hipMalloc(&out, 64); hipMemset(x, 1, bytes);
  printf("%d frames: %.1f MB\n", F, bytes / 1e6);
  for (int blocks : {4096, 8192, 16384})
    timeit(("contiguous " + std::to_string(blocks)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_contig, dim3(blocks), dim3(256), 0, 0, x, n16, out); });
  for (int rpg : {196, 98, 49})
    timeit(("slab rows/group " + std::to_string(rpg)).c_str(), bytes, [&] { hipLaunchKernelGGL(k_slab, dim3(7, (R + rpg - 1) / rpg), dim3(256), 0, 0, x, pitch16, rpg, R, out); });
  for (int rpg : {98, 49, 28})
    timeit(("rows 7x2 rows/group " + std::to_string(rpg)).c_str(), bytes, [&] { hipLaunchKernelGGL((k_rows<7, 2>), dim3((R + rpg - 1) / rpg), dim3(7 * 2 * 64), 0, 0, x, pitch16, rpg, R, out); });
  for (int rpg : {49, 28, 14})
    timeit(("rows 7x1 rows/group " + std::to_string(rpg)).c_str(), bytes, [&] { hipLaunchKernelGGL((k_rows<7, 1>), dim3((R + rpg - 1) / rpg), dim3(7 * 64), 0, 0, x, pitch16, rpg, R, out); });
  return 0;
}
