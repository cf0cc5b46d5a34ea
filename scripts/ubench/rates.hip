// VALU issue-rate micro-benchmark (gfx950): cycles per wave64 instruction for the ops of the sweep loops.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP> __global__ void k(float* out, int iters, float seed, double dseed) {
  float a[8]; double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; d[i] = dseed + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = a[i] * 1.0001f;                                        // v_mul_f32
        if (OP == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        if (OP == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        if (OP == 3) d[i] = d[i] * 1.0000001;                                      // v_mul_f64
        if (OP == 4) d[i] = d[i] + 1.5;                                            // v_add_f64
        if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(a[i]) : "v"(a[i]));
        if (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d[i]) : "v"(d[i]));
        if (OP == 7) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a[i]) : "v"(a[i]));
        if (OP == 8) d[i] = fma(d[i], 1.0000001, 0.5);                             // v_fma_f64
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + float(d[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* out) {
  const int iters = 200, blocks = 256 * 4, threads = 256;     // 4 waves per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 2, 1.f, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: (blocks*4 waves / (256 CUs * 4 SIMDs)) waves, each iters*REP*8 instructions
  const double waves_per_simd = double(blocks) * 4 / (256.0 * 4);
  const double instr = waves_per_simd * iters * REP * 8;
  printf("%-18s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 1 << 22);
  run<0>("v_mul_f32", out); run<1>("v_cvt_f64_f32", out); run<2>("v_cvt_f32_f64", out); run<3>("v_mul_f64", out);
  run<4>("v_add_f64", out); run<8>("v_fma_f64", out); run<5>("v_cvt_pk_bf16_f32", out); run<6>("v_pk_mul_f32", out);
  run<7>("v_lshlrev_b32", out);
  return 0;
}
