// VALU issue rates of the instructions sweep 3 is made of (gfx950): cycles per wave64 instruction per SIMD with 8 waves
// per SIMD and 8 independent chains per wave, plus what v_dot2c_f32_bf16 does to subnormals / rounding.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o scripts/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
  float v[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; p[i] = (f2){v[i], v[i] + 1.f}; }
  const float c = seed * 0.5f;
  const f2 cp = (f2){c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (OP == 0) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
        if constexpr (OP == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[i]));
        if constexpr (OP == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
        if constexpr (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
        if constexpr (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 7) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(seed));
        if constexpr (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 9) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
        if constexpr (OP == 10) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
        if constexpr (OP == 11) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 12) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 13) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(seed));
        if constexpr (OP == 14) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(seed));
        if constexpr (OP == 15) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if constexpr (OP == 16) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&p[i])) : "v"(*reinterpret_cast<const double*>(&cp)));
        if constexpr (OP == 17) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*reinterpret_cast<double*>(&p[i])) : "v"(v[i]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  if (s == 12345.f) out[0] = s;
}

__global__ void k_dot2h(const uint32_t* a, const float* c, float* o, uint32_t* o2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  union { uint32_t u; h2_t h; } x, one, r; x.u = a[i]; one.u = 0x3c003c00u;
  o[i] = __builtin_amdgcn_fdot2(x.h, one.h, c[i], false);
  r.h = x.h * x.h;                      // v_pk_mul_f16: subnormal products
  o2[i] = r.u;
  r.h = x.h - one.h * (_Float16)0.0f;   // v_pk_add: keeps subnormals?
  o2[i + n] = r.u;
}
__global__ void k_dot2(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = c[i];
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a[i]), "v"(b[i]));
  o[i] = acc;
}

template <int OP> void run(const char* name, float* out) {
  const int iters = 2000, blocks = 256 * 8;       // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = double(iters) * 32 * 8;         // per wave x 8 waves per SIMD
  printf("%-22s %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms * 1e6 / instr_per_simd,
         ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 64);
  run<0>("v_lshlrev_b32", out); run<1>("v_and_b32", out); run<2>("v_sub_f32", out); run<3>("v_mul_f32", out);
  run<4>("v_pk_add_f32", out); run<5>("v_pk_mul_f32", out); run<6>("v_cvt_pk_bf16_f32", out);
  run<7>("v_dot2c_f32_bf16", out); run<8>("v_fma_f32", out); run<9>("v_add_f32_dpp", out);
  run<10>("v_cvt_f32_f16", out); run<11>("v_pk_mul_f16", out); run<12>("v_pk_add_f16", out); run<13>("v_dot2c_f32_f16", out);
  run<14>("v_perm_b32", out); run<15>("v_cvt_pk_f16_f32", out); run<16>("v_add_f64", out); run<17>("v_cvt_f64_f32", out);
  // dot2c semantics: subnormal bf16 inputs, rounding
  const int n = 8;
  // a = (lo, hi) bf16 pairs; b = (1.0, 1.0)
  uint32_t ha[n] = {0x00010001u, 0x3f803f80u, 0x3f800001u, 0x00400040u, 0x3f813f81u, 0x33803f80u, 0x007f3f80u, 0x3f803380u};
  uint32_t hb[n]; for (int i = 0; i < n; ++i) hb[i] = 0x3f803f80u;
  float hc[n] = {0.f, 0.f, 0.f, 0.f, 16777216.f, 1.f, 0.f, 1.f};
  uint32_t *da, *db; float *dc, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dot2, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
  float ho[n]; hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) {
    uint32_t bits; memcpy(&bits, &ho[i], 4);
    auto bf = [](uint32_t h) { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; };
    const double exact = double(bf(ha[i] & 0xffff)) + double(bf(ha[i] >> 16)) + double(hc[i]);
    printf("dot2c(a=%08x, ones, c=%g) = %.9g (bits %08x)   exact %.17g\n", ha[i], hc[i], ho[i], bits, exact);
  }
  {
    // f16: dot2 with subnormal inputs (0x0001 = 2^-24, 0x03ff largest subnormal), pk_mul into the subnormal range
    uint32_t hh[n] = {0x00010001u, 0x03ff03ffu, 0x3c000001u, 0x0c000c00u, 0x10001000u, 0x14001400u, 0x00013c00u, 0x1c001c00u};
    float cc[n] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    uint32_t* o2; hipMalloc(&o2, 2 * n * 4);
    hipMemcpy(da, hh, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, cc, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dot2h, dim3(1), dim3(64), 0, 0, da, dc, dout, o2, n);
    uint32_t h2o[2 * n]; hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2o, o2, 2 * n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("f16 dot2(a=%08x, ones, c=%g) = %.9g | pk_mul(a,a) = %08x | pk_add(a,-0) = %08x\n", hh[i], cc[i], ho[i], h2o[i], h2o[i + n]);
  }
  return 0;
}
