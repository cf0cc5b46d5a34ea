// Round-3 design probe: what does the pass cost when sweep 2 MATERIALISES the normalised, channel-compacted tokens
// (xh = RN_T(x[:, cols] / ||.||), R x C in T) so that the centres and sweep 3 stream 90 MB of clean rows instead of
// gathering from the 180 MB X again?  Prototypes (bf16, 128 x 196 x 3584, C = 1792) with realistic arithmetic:
//   k_s1     stream X once (stands in for sweep 1; sets the cache state)
//   k_xhat   per row: LDS-DMA the row, gather the selected elements (pair mapping), norm, x^, dword stores of x^
//   k_cen    column-owner threads walk rows in order: fp32 chains in torch's cascade blocking (frame + video)
//   k_dist3  half a wave per row, 7 x 16 B per lane, centres from LDS, both distances
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/pipe3.hip -o scripts/ubench/pipe3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  union { b2_t h; uint32_t u; } c;
  c.h = __builtin_convertvector((f2_t){a, b}, b2_t);
  return c.u;
}
__device__ __forceinline__ f2_t rn2(f2_t v) { return __builtin_convertvector(__builtin_convertvector(v, b2_t), f2_t); }
__device__ __forceinline__ f2_t unpack2(uint32_t w) { return (f2_t){__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)}; }

__global__ __launch_bounds__(256) void k_s1(const uint4* __restrict__ x, size_t n16, unsigned* out) {
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

// ---- sweep 2': lane owns compact positions 128*(i>>1) + 2*lane + (i&1), i < NPL ----
constexpr int NPL = 28;
template <int DB>
__global__ __launch_bounds__(256) void k_xhat(const uint16_t* __restrict__ x, int R, int D, const int* __restrict__ cols, int C,
                                              uint32_t* __restrict__ xh, float* __restrict__ den, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t rowb = size_t(D) * 2 + 16;
  unsigned char* buf0 = smem + size_t(wave) * (DB ? 2 : 1) * rowb;
  unsigned char* buf1 = buf0 + rowb;
  int coff[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { const int p = 128 * (i >> 1) + 2 * lane + (i & 1); coff[i] = cols[p < C ? p : C - 1]; }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  const int nch = (D / 8 + 63) >> 6;
  auto issue = [&](int r, unsigned char* buf) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(x) + size_t(r) * D * 2;
    for (int j = 0; j < nch; ++j) {
      const int cv = j * 64 + lane;
      if (cv < D / 8) __builtin_amdgcn_global_load_lds((glb_void_t*)(src + size_t(cv) * 16), (lds_void_t*)(buf + j * 1024), 16, 0, 0);
    }
  };
  int r = r0 + wave;
  if (DB && r < r1) issue(r, buf0);
  for (; r < r1; r += 4) {
    if (!DB) issue(r, buf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (DB && r + 4 < r1) issue(r + 4, buf1);
    float xv[NPL];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      xv[i] = __uint_as_float(uint32_t(reinterpret_cast<const uint16_t*>(buf0)[coff[i]]) << 16);
      if (i & 1) s1 = fmaf(xv[i], xv[i], s1); else s0 = fmaf(xv[i], xv[i], s0);
    }
    const float nrm = sqrtf(wave_sum_f32(s0 + s1));
    const float dn = float(__bf16(fmaxf(float(__bf16(nrm)), 1e-12f)));
    const float rc = __builtin_amdgcn_rcpf(dn);
    if (lane == 0) den[r] = dn;
    uint32_t* orow = xh + size_t(r) * (C / 2);
#pragma unroll
    for (int i = 0; i < NPL; i += 2) orow[64 * (i >> 1) + lane] = pack_bf16(xv[i] * rc, xv[i + 1] * rc);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (DB) { unsigned char* t = buf0; buf0 = buf1; buf1 = t; }
  }
}

// ---- centres: wave = (row run of one frame part, slab of 64 16-byte column vectors) ----
// run = rows [a, b) of a frame (frame chain: blocks of 16 from the frame start) plus the rows up to the end of the last
// video block (16 global rows) that starts inside [a, b).  fb[f][j][C] / vb[v][C] = level-0 block sums.
template <int HALVES>
__global__ __launch_bounds__(256) void k_cen(const uint4* __restrict__ xh, int F, int N, int C, float* __restrict__ fb,
                                             float* __restrict__ vb, int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C / 8;
  const int cv = wave * 64 + lane;                     // 4 waves = 4 slabs of 64 vectors (the last one partly idle)
  if (cv >= nvec) return;
  const int f = blockIdx.x / HALVES, h = blockIdx.x % HALVES;
  const int nblk = (N + 15) / 16;                      // frame blocks incl. the tail block
  const int b0 = (nblk * h) / HALVES, b1 = (nblk * (h + 1)) / HALVES;
  const int a = f * N + b0 * 16, b = min((f + 1) * N, f * N + b1 * 16);
  // video blocks starting in [a, b)
  const int v0 = (a + 15) / 16;
  const int vend = min(R, ((b + 15) / 16) * 16);       // rows needed: up to the end of the last such block
  const int rstart = a, rend = max(b, vend);
  float fa[8], va[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = 0.f; va[e] = 0.f; }
  const bool vstarted0 = (a % 16) == 0;
  bool von = vstarted0;
  for (int r = rstart; r < rend; r += 16) {
    uint4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int rr = min(r + u, rend - 1); v[u] = xh[size_t(rr) * nvec + cv]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int rr = r + u;
      if (rr < rend) {
        if ((rr & 15) == 0) {                           // a video block starts
          if (von && rr > a) {
            float* o = vb + (size_t(rr / 16 - 1) * C + size_t(cv) * 8);
            *reinterpret_cast<float4*>(o) = make_float4(va[0], va[1], va[2], va[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(va[4], va[5], va[6], va[7]);
          }
          von = rr < b;                                 // blocks starting at or after b belong to the next run
#pragma unroll
          for (int e = 0; e < 8; ++e) va[e] = 0.f;
        }
        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const bool fin = rr < b;
        if (fin && ((rr - f * N) & 15) == 0 && rr > a) {
          float* o = fb + ((size_t(f) * nblk + (rr - f * N) / 16 - 1) * C + size_t(cv) * 8);
          *reinterpret_cast<float4*>(o) = make_float4(fa[0], fa[1], fa[2], fa[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(fa[4], fa[5], fa[6], fa[7]);
#pragma unroll
          for (int e = 0; e < 8; ++e) fa[e] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f2_t t = unpack2(w[q]);
          if (fin) { fa[2 * q] += t.x; fa[2 * q + 1] += t.y; }
          if (von) { va[2 * q] += t.x; va[2 * q + 1] += t.y; }
        }
      }
    }
  }
  {
    float* o = fb + ((size_t(f) * nblk + (b - 1 - f * N) / 16) * C + size_t(cv) * 8);
    *reinterpret_cast<float4*>(o) = make_float4(fa[0], fa[1], fa[2], fa[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(fa[4], fa[5], fa[6], fa[7]);
    if (von || true) {
      float* o2 = vb + (size_t((rend - 1) / 16) * C + size_t(cv) * 8);
      *reinterpret_cast<float4*>(o2) = make_float4(va[0], va[1], va[2], va[3]);
      *reinterpret_cast<float4*>(o2 + 4) = make_float4(va[4], va[5], va[6], va[7]);
    }
  }
}

// ---- sweep 3': WG = (frame, split); half a wave per row; lane = 7 vectors (l + 32 j); centres in LDS (bf16) ----
constexpr int NV = 7;
__global__ __launch_bounds__(256) void k_dist3(const uint4* __restrict__ xh, int N, int C, int S, int rows_per_split,
                                               const uint16_t* __restrict__ vc, const uint16_t* __restrict__ fc,
                                               float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* cvc = reinterpret_cast<uint4*>(smem);                   // [C/8] video centre
  uint4* cfc = cvc + C / 8;                                      // [C/8] frame centre
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = blockIdx.x / S, sp = blockIdx.x % S;
  const int nvec = C / 8;
  for (int i = tid; i < nvec; i += 256) {
    cvc[i] = reinterpret_cast<const uint4*>(vc)[i];
    cfc[i] = reinterpret_cast<const uint4*>(fc + size_t(f) * C)[i];
  }
  __syncthreads();
  const int n0 = sp * rows_per_split, n1 = min(N, n0 + rows_per_split);
  const int hl = lane & 31, hw = lane >> 5;
  for (int n = n0 + wave * 2 + hw; n < n1 + 1; n += 8) {        // (+1: both halves enter together; guarded below)
    const bool live = n < n1;
    const size_t row = size_t(f) * N + (live ? n : n1 - 1);
    uint4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = xh[row * nvec + hl + 32 * j];
    f2_t acc = (f2_t){0.f, 0.f};                                 // (video, frame)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const uint4 a = cvc[hl + 32 * j], b = cfc[hl + 32 * j];
      const uint32_t xw[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f2_t xx = unpack2(xw[q]), cv2 = unpack2(aw[q]), cf2 = unpack2(bw[q]);
        const f2_t d0 = rn2((f2_t){xx.x - cv2.x, xx.x - cf2.x});   // element lo: (video, frame)
        const f2_t d1 = rn2((f2_t){xx.y - cv2.y, xx.y - cf2.y});
        const f2_t q0 = rn2(d0 * d0), q1 = rn2(d1 * d1);
        acc = acc + q0;
        acc = acc + q1;
      }
    }
    // reduce over the 32 lanes of the half
    float sv = acc.x, sf = acc.y;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); sf += __shfl_xor(sf, o, 64); }
    if (live && hl == 0) { out[row * 2] = sv; out[row * 2 + 1] = sf; }
  }
}

int main() {
  const int F = 128, N = 196, D = 3584, R = F * N, C = D / 2;
  const size_t xbytes = size_t(R) * D * 2;
  uint16_t* x; hipMalloc(&x, xbytes);
  {
    std::vector<uint16_t> h(size_t(R) * D);
    uint32_t s = 12345;
    for (auto& e : h) { s = s * 1664525u + 1013904223u; e = uint16_t(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 31) << 15)); }
    hipMemcpy(x, h.data(), xbytes, hipMemcpyHostToDevice);
  }
  std::vector<int> all(D); for (int i = 0; i < D; ++i) all[i] = i;
  srand(1); std::random_shuffle(all.begin(), all.end());
  std::vector<int> cols(all.begin(), all.begin() + C); std::sort(cols.begin(), cols.end());
  int* dc; hipMalloc(&dc, C * 4); hipMemcpy(dc, cols.data(), C * 4, hipMemcpyHostToDevice);
  uint32_t* xh; hipMalloc(&xh, size_t(R) * C * 2);
  float* den; hipMalloc(&den, R * 4);
  const int nblk = (N + 15) / 16;
  float* fb; hipMalloc(&fb, size_t(F) * nblk * C * 4);
  float* vb; hipMalloc(&vb, size_t(R / 16 + 2) * C * 4);
  uint16_t* vc; hipMalloc(&vc, C * 2); hipMemset(vc, 0x3c, C * 2);
  uint16_t* fc; hipMalloc(&fc, size_t(F) * C * 2); hipMemset(fc, 0x3b, size_t(F) * C * 2);
  float* dout; hipMalloc(&dout, size_t(R) * 8);
  unsigned* junk; hipMalloc(&junk, 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_xhat<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_xhat<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  const int NE = 8;
  hipEvent_t ev[NE]; for (auto& e : ev) hipEventCreate(&e);
  const size_t rowb = size_t(D) * 2 + 16;
  struct Cfg { int db; int rpw; int halves; int S; };
  const Cfg cfgs[] = {{0, 25, 2, 8}, {0, 49, 2, 8}, {1, 49, 2, 8}, {1, 65, 1, 8}, {0, 16, 2, 4}, {0, 32, 1, 12}, {1, 25, 2, 6}};
  for (const Cfg& c : cfgs) {
    double t[4] = {0, 0, 0, 0};
    const int reps = 12;
    for (int rep = 0; rep < reps + 2; ++rep) {
      hipEventRecord(ev[0], 0);
      hipLaunchKernelGGL(k_s1, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(x), xbytes / 16, junk);
      hipEventRecord(ev[1], 0);
      const int g2 = (R + c.rpw - 1) / c.rpw;
      if (c.db) hipLaunchKernelGGL(k_xhat<1>, dim3(g2), dim3(256), 4 * 2 * rowb, 0, x, R, D, dc, C, xh, den, c.rpw);
      else hipLaunchKernelGGL(k_xhat<0>, dim3(g2), dim3(256), 4 * rowb, 0, x, R, D, dc, C, xh, den, c.rpw);
      hipEventRecord(ev[2], 0);
      if (c.halves == 2) hipLaunchKernelGGL(k_cen<2>, dim3(F * 2), dim3(256), 0, 0, reinterpret_cast<const uint4*>(xh), F, N, C, fb, vb, R);
      else hipLaunchKernelGGL(k_cen<1>, dim3(F), dim3(256), 0, 0, reinterpret_cast<const uint4*>(xh), F, N, C, fb, vb, R);
      hipEventRecord(ev[3], 0);
      const int rps = (N + c.S - 1) / c.S;
      hipLaunchKernelGGL(k_dist3, dim3(F * c.S), dim3(256), size_t(C) * 4, 0, reinterpret_cast<const uint4*>(xh), N, C, c.S, rps, vc, fc, dout);
      hipEventRecord(ev[4], 0);
      hipEventSynchronize(ev[4]);
      if (rep >= 2) for (int k = 0; k < 4; ++k) { float ms; hipEventElapsedTime(&ms, ev[k], ev[k + 1]); t[k] += ms; }
    }
    hipError_t e = hipGetLastError();
    printf("db=%d rows/wg=%d halves=%d S3=%d | s1 %.1f us | xhat %.1f us (%.2f TB/s of 270 MB) | centres %.1f us | dist3 %.1f us | %s\n", c.db, c.rpw,
           c.halves, c.S, t[0] / reps * 1e3, t[1] / reps * 1e3, 270e6 / (t[1] / reps * 1e-3) / 1e12 * 1.0 * (double(R) * (D * 2 + C * 2) / 270e6),
           t[2] / reps * 1e3, t[3] / reps * 1e3, hipGetErrorString(e));
  }
  return 0;
}
