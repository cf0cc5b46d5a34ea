// vc2_kernels.hip -- MI355X (gfx950) kernels + C ABI of the VidCom2 token-compression hot path.
//
// Replaces the tensor work of token_compressor/vidcom2/vidcom2.py:15-115 (reference).  The pass
// is three dependent streaming sweeps over X[F*N, D] plus O(F*N) scalar work:
//
//   sweep 1  k_chan_stats      per-channel shifted sum / sum of squares (fp64) vidcom2.py:40
//            k_var_from_stats  canonical stat blocks, Chan-folded in a fixed shape -> var (T)
//            k_chan_select     lowest-variance half, CPU-reference ties       vidcom2.py:41-42
//            (k_chan_order     torch.topk's sorted ORDER: normally rider workgroups of sweep 2)
//   sweep 2  k_norm_colsum     token L2 norms, x^ = x/||x||, per-frame sums   vidcom2.py:47-52
//            k_frame_centres, k_video_centre   frame / video centres (T)      vidcom2.py:51-52
//   sweep 3  k_dist            squared distances to both centres + the 5-scale Gaussian sums, v + f and the
//                              per-frame score partials in its epilogue       vidcom2.py:61-62,32-33
//            k_select          softmax budgets, ks, per-frame bottom-k (libstdc++ ties) + map
//                                                                              vidcom2.py:64-77,99-115
//            k_gather_rows     kept rows (+ tail rows; several tensors)       vidcom2.py:91,96
//   "torch order" mode: values whose exact result lies next to a T rounding boundary are replayed in torch's fp32
//   accumulation order -- inside the sweeps' own workgroups (distances, centre means) or by k_norm_fix (norms)
//   (DESIGN.md "Numerics contract").
//   standalone helpers: k_multi_scale_gaussian (vidcom2.py:59-62), k_scales (:64-68), k_keep_positions (hooks);
//   POOL variant of k_chan_stats: LLaVA's get_2dPool fused with sweep 1.
//
// HBM-bound: no dense contraction exists in the reference, so no MFMA.  All global
// loads are 16 B/lane coalesced along D; reductions are fixed-order (no float atomics) so results
// are run-to-run deterministic.  Built with -ffp-contract=off (see vc2_device.h).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>
#include <vector>

#include "vc2_device.h"

#ifndef VC2_PROBE_D3
#define VC2_PROBE_D3 0
#endif
#ifndef VC2_PROBE_S2
#define VC2_PROBE_S2 0
#endif

using namespace vc2;

#ifdef VC2_DEBUG_TIMING
namespace vc2 {
__device__ unsigned long long g_dbg_t[512];
__device__ int g_dbg_v[512];
__device__ int g_dbg_n;
__device__ __forceinline__ void dbg_stamp(int tag) {      // (one lane) wall clock, 100 MHz
  const int i = atomicAdd(&g_dbg_n, 1);
  if (i < 512) { g_dbg_t[i] = wall_clock64(); g_dbg_v[i] = tag; }
}
}
#define VC2_STAMP(tag) dbg_stamp(tag)
// per-workgroup begin / end times of the three sweeps (slot 0: k_chan_stats, 1: k_norm_colsum, 2: k_dist)
namespace vc2 { __device__ unsigned long long g_dbg_wg[8][2][4096]; }
namespace vc2 { __device__ unsigned long long g_dbg_vc[6][4096]; }        // k_video_centre (scripts/dev/vc_waves.py): see there     // (slots 4, 5: sweep 2's first row landed / row loops over, combine done)
#define VC2_WGTIME(slot, which) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_dbg_wg[slot][which][blockIdx.x] = wall_clock64(); } while (0)
#else
#define VC2_STAMP(tag) ((void)0)
#define VC2_WGTIME(slot, which) ((void)0)
#endif

#include "vc2_select2.h"        // (after the debug macros: the selection rounds carry stamps in debug builds)

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ======================================================================================
// sweep 1: per-channel statistics
// ======================================================================================
// grid = (column slabs of 64 lanes x VEC, row groups); block = 4 waves sharing one slab, wave w
// takes rows r0+w, r0+w+4, ...  Each lane accumulates sum(x-K) and sum((x-K)^2) in fp64 for its
// VEC columns, K = x[0][c] (a shift: constant columns give exactly 0 and the final subtraction
// is well conditioned).  The 4 waves combine through LDS; one partial per (row group, column).
constexpr int kStatsWaves = 4;
#ifndef VC2_S1_PROBE
#define VC2_S1_PROBE 0
#endif

// Row groups are CANONICAL: a group is one of `splits` pieces of one frame, the shift K of a group
// is the first row of its STAT BLOCK (kStatBlockFrames frames), and the groups of a block are added in a fixed order
// (k_stats_reduce) -- so the per-block (mean, M2) and everything reduced from them are the same bits whether the
// video is processed whole or frame-sharded over 1, 2, 4 or 8 ranks.
constexpr int kStatBlockFrames = 8;

// POOL != 0: the rows are not read but MADE here -- row (f, oy, ox) of the pooled video is the
// 2x2 pool of four rows of the un-pooled projector output xin[F][H*W][D] (LLaVA's get_2dPool, reference
// llava/model/llava_arch.py:171-190, in the token-major layout it permutes from and back to) -- and stored to x on the
// way, so the pooled tensor is written once and its channel statistics cost no sweep of their own.
//   1 average: ((a + b) + c) + d in fp32, / 4, one rounding to T      (torch avg_pool2d's accumulation order)
//   2 max
//   3 bilinear to ceil(H/2) x ceil(W/2), align_corners=False: index / weights as ATen's
//     compute_source_index_and_lambda (fma'd source index); the four products w_yx * v are added in the order ATen's
//     vectorised channel loop (cpu_upsample_linear_channels_last, the kernel NCHW inputs with C >= 8 end up in) is
//     contracted to by the x86 build of torch 2.10: fp32  fma(w00,a, fma(w01,b, fma(w11,d, w10*c))),
//     bf16 / fp16  fma(w00,a, fma(w01,b, fma(w10,c, w11*d))); one rounding to T.  (Found by exhaustive search over
//     the association orders against torch's own outputs; the loop's scalar tail -- D % 8 channels in fp32, D % 16 in
//     16-bit -- contracts differently, so such D are refused.)
struct PoolSrc { const void* xin; int H, W, h, w, mode; };

template <int DT, int VEC>
__device__ __forceinline__ void pooled_row(const PoolSrc& ps, int64_t rr, int N, int D, int cv, float (&v)[VEC]) {
  const int64_t f = rr / N;
  const int t = int(rr - f * N), oy = t / ps.w, ox = t - oy * ps.w;
  int y0, y1, x0, x1;
  float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
  if (ps.mode == 3) {
    auto src = [](int in, int out, int i, int& i0, int& i1, float& l0, float& l1) {
      const float scale = float(in) / float(out);
      float real = __builtin_fmaf(scale, float(i) + 0.5f, -0.5f);
      real = real < 0.f ? 0.f : real;
      i0 = min(int(floorf(real)), in - 1);
      l1 = fminf(fmaxf(real - float(i0), 0.f), 1.f);
      l0 = 1.f - l1;
      i1 = i0 + (i0 < in - 1 ? 1 : 0);
    };
    float ly0, ly1, lx0, lx1;
    src(ps.H, ps.h, oy, y0, y1, ly0, ly1);
    src(ps.W, ps.w, ox, x0, x1, lx0, lx1);
    w00 = ly0 * lx0; w01 = ly0 * lx1; w10 = ly1 * lx0; w11 = ly1 * lx1;
  } else {
    y0 = 2 * oy; y1 = y0 + 1; x0 = 2 * ox; x1 = x0 + 1;
  }
  const int64_t fb = f * ps.H * ps.W;
  float a[VEC], b[VEC], c[VEC], d[VEC];
  const RawVec<DT, VEC> ra = load_raw<DT, VEC>(ps.xin, (fb + int64_t(y0) * ps.W + x0) * D + int64_t(cv) * VEC);
  const RawVec<DT, VEC> rb = load_raw<DT, VEC>(ps.xin, (fb + int64_t(y0) * ps.W + x1) * D + int64_t(cv) * VEC);
  const RawVec<DT, VEC> rc = load_raw<DT, VEC>(ps.xin, (fb + int64_t(y1) * ps.W + x0) * D + int64_t(cv) * VEC);
  const RawVec<DT, VEC> rd = load_raw<DT, VEC>(ps.xin, (fb + int64_t(y1) * ps.W + x1) * D + int64_t(cv) * VEC);
  unpack<DT, VEC>(ra, a); unpack<DT, VEC>(rb, b); unpack<DT, VEC>(rc, c); unpack<DT, VEC>(rd, d);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float o;
    if (ps.mode == 1) o = (((a[j] + b[j]) + c[j]) + d[j]) / 4.f;
    else if (ps.mode == 2) {
      // torch max_pool2d: (val > max) || isnan(val) in scan order
      o = a[j];
      if (b[j] > o || b[j] != b[j]) o = b[j];
      if (c[j] > o || c[j] != c[j]) o = c[j];
      if (d[j] > o || d[j] != d[j]) o = d[j];
    } else if constexpr (DT == VC2_F32) {      // ATen's Vectorized<float> loop (8 channels per step) as GCC contracts it
      o = w10 * c[j];
      o = __builtin_fmaf(w11, d[j], o); o = __builtin_fmaf(w01, b[j], o); o = __builtin_fmaf(w00, a[j], o);
    } else {                                   // ... and its reduced-precision instantiation (16 channels per step)
      o = w11 * d[j];
      o = __builtin_fmaf(w10, c[j], o); o = __builtin_fmaf(w01, b[j], o); o = __builtin_fmaf(w00, a[j], o);
    }
    v[j] = rnT<DT>(o);
  }
}
template <int DT, int VEC>
__device__ __forceinline__ void store_row_T(void* x, int64_t elem, const float (&v)[VEC]) {   // v: T-representable
  if constexpr (VEC == 1) {
    stT<DT>(x, elem, v[0]);
  } else if constexpr (DT == VC2_F32) {
    *reinterpret_cast<float4*>(static_cast<float*>(x) + elem) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (DT == VC2_BF16) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (__float_as_uint(v[2 * i]) >> 16) | (__float_as_uint(v[2 * i + 1]) & 0xFFFF0000u);
    *reinterpret_cast<uint4*>(static_cast<uint16_t*>(x) + elem) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    union { uint4 u; _Float16 h[8]; } c;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.h[i] = static_cast<_Float16>(v[i]);
    *reinterpret_cast<uint4*>(static_cast<uint16_t*>(x) + elem) = c.u;
  }
}

template <int DT, int VEC, int U, int POOL>
__global__ __launch_bounds__(kStatsWaves * 64) void k_chan_stats(const void* __restrict__ x, int64_t R,
                                                                 int D, int CV, int N, int splits,
                                                                 int rows_per_group, int block_frames,
                                                                 double* __restrict__ part, PoolSrc pool,
                                                                 uint32_t* __restrict__ var_reset = nullptr) {
  __shared__ double sm[kStatsWaves][2 * VEC][64];
  // (the pass's variance array back to "not written yet" for k_var_select, which polls it inside its own launch)
  if (var_reset && blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < D; i += kStatsWaves * 64) var_reset[i] = 0xFFFFFFFFu;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cv = blockIdx.x * 64 + lane;
  const bool active = cv < CV;
  const int64_t frame = int64_t(blockIdx.y) / splits;
  const int64_t r0 = frame * N + int64_t(blockIdx.y % splits) * rows_per_group;
  const int64_t r1 = min(min(R, (frame + 1) * N), r0 + rows_per_group);
  const int64_t rK = (frame / block_frames) * block_frames * N;               // the block's first row: the shift
  double s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = 0.0; q[j] = 0.0; }
  if (active) {
    float kf[VEC];
    if constexpr (POOL != 0) pooled_row<DT, VEC>(pool, rK, N, D, cv, kf);
    else unpack<DT, VEC>(load_raw<DT, VEC>(x, rK * D + int64_t(cv) * VEC), kf);
    double K[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) K[j] = double(kf[j]);
    if constexpr (POOL != 0) {
      for (int64_t rr = r0 + wave; rr < r1; rr += kStatsWaves) {
        float v[VEC];
        pooled_row<DT, VEC>(pool, rr, N, D, cv, v);
        store_row_T<DT, VEC>(const_cast<void*>(x), rr * D + int64_t(cv) * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const double d = double(v[j]) - K[j];
          s[j] += d;
          q[j] = fma(d, d, q[j]);
        }
      }
    } else
    for (int64_t r = r0 + wave; r < r1; r += int64_t(kStatsWaves) * U) {
      RawVec<DT, VEC> raw[U];
#if VC2_S1_PROBE == 2      // (timing probe, results invalid: the arithmetic alone -- one round of loads, reused)
      if (r == r0 + wave) {
#endif
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + int64_t(u) * kStatsWaves;
        raw[u] = rr < r1 ? load_raw<DT, VEC>(x, rr * D + int64_t(cv) * VEC) : zero_raw<DT, VEC>();
      }
#if VC2_S1_PROBE == 2
      }
      if constexpr (VEC == 8) {
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(raw[u].v.x), "+v"(raw[u].v.y), "+v"(raw[u].v.z), "+v"(raw[u].v.w));
      }
#endif
#if VC2_S1_PROBE == 1      // (timing probe, results invalid: the loads alone)
      if constexpr (VEC == 8) {
        unsigned acc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= raw[u].v.x ^ raw[u].v.y ^ raw[u].v.z ^ raw[u].v.w;
        s[0] += double(acc);
      }
      continue;
#endif
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = r + int64_t(u) * kStatsWaves;
        if (rr < r1) {
          float v[VEC];
          unpack<DT, VEC>(raw[u], v);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const double d = double(v[j]) - K[j];
            s[j] += d;
            q[j] = fma(d, d, q[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sm[wave][j][lane] = s[j]; sm[wave][VEC + j][lane] = q[j]; }
  __syncthreads();
  // all four waves add the four wave sums (wave order: fixed) and store with consecutive threads on consecutive
  // columns of the slab: the slab's 64 * VEC columns, sums first, then squares
  const int64_t c0 = int64_t(blockIdx.x) * 64 * VEC;
  for (int t = threadIdx.x; t < 2 * 64 * VEC; t += kStatsWaves * 64) {
    const int which = t / (64 * VEC), lc = t % (64 * VEC);
    const int l = lc / VEC, j = lc % VEC;
    if (c0 + lc < D) {
      double a = 0.0;
#pragma unroll
      for (int w = 0; w < kStatsWaves; ++w) a += sm[w][which * VEC + j][l];
      part[(int64_t(blockIdx.y) * 2 + which) * D + c0 + lc] = a;
    }
  }
}

// Level 1: per stat block b (kStatBlockFrames frames; the last one may be shorter) the fixed-order sum of its
// groups' partials -> (mean_b, M2_b) in fp64.  bstats[b][2][D].  grid = (ceil(D/64), nb), 64 threads.
struct PartSrc {           // where the sweep-1 partials of a rank live (null part: none)
  const double* part; const void* x; int64_t R; int groups_per_block; int G; int N; int block_frames;
};
template <int DT>
__device__ __forceinline__ void block_stat(const PartSrc& ps, int b, int c, int D, double& mean, double& m2) {
  const int g0 = b * ps.groups_per_block, g1 = min(ps.G, g0 + ps.groups_per_block);
  double s = 0.0, q = 0.0;
  for (int gb = g0; gb < g1; gb += 8) {                  // eight groups' loads in flight, added in group order
    double vs[8], vq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = min(gb + u, g1 - 1);
      vs[u] = ps.part[(int64_t(g) * 2 + 0) * D + c];
      vq[u] = ps.part[(int64_t(g) * 2 + 1) * D + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) if (gb + u < g1) { s += vs[u]; q += vq[u]; }
  }
  const int64_t rK = int64_t(b) * ps.block_frames * ps.N;
  const double n = double(min<int64_t>(ps.R - rK, int64_t(ps.block_frames) * ps.N));
  const double K = double(ldT<DT>(ps.x, rK * D + c));
  m2 = q - s * s / n;
  if (m2 < 0.0) m2 = 0.0;
  mean = K + s / n;
}
template <int DT>
__global__ __launch_bounds__(64) void k_block_stats(PartSrc ps, int D, double* __restrict__ bstats) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= D) return;
  double mean, m2;
  block_stat<DT>(ps, b, c, D, mean, m2);
  bstats[(int64_t(b) * 2 + 0) * D + c] = mean;
  bstats[(int64_t(b) * 2 + 1) * D + c] = m2;
}

// Level 2: Chan et al. combination of the NB block statistics (every block n_each rows, the last one n_last), in a
// FIXED shape -- 16 lanes, lane l folds blocks l, l + 16, ... in order, then the 16 lane aggregates are folded in
// lane order -- so that the result depends on the blocks only, not on who computed them.
// var = M2_total / R_total rounded fp64 -> fp32 -> T (vidcom2.py:40).
constexpr int kRedGL = 16;
struct ChanAgg { double n, mean, m2; };
__device__ __forceinline__ void chan_fold(ChanAgg& a, double nb, double mb, double m2b) {
  if (nb <= 0.0) return;
  if (a.n <= 0.0) { a.n = nb; a.mean = mb; a.m2 = m2b; return; }
  const double n = a.n + nb, d = mb - a.mean;
  a.mean = a.mean + d * (nb / n);
  a.m2 = a.m2 + m2b + d * d * (a.n * nb / n);
  a.n = n;
}
// The second-level fold (16 lane aggregates in lane order) only depends on the block sizes, not on the column: its
// two quotients per step -- nb / n and (a.n * nb) / n, each a ~40-instruction fp64 division, 30 of them in a serial
// chain per column (2.5 us of the kernel's 6.9) -- are computed once on the host.  IEEE division on both sides: the
// same bits as chan_fold's.
struct FoldTab { double r1[kRedGL], r2[kRedGL], ntot; int kind[kRedGL]; };     // kind: 0 skip, 1 assign, 2 fold
inline FoldTab make_fold_tab(int NB, int64_t n_each, int64_t n_last) {
  FoldTab t{};
  double an = 0.0;
  for (int i = 0; i < kRedGL; ++i) {
    double nb = 0.0;                                     // rows of lane i's blocks i, i + 16, ... (added like chan_fold does)
    for (int b = i; b < NB; b += kRedGL) nb += double(b == NB - 1 ? n_last : n_each);
    if (nb <= 0.0) { t.kind[i] = 0; continue; }
    if (an <= 0.0) { t.kind[i] = 1; an = nb; continue; }
    const double n = an + nb;
    t.kind[i] = 2; t.r1[i] = nb / n; t.r2[i] = an * nb / n;
    an = n;
  }
  t.ntot = an;
  return t;
}
// CW columns per workgroup, kRedGL lanes per column.  Narrow slabs (CW = 16: 224 workgroups at D = 3584) spread the
// 7 MB of sweep-1 partials over the whole chip instead of 56 CUs; the fold shape does not depend on CW.
// (the body: `vb` = the virtual workgroup -- CW columns --, `lt` = the thread inside it (CW * kRedGL of them), `sm` = its
//  exchange area; VIS: the variances are stored with agent-scope atomic stores, for a consumer that POLLS them inside the
//  same launch -- k_var_select)
template <int DT, int CW, bool VIS>
__device__ __forceinline__ void var_from_stats_body(const double* __restrict__ bstats, int NB, int64_t n_each, int64_t n_last, int D,
                                                    void* __restrict__ var_T, float* __restrict__ var_f32, const PartSrc& ps,
                                                    const FoldTab& ft, int vb, int lt, double (*sm)[kRedGL][CW]) {
  const int cl = lt % CW, gl = lt / CW;
  const int c = vb * CW + cl;
  ChanAgg a{0.0, 0.0, 0.0};
  if (c < D)
    for (int b = gl; b < NB; b += kRedGL) {
      double mb, m2b;
      if (ps.part) block_stat<DT>(ps, b, c, D, mb, m2b);
      else { mb = bstats[(int64_t(b) * 2 + 0) * D + c]; m2b = bstats[(int64_t(b) * 2 + 1) * D + c]; }
      chan_fold(a, double(b == NB - 1 ? n_last : n_each), mb, m2b);
    }
  sm[0][gl][cl] = a.n; sm[1][gl][cl] = a.mean; sm[2][gl][cl] = a.m2;
  __syncthreads();
  if (gl != 0 || c >= D) return;
  ChanAgg t{0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < kRedGL; ++i) {                     // chan_fold with the host's quotients (FoldTab)
    const double mb = sm[1][i][cl], m2b = sm[2][i][cl];
    if (ft.kind[i] == 1) { t.mean = mb; t.m2 = m2b; }
    else if (ft.kind[i] == 2) {
      const double d = mb - t.mean;
      t.mean = t.mean + d * ft.r1[i];
      t.m2 = t.m2 + m2b + d * d * ft.r2[i];
    }
  }
  t.n = ft.ntot;
  const float v = rnT<DT>(float(t.m2 / t.n));
  if (var_f32) {
    if constexpr (VIS) __hip_atomic_store(var_f32 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else var_f32[c] = v;
  }
  if (var_T) stT<DT>(var_T, c, v);
}
// CW columns per workgroup, kRedGL lanes per column.  Narrow slabs (CW = 16: 224 workgroups at D = 3584) spread the
// 7 MB of sweep-1 partials over the whole chip instead of 56 CUs; the fold shape does not depend on CW.
template <int DT, int CW = 64>
__global__ __launch_bounds__(CW * kRedGL) void k_var_from_stats(const double* __restrict__ bstats, int NB,
                                                                int64_t n_each, int64_t n_last, int D,
                                                                void* __restrict__ var_T, float* __restrict__ var_f32,
                                                                int* __restrict__ counters, PartSrc ps, FoldTab ft,
                                                                unsigned long long* __restrict__ fixq = nullptr,
                                                                int nfixq = 0, unsigned long long* __restrict__ kstatus = nullptr) {
  // ps.part != null: the block statistics are computed here from this rank's sweep-1 partials (the same arithmetic
  // as k_block_stats: one launch less on the single-rank path); else they are read from bstats
  __shared__ double sm[3][kRedGL][CW];
  if (blockIdx.x == 0 && threadIdx.x == 0) VC2_STAMP(100);
  if (counters && blockIdx.x == 0 && threadIdx.x < 16) counters[threadIdx.x] = 0;  // strict-mode queues of this pass
  if (kstatus && blockIdx.x == 0 && threadIdx.x == 0) *kstatus = 0ull;             // K_out[1]: k_select ORs its bits in
  if (fixq) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nfixq; i += gridDim.x * blockDim.x) fixq[i] = 0ull;
  var_from_stats_body<DT, CW, false>(bstats, NB, n_each, n_last, D, var_T, var_f32, ps, ft, int(blockIdx.x), int(threadIdx.x), sm);
  if (blockIdx.x == 0 && threadIdx.x == 0) VC2_STAMP(109);
}

// ======================================================================================
// channel selection (single workgroup of 4 waves; D <= 8192)
// ======================================================================================
// vidcom2.py:41-42.  k_chan_select replays torch.topk's SELECTION (libstdc++ introselect / heap-select,
// vc2_select2.h) and writes the ascending list of kept channels the scoring sweeps consume, plus -- perm -- the kept
// channels in the order nth_element left them.  torch.topk(sorted=True)'s ORDER of those channels (std::sort on that
// permutation; only the "torch order" replays and the select_low_var_channels API need it) is a second, separate
// piece of work, chan_order_body, done by several workgroups side by side (slices of the sorted order, see
// introsort2): either its own kernel, or RIDER workgroups of sweep 2 (k_norm_colsum) -- its consumers run after
// the sweep.
#ifndef VC2_SEL_NT
#define VC2_SEL_NT 1024
#endif
constexpr int kSelNT = VC2_SEL_NT;  // k_chan_select
#ifndef VC2_SEL_ACTIVE
#define VC2_SEL_ACTIVE 16
#endif
// waves that take part in a cooperative round (the rest idle at the barriers).  Measured at D = 3584: 16 -> 20.8 us,
// 8 -> 22.2, 4 -> 27.4: the elements per thread cost more than the extra waves per SIMD
constexpr int kSelActive = VC2_SEL_ACTIVE;
constexpr int kSelCoop = 8192 / (64 * kSelActive * 4);                      // quads per thread at D = 8192
constexpr int kSelSolo = kSelActive >= 16 ? 2 : 4;
constexpr int kOrdNT = 256;         // k_chan_order: 4 waves per slice, like the rider workgroups of sweep 2

// block-wide exclusive prefix of a small count (thread-contiguous chunks), fixed order; NW waves
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t c, uint32_t* xch, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(c);
  if (lane == 63) xch[wave] = incl;
  __syncthreads();
  uint32_t pre = 0u, all = 0u;
#pragma unroll
  for (int v = 0; v < NW; ++v) { const uint32_t t = xch[v]; pre += v < wave ? t : 0u; all += t; }
  total = all;
  return pre + incl - c;
}

constexpr int kSelPre = (8192 + kSelNT - 1) / kSelNT;     // variances a thread holds (D <= 8192)
// what follows the selection replay: S.w[0, k) holds the kept channels in the order nth_element left them
template <typename W, int NT>
__device__ __forceinline__ void chan_select_epilogue(const Sel2<W>& S, int D, int k, uint8_t* __restrict__ mask,
                                                     int* __restrict__ cols, int* __restrict__ perm,
                                                     uint32_t* __restrict__ wperm, uint32_t* __restrict__ wcpos) {
  using T = WordTr<W>;
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x;
  if (tid == 0) VC2_STAMP(205);
  if (tid == 0) VC2_ROUND(S, 291, k);
  if (perm) for (int i = tid; i < k; i += NT) perm[i] = T::idx(S.w[i]);
  // kept flags (la is free now), mask bytes, and the ascending list of kept channels (ordered compaction).  S.w holds
  // every channel exactly once, the kept ones in [0, k): ONE pass writes every flag (no zeroing pass, no barrier between)
  for (int i = tid; i < D; i += NT) S.la[T::idx(S.w[i])] = (i < k || k >= D) ? 1 : 0;
  __syncthreads();
  const int Ept = (D + NT - 1) / NT;
  const int b = tid * Ept, e = min(D, b + Ept);
  uint32_t cnt = 0;
  for (int p = b; p < e; ++p) cnt += S.la[p];
  uint32_t tot;
  int o = int(block_excl_scan<NW>(cnt, S.xch, tot));
  if (tid == 0) VC2_ROUND(S, 292, k);                             // flags + scan done
  for (int p = b; p < e; ++p) {
    const bool on = S.la[p] != 0;
    if (mask) mask[p] = on ? 1 : 0;
    if (on) { if (cols) cols[o] = p; S.lb[p] = uint16_t(o); ++o; }
  }
  // for the ORDER riders: the kept channels as packed words in nth_element's order, and each one's position in cols
  // (one coalesced load each instead of perm -> var_f32 and cols -> table)
  if constexpr (sizeof(W) == 4) {
    if (wperm && wcpos && k < D) {
      __syncthreads();
      for (int i = tid; i < k; i += NT) { const W w = S.w[i]; wperm[i] = uint32_t(w); wcpos[i] = S.lb[T::idx(w)]; }
    }
  }
  if (tid == 0) VC2_ROUND(S, 299, k);
}

// The full pass's epilogue (round 6; 1024 threads, D <= 4096, 32-bit words, no mask bytes wanted): ONE barrier instead of three.
// The kept channels set their bits in a 128-word LDS bitmap (and the ORDER riders' words go out) -- barrier -- every wave
// forms the prefix popcounts of the 128 words for itself (two DPP scans: no exchange), then thread t lists the kept ones of
// channels 4t .. 4t + 3 in `cols`, and the position of a kept channel c in `cols` (wcpos) is prefix[c / 32] + the set bits
// below c in its word -- no flag array, no per-thread chunk counts, no block scan.  bitmap: 128 zeroed words (S.xch is not
// used by the replay's last phase; the caller zeroes them before the replay).
__device__ __forceinline__ void chan_select_epilogue_bitmap(const Sel2<uint32_t>& S, uint32_t* bitmap, int D, int k,
                                                            int* __restrict__ cols, int* __restrict__ perm,
                                                            uint32_t* __restrict__ wperm, uint32_t* __restrict__ wcpos) {
  using T = WordTr<uint32_t>;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) VC2_STAMP(205);
  if (tid == 0) VC2_ROUND(S, 291, k);
  uint32_t wk[4];                                                  // my kept words: positions tid, tid + 1024, ... (k <= 4096)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + j * kSelNT;
    wk[j] = 0u;
    if (i < k) {
      wk[j] = S.w[i];
      const int c = T::idx(wk[j]);
      atomicOr(&bitmap[c >> 5], 1u << (c & 31));
      if (wperm) wperm[i] = wk[j];
      if (perm) perm[i] = c;
    }
  }
  __syncthreads();
  if (tid == 0) VC2_ROUND(S, 292, k);
  const uint32_t w0 = bitmap[lane], w1 = bitmap[64 + lane];
  const uint32_t c0 = uint32_t(__builtin_popcount(w0)), c1 = uint32_t(__builtin_popcount(w1));
  const uint32_t i0 = wave_incl_scan_u32(c0);
  const uint32_t t0 = uint32_t(__builtin_amdgcn_readlane(int(i0), 63));
  const uint32_t i1 = wave_incl_scan_u32(c1) + t0;
  const uint32_t e0 = i0 - c0, e1 = i1 - c1;                        // lane l: kept channels below word l / below word 64 + l
  auto prefix_of = [&](int word) -> uint32_t {                      // (any lane asks for any word: two crossbar reads)
    const uint32_t a = uint32_t(__builtin_amdgcn_ds_bpermute((word & 63) << 2, int(e0)));
    const uint32_t b = uint32_t(__builtin_amdgcn_ds_bpermute((word & 63) << 2, int(e1)));
    return word < 64 ? a : b;
  };
  {                                                                 // cols: thread t answers for channels 4t .. 4t + 3
    const int cb = 4 * tid, word = cb >> 5;
    const uint32_t wv = bitmap[word < 128 ? word : 127];
    uint32_t o = prefix_of(word) + uint32_t(__builtin_popcount(wv & ((1u << (cb & 31)) - 1u)));
    if (cols && cb < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if ((wv >> ((cb & 31) + e)) & 1u) cols[o++] = cb + e;
    }
  }
  if (wcpos) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + j * kSelNT;
      const int c = T::idx(wk[j]);
      const uint32_t wv = bitmap[c >> 5];
      const uint32_t pos = prefix_of(c >> 5) + uint32_t(__builtin_popcount(wv & ((1u << (c & 31)) - 1u)));
      if (i < k) wcpos[i] = pos;
    }
  }
  if (tid == 0) VC2_ROUND(S, 299, k);
}

// the LDS-round engine (vc2_select2.h, second generation) on NT threads: every D <= 8192, both word widths
template <typename W, int NT, int NPRE>
__device__ __forceinline__ void chan_select_body(unsigned char* smem, const float (&pre)[NPRE], int D, int k,
                                                 uint8_t* __restrict__ mask, int* __restrict__ cols,
                                                 int* __restrict__ perm, uint32_t* __restrict__ wperm,
                                                 uint32_t* __restrict__ wcpos, int* status) {
  using T = WordTr<W>;
  constexpr int NW = NT / 64;
  constexpr int SOLO = NT == kSelNT ? kSelSolo : 4, COOP = NT == kSelNT ? kSelCoop : 8, ACT = NT == kSelNT ? kSelActive : NW;
  const int tid = threadIdx.x;
  Sel2<W> S = sel2_carve<W>(smem, D, status);
#if defined(VC2_DEBUG_TIMING)
  S.dbg_slot = 0;
#endif
  if (tid == 0) VC2_ROUND(S, 201, D);                             // the variances are in registers
#pragma unroll
  for (int j = 0; j < NPRE; ++j) { const int i = tid + j * NT; if (i < D) S.w[i] = T::pack(topk_key(pre[j]), i); }
  __syncthreads();
  if (tid == 0) VC2_ROUND(S, 202, D);                             // packed words in LDS
  if (k >= D) { if (perm) introselect2<W, NW, SOLO, COOP, ACT>(S, D, D - 1, tid); }    // nth_element(n-1) still permutes
  else topk_smallest2<W, NW, SOLO, COOP, ACT>(S, D, k, tid);
  // (no barrier here: both replays end with one, and the cases that replay nothing have the barrier above behind them)
  chan_select_epilogue<W, NT>(S, D, k, mask, cols, perm, wperm, wcpos);
}

#ifndef VC2_DEV_ONLY
__global__ __launch_bounds__(kSelNT) void k_chan_select(const float* __restrict__ var_f32, int D, int k,
                                                        uint8_t* __restrict__ mask, int* __restrict__ cols,
                                                        int* __restrict__ perm, uint32_t* __restrict__ wperm,
                                                        uint32_t* __restrict__ wcpos, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int bad = 0;
  if (threadIdx.x == 0) VC2_STAMP(200);
  if (threadIdx.x == 0) VC2_ROUND_RAW(0, 120, 200);
  float pre[kSelPre];                                  // (one round of loads; the bodies pack from registers)
#pragma unroll
  for (int j = 0; j < kSelPre; ++j) { const int i = threadIdx.x + j * kSelNT; pre[j] = var_f32[i < D ? i : D - 1]; }
#pragma unroll
  for (int j = 0; j < kSelPre; ++j) bad |= key_fits_u32(pre[j]) ? 0 : 1;
  // widened 16-bit variances pack into 32-bit words (the common case); arbitrary fp32 ones take 64-bit words
  // (wperm / wcpos are only requested for 16-bit inputs)
  if (__syncthreads_or(bad)) chan_select_body<uint64_t, kSelNT, kSelPre>(smem, pre, D, k, mask, cols, perm, nullptr, nullptr, status);
  else chan_select_body<uint32_t, kSelNT, kSelPre>(smem, pre, D, k, mask, cols, perm, wperm, wcpos, status);
  if (threadIdx.x == 0) VC2_STAMP(209);
}
#endif

// Round 6, the form that is the default for D <= 4096: sixteen waves as in k_chan_select, but thread t OWNS channels 4t .. 4t + 3
// in registers for every partition round above 64 elements (vc2_select2.h, sel4_rounds); variances that need 64-bit words
// (fp32 inputs) take the LDS-round engine on the same sixteen waves.
constexpr int kStatusSpinExpired = 1;    // a bounded wait inside a launch ran out (reported through K_out[1])
constexpr uint32_t kVarSentinel = 0xFFFFFFFFu;       // "not written yet" (k_var_select): no variance is this NaN -- arithmetic NaNs are
                                                     // canonical, a widened 16-bit value has sixteen zero low bits
// POLL: the variances are being written by OTHER workgroups of this launch (k_var_select): every thread spins on its own four
// words until none is the sentinel (single-word data is its own flag: no ordering between words is needed).  A bounded wait:
// a word that never arrives is reported (status bit kStatusSpinExpired) and read as it is.
template <bool POLL>
__device__ __forceinline__ void chan_select4_body(unsigned char* smem, const float* __restrict__ var_f32, int D, int k,
                                                  uint8_t* __restrict__ mask, int* __restrict__ cols,
                                                  int* __restrict__ perm, uint32_t* __restrict__ wperm,
                                                  uint32_t* __restrict__ wcpos, int* status) {
  using T = WordTr<uint32_t>;
  static_assert(kSelNT == 1024 && kSelPre >= 4, "thread t: channels 4t - off .. 4t - off + 3");
  const int tid = threadIdx.x;
  int bad = 0;
  if (tid == 0) VC2_STAMP(200);
  if (tid == 0) VC2_ROUND_RAW(0, 120, 200);
  constexpr int NW = kSelNT / 64;
  const int nth = (k >= D || k <= 0) ? D - 1 : k - 1;
  const int off = sel4_offset(D, nth, NW);                         // thread t: channels 4t - off .. 4t - off + 3 (see introselect4)
  float pv[4];
  if constexpr (POLL) {
    uint32_t raw[4];
    bool missing = true;
    for (int spin = 0; missing && spin < (1 << 16); ++spin) {
      missing = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * tid - off + e;
        raw[e] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(var_f32) + (i < 0 ? 0 : (i < D ? i : D - 1)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        missing |= raw[e] == kVarSentinel;
      }
      if (missing) __builtin_amdgcn_s_sleep(2);
    }
    if (missing && status) atomicOr(status, kStatusSpinExpired);
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] = __uint_as_float(raw[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = 4 * tid - off + e; pv[e] = var_f32[i < 0 ? 0 : (i < D ? i : D - 1)]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) bad |= key_fits_u32(pv[e]) ? 0 : 1;
  if (__syncthreads_or(bad)) {                           // 64-bit words: the strided mapping of the LDS-round body
    float pre[kSelPre];                                  // (POLL: every thread has seen its own words, the barrier has passed: all are there)
#pragma unroll
    for (int j = 0; j < kSelPre; ++j) { const int i = tid + j * kSelNT; pre[j] = var_f32[i < D ? i : D - 1]; }
    chan_select_body<uint64_t, kSelNT, kSelPre>(smem, pre, D, k, mask, cols, perm, nullptr, nullptr, status);
    if (tid == 0) VC2_STAMP(209);
    return;
  }
  Sel2<uint32_t> S = sel2_carve<uint32_t>(smem, D, status);
#if defined(VC2_DEBUG_TIMING)
  S.dbg_slot = 0;
#endif
  if (tid == 0) VC2_ROUND(S, 201, D);
  // the epilogue's bitmap of kept channels: 128 words behind the selection's own LDS (chan_select_lds reserves them), zeroed here
  uint32_t* const bitmap = reinterpret_cast<uint32_t*>(smem + (sel2_bytes(D, 4) + 15) / 16 * 16);
  if (tid < 128) bitmap[tid] = 0u;
  uint32_t el[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int i = 4 * tid - off + e; el[e] = T::pack(topk_key(pv[e]), (i >= 0 && i < D) ? i : 0); }
  if (uint32_t(4 * tid - off) < uint32_t(D)) *reinterpret_cast<uint4*>(S.w + (4 * tid - off)) = make_uint4(el[0], el[1], el[2], el[3]);   // (the pad takes the last thread's excess)
  __syncthreads();
  if (tid == 0) VC2_ROUND(S, 202, D);
  if (k >= D) { if (perm) introselect4<NW>(S, el, D, D - 1, tid, off); }                // nth_element(n-1) still permutes
  else if (k > 0 && int64_t(k) * 64 <= int64_t(D)) { if (tid == 0) s2_heap_select(S.w, 0, k, D); __syncthreads(); }   // partial_sort regime
  else if (k > 0) introselect4<NW>(S, el, D, k - 1, tid, off);
  if (!mask && k > 0 && k < D) chan_select_epilogue_bitmap(S, bitmap, D, k, cols, perm, wperm, wcpos);
  else chan_select_epilogue<uint32_t, kSelNT>(S, D, k, mask, cols, perm, wperm, wcpos);
  if (tid == 0) VC2_STAMP(209);
}
__global__ __launch_bounds__(kSelNT) void k_chan_select4(const float* __restrict__ var_f32, int D, int k,
                                                         uint8_t* __restrict__ mask, int* __restrict__ cols,
                                                         int* __restrict__ perm, uint32_t* __restrict__ wperm,
                                                         uint32_t* __restrict__ wcpos, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  chan_select4_body<false>(smem, var_f32, D, k, mask, cols, perm, wperm, wcpos, status);
}

// k_var_from_stats AND the channel selection in ONE launch (round 6; the full pass, 16-bit inputs, D <= 4096): workgroup 0 is
// the selection -- it sets up and then POLLS the variances --, workgroups 1 .. are the variance reduction, four virtual
// 256-thread workgroups of k_var_from_stats<DT, 16> each (same fold shape: same bits).  What it saves is a kernel boundary
// (~1.5 us between the last wave of one launch and the first instruction of the next) and the selection's own load round
// trip.  var_f32 must hold the sentinel when the launch begins: the sweep-1 launch of the same pass wrote it (k_chan_stats).
struct VarSelArgs {
  int NB; int64_t n_each, n_last; int D; float* var_f32; int* counters; PartSrc ps; FoldTab ft;
  unsigned long long* fixq; int nfixq; unsigned long long* kstatus;
  int k; int* cols; int* perm; uint32_t* wperm; uint32_t* wcpos; int* status;
  int warm;      // bytes of this kernel's own code the selection workgroup reads (as data) while it waits: see k_var_select
};
template <int DT>
__global__ __launch_bounds__(kSelNT) void k_var_select(VarSelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CW = 16, VT = CW * kRedGL;                          // the variance workgroups use their first 256 threads (the other
  __shared__ double sm[3][kRedGL][CW];                             //   waves leave at once): 224 CUs share the 7 MB of partials, as
  if (blockIdx.x == 0) {                                           //   k_var_from_stats<DT, 16> has it (64-column workgroups: +1 us)
    if (a.counters && threadIdx.x < 16) a.counters[threadIdx.x] = 0;     // strict-mode queues + status word of this pass (the word
    __syncthreads();                                                     //   this workgroup's own replay may OR a bit into: zeroed here)
    if (a.warm > 0) {
      // ONE workgroup runs ~40 KB of branchy code once per pass, after three sweeps have pushed it out of the L2: every jump
      // into a new piece (the first multi-wave round, the one-wave rounds, the tail, the epilogue) waits for an instruction
      // fetch from memory.  The variances take ~4 us to arrive: meanwhile the threads read the kernel's code as DATA, one
      // 64-byte line each, so that the fetches that follow hit the L2.  (a.warm < this function's code size: the range
      // [pc, pc + warm) lies inside it.)
      const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc());
      const int o0 = int(threadIdx.x) * 64, o1 = o0 + kSelNT * 64;  // (warm <= 128 KB: two lines per thread, both in flight)
      const uint32_t w0 = o0 < a.warm ? *reinterpret_cast<const uint32_t*>(pc + o0) : 0u;
      const uint32_t w1 = o1 < a.warm ? *reinterpret_cast<const uint32_t*>(pc + o1) : 0u;
      asm volatile("" :: "v"(w0), "v"(w1));                        // (keeps the loads; nobody waits for them before the poll does)
    }
    chan_select4_body<true>(smem, a.var_f32, a.D, a.k, nullptr, a.cols, a.perm, a.wperm, a.wcpos, a.status);
    return;
  }
  if (threadIdx.x >= VT) return;
  const int vb = int(blockIdx.x) - 1, nvt = (int(gridDim.x) - 1) * VT, vt = vb * VT + int(threadIdx.x);
  if (vb == 0 && threadIdx.x == 0) VC2_STAMP(100);
  if (a.kstatus && vt == 0) *a.kstatus = 0ull;                     // K_out[1]: k_select ORs its bits in
  if (a.fixq) for (int i = vt; i < a.nfixq; i += nvt) a.fixq[i] = 0ull;
  var_from_stats_body<DT, CW, true>(nullptr, a.NB, a.n_each, a.n_last, a.D, nullptr, a.var_f32, a.ps, a.ft, vb, int(threadIdx.x), sm);
  if (vb == 0 && threadIdx.x == 0) VC2_STAMP(109);
}

// Round 6: the channel selection on FOUR waves (one per SIMD) with the whole array in registers for every partition round
// (vc2_select2.h, third generation: sel3_rounds) -- D <= 256 E channels whose variances pack into 32-bit words (16-bit
// inputs: always).  Thread t holds channels t, t + 256, ...: position p of the array = slot p / 256 of thread p % 256, which
// is row (p / 64) of wave (p / 64) % 4 -- the layout sel3_rounds wants.  Variances that need 64-bit words take the LDS-round
// engine on the same four waves.
constexpr int kSel3NT = 256;
template <int E>
__global__ __launch_bounds__(kSel3NT) void k_chan_select3(const float* __restrict__ var_f32, int D, int k,
                                                          uint8_t* __restrict__ mask, int* __restrict__ cols,
                                                          int* __restrict__ perm, uint32_t* __restrict__ wperm,
                                                          uint32_t* __restrict__ wcpos, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using T = WordTr<uint32_t>;
  const int tid = threadIdx.x;
  int bad = 0;
  if (tid == 0) VC2_STAMP(200);
  if (tid == 0) VC2_ROUND_RAW(0, 120, 200);
  float pre[E];
#pragma unroll
  for (int j = 0; j < E; ++j) { const int i = tid + j * kSel3NT; pre[j] = var_f32[i < D ? i : D - 1]; }
#pragma unroll
  for (int j = 0; j < E; ++j) bad |= key_fits_u32(pre[j]) ? 0 : 1;
  if (__syncthreads_or(bad)) {
    chan_select_body<uint64_t, kSel3NT, E>(smem, pre, D, k, mask, cols, perm, nullptr, nullptr, status);
    if (tid == 0) VC2_STAMP(209);
    return;
  }
  Sel2<uint32_t> S = sel2_carve<uint32_t>(smem, D, status);
#if defined(VC2_DEBUG_TIMING)
  S.dbg_slot = 0;
#endif
  if (tid == 0) VC2_ROUND(S, 201, D);
  uint32_t el[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tid + j * kSel3NT;
    el[j] = T::pack(topk_key(pre[j]), i < D ? i : 0);
    if (i < D) S.w[i] = el[j];
  }
  __syncthreads();
  if (tid == 0) VC2_ROUND(S, 202, D);
  constexpr int NW = kSel3NT / 64;
  if (k >= D) { if (perm) introselect3<NW, E, 4>(S, el, D, D - 1, tid); }              // nth_element(n-1) still permutes
  else if (k > 0 && int64_t(k) * 64 <= int64_t(D)) { if (tid == 0) s2_heap_select(S.w, 0, k, D); __syncthreads(); }   // partial_sort regime
  else if (k > 0) introselect3<NW, E, 4>(S, el, D, k - 1, tid);
  chan_select_epilogue<uint32_t, kSel3NT>(S, D, k, mask, cols, perm, wperm, wcpos);
  if (tid == 0) VC2_STAMP(209);
}
__host__ inline size_t chan_select_lds(int D) {     // 64-bit words (the fallback), or 32-bit words + the epilogue's 128-word bitmap
  return std::max(sel2_bytes(D, 8) + 64, (sel2_bytes(D, 4) + 15) / 16 * 16 + 512 + 64);
}

// torch.topk(sorted=True)'s ORDER of the k kept channels from `perm` (what nth_element / partial_sort left in
// [0, k)): std::sort(q, q + k - 1) with the nth_element pivot staying last, or sort_heap in the partial_sort
// regime (TopKImpl.h).  order[p] = channel at sorted position p, opos[p] = its position in the ascending list
// `cols`, spos = the inverse of opos.  64*NW threads; any of the outputs may be null.
template <typename W, int NW, int SOLO, int COOP>
__device__ __forceinline__ void chan_order_body(unsigned char* smem, const float* __restrict__ var_f32, int D, int k,
                                                const int* __restrict__ perm, const int* __restrict__ cols,
                                                int* __restrict__ order, int* __restrict__ opos,
                                                int* __restrict__ spos, int part, int nparts,
                                                const uint32_t* __restrict__ wperm = nullptr,
                                                const uint32_t* __restrict__ wcpos = nullptr, int* status = nullptr) {
  using T = WordTr<W>;
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  Sel2<W> S = sel2_carve<W>(smem, k, status);
  unsigned char* p = smem + (sel2_bytes(k, int(sizeof(W))) + 15) / 16 * 16;
  SortScratch2 Q = sort2_carve(p, NW);
  uint16_t* cpos = reinterpret_cast<uint16_t*>(p + (sort2_bytes(k, NW) + 15) / 16 * 16);   // [D] channel -> position in cols
  bool packed = false;
  if constexpr (sizeof(W) == 4) packed = wperm != nullptr && wcpos != nullptr && k < D;
  if (packed) {                                      // k_chan_select left the words and their cols positions ready
    if constexpr (sizeof(W) == 4)
      for (int i = tid; i < k; i += NT) { const W w = wperm[i]; S.w[i] = w; cpos[T::idx(w)] = uint16_t(wcpos[i]); }
  } else {
    for (int i = tid; i < k; i += NT) { const int c = perm[i]; S.w[i] = T::pack(topk_key(var_f32[c]), c); }
    if (cols) for (int j = tid; j < k; j += NT) cpos[cols[j]] = uint16_t(j);
  }
  __syncthreads();
  auto emit = [&](int q, int c) {                    // channel c stands at sorted position q
    if (order) order[q] = c;
    if (cols && opos) { const int cp = int(cpos[c]); opos[q] = cp; if (spos) spos[cp] = q; }
  };
  const bool partial = int64_t(k) * 64 <= int64_t(D) && k < D;
  if (partial) {
    if (part != 0) return;                                                     // (serial: one workgroup does it all)
    if (tid == 0) s2_sort_heap(S.w, 0, k);                                     // partial_sort = heap_select + sort_heap
    __syncthreads();
    for (int q = tid; q < k; q += NT) emit(q, T::idx(S.w[q]));
  } else {
    // workgroup `part` of `nparts` answers for one arrival segment of std::sort(q, q + k - 1) (see introsort2)
    introsort2<W, NW, SOLO, COOP>(S, Q, k - 1, emit, tid, part, nparts);
    if (tid == 0 && part == nparts - 1) emit(k - 1, T::idx(S.w[k - 1]));       // the nth_element pivot stays last
  }
}
__host__ inline size_t chan_order_lds(int D, int k, int wbytes) {
  return (sel2_bytes(k, wbytes) + 15) / 16 * 16 + (sort2_bytes(k, 4) + 15) / 16 * 16 + size_t(D) * 2 + 64;
}
struct OrderArgs {          // the ORDER job (all null: none), done by `parts` workgroups side by side
  const float* var_f32; const int* perm; const int* cols; int* order; int* opos; int* spos; int D; int k; int parts;
  const uint32_t* wperm; const uint32_t* wcpos;        // optional: k_chan_select's packed words / cols positions
  int* status;                                         // optional: the pass's status word (selection guard hits)
};
// workgroups for the ORDER job: 2^L arrival segments of >= 128 positions on average
#ifndef VC2_RIDER_PARTS
#define VC2_RIDER_PARTS 64     // rider workgroups of sweep 2 (4 waves each); 16 -> 64: the slowest rider (largest
                               // arrival segment) sets the time: cfg2 131 -> 124 us
#endif
#ifndef VC2_RIDER_PARTS_LONG
#define VC2_RIDER_PARTS_LONG 64     // ... when the sweep is expected to last > 30 us (see rider_parts_max)
#endif
#ifndef VC2_ORDER_PARTS
#define VC2_ORDER_PARTS 64     // workgroups of the stand-alone k_chan_order
#endif
#ifndef VC2_ORDER_MINSEG
#define VC2_ORDER_MINSEG 16
#endif
// How many rider workgroups a sweep-2 launch carries: every rider takes one of the 512 resident workgroup slots from the
// streaming workgroups (and the one it shares a CU with streams 40 % faster than the others: imbalance), while FEWER riders
// take longer (16: ~26 us, 32: ~22, 64: ~20, whatever the clip) -- so a long sweep carries few, a short one many.
// Environment VC2_RIDERS overrides (experiments).
inline int rider_parts_max(int64_t rows, int64_t row_bytes) {
  static const int env = [] { const char* e = getenv("VC2_RIDERS"); return e ? atoi(e) : 0; }();
  if (env > 0) return env;
  const double sweep_us = double(rows) * double(row_bytes) / 6.5e6 + 4.0;      // ~6.5 TB/s + launch ramp
  return sweep_us > 30.0 ? VC2_RIDER_PARTS_LONG : VC2_RIDER_PARTS;
}
__host__ inline int order_parts(int k, int max_parts) {        // a power of two
  int parts = 1;
  while (parts * 2 <= max_parts && parts * 2 <= (k - 1) / VC2_ORDER_MINSEG) parts *= 2;
  return parts;
}

#ifndef VC2_DEV_ONLY
__global__ __launch_bounds__(kOrdNT) void k_chan_order(OrderArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = kOrdNT / 64;
  int bad = 0;
  for (int i = threadIdx.x; i < a.k; i += kOrdNT) bad |= key_fits_u32(a.var_f32[a.perm[i]]) ? 0 : 1;
  if (__syncthreads_or(bad))
    chan_order_body<uint64_t, NW, 4, 8>(smem, a.var_f32, a.D, a.k, a.perm, a.cols, a.order, a.opos, a.spos,
                                        int(blockIdx.x), int(gridDim.x), nullptr, nullptr, a.status);
  else
    chan_order_body<uint32_t, NW, 4, 8>(smem, a.var_f32, a.D, a.k, a.perm, a.cols, a.order, a.opos, a.spos,
                                        int(blockIdx.x), int(gridDim.x), a.wperm, a.wcpos, a.status);
}
#endif

template <int DT>
__global__ void k_gather_cols(const void* __restrict__ x, int64_t /*R*/, int D, const int64_t* __restrict__ idx,
                              int C, void* __restrict__ out) {
  const int64_t r = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= C) return;
  const int64_t c = idx[j];
  if constexpr (Tr<DT>::ES == 4) {
    static_cast<float*>(out)[r * C + j] = static_cast<const float*>(x)[r * D + c];
  } else {
    static_cast<uint16_t*>(out)[r * C + j] = static_cast<const uint16_t*>(x)[r * D + c];
  }
}

// ======================================================================================
// sweeps 2 and 3: one wave per token row over the COMPACTED selected channels
// ======================================================================================
// Only the C selected channels (ascending list cols[C]; nullptr = all D channels) enter the scores,
// but they are scattered at 50 % density over every 128-byte line, so the whole row is streamed:
// each wave DMAs its row straight into LDS (global_load_lds, 16 B per lane, no VGPR round trip) and
// then gathers just the selected elements from LDS -- lane l owns compact positions l, l+64, ... --
// which halves the VALU work of both sweeps.  A workgroup (4 waves) serves rows of ONE frame.
constexpr int kRowWaves = 4;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int DT> __device__ __forceinline__ float lds_elem(const unsigned char* rowbuf, int c) {
  if constexpr (DT == VC2_F32) return reinterpret_cast<const float*>(rowbuf)[c];
  else if constexpr (DT == VC2_BF16)
    return __uint_as_float(uint32_t(reinterpret_cast<const uint16_t*>(rowbuf)[c]) << 16);
  else return float(reinterpret_cast<const _Float16*>(rowbuf)[c]);
}

// the raw bits of one row element in LDS (16-bit types zero-extended), and their fp32 value
template <int DT> __device__ __forceinline__ uint32_t lds_raw(const unsigned char* rowbuf, int c) {
  if constexpr (DT == VC2_F32) return reinterpret_cast<const uint32_t*>(rowbuf)[c];
  else return uint32_t(reinterpret_cast<const uint16_t*>(rowbuf)[c]);
}
template <int DT> __device__ __forceinline__ float raw_to_f32(uint32_t raw) {
  if constexpr (DT == VC2_F32) return __uint_as_float(raw);
  else if constexpr (DT == VC2_BF16) return __uint_as_float(raw << 16);
  else { union { uint16_t u; _Float16 h; } c; c.u = uint16_t(raw); return float(c.h); }
}
// two raw row elements in registers
template <int DT> struct RawPair {
  typedef uint32_t type;
  static __device__ __forceinline__ type make(uint32_t a, uint32_t b) { return a | (b << 16); }
  static __device__ __forceinline__ float lo(type p) {
    if constexpr (DT == VC2_BF16) return __uint_as_float(p << 16); else return raw_to_f32<DT>(p & 0xFFFFu);
  }
  static __device__ __forceinline__ float hi(type p) {
    if constexpr (DT == VC2_BF16) return __uint_as_float(p & 0xFFFF0000u); else return raw_to_f32<DT>(p >> 16);
  }
};
template <> struct RawPair<VC2_F32> {
  struct type { uint32_t a, b; };
  static __device__ __forceinline__ type make(uint32_t a, uint32_t b) { return type{a, b}; }
  static __device__ __forceinline__ float lo(type p) { return __uint_as_float(p.a); }
  static __device__ __forceinline__ float hi(type p) { return __uint_as_float(p.b); }
};
// two T values (both exactly representable in T) kept in registers: fp32 -> a register pair, 16-bit T -> packed
template <int DT> struct CentrePair {
  typedef uint32_t type;
  static __device__ __forceinline__ uint32_t t_bits(float v) {
    if constexpr (DT == VC2_BF16) return __float_as_uint(v) >> 16;
    else { union { uint16_t u; _Float16 h; } c; c.h = _Float16(v); return uint32_t(c.u); }
  }
  static __device__ __forceinline__ type pack(float a, float b) { return t_bits(a) | (t_bits(b) << 16); }
  // (volatile asm: the unpack must stay inside the row loop -- hoisted, it would double the registers again)
  static __device__ __forceinline__ f2_t unpack(type p) {
    f2_t r;
    if constexpr (DT == VC2_BF16)
      asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2" : "=&v"(r.x), "=v"(r.y) : "v"(p));
    else
      asm volatile("v_cvt_f32_f16_e32 %0, %2\n\t"
                   "v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                   : "=&v"(r.x), "=v"(r.y) : "v"(p));
    return r;
  }
};
template <> struct CentrePair<VC2_F32> {
  typedef f2_t type;
  static __device__ __forceinline__ type pack(float a, float b) { return (f2_t){a, b}; }
  static __device__ __forceinline__ f2_t unpack(type p) { return p; }
};

// x^ = RN_f32(v / dn) through one fp64 multiply: inv = RN_f64(1/dn); a quotient of two fp32 numbers is
// never closer than 2^-49 (relative) to a fp32 rounding boundary, the fp64 product is within 2^-52 of
// it, so the final fp64->fp32 rounding lands where the IEEE fp32 division does -- at a third of the
// instructions of v_div_scale/v_rcp/v_fma*4/v_div_fmas/v_div_fixup.
__device__ __forceinline__ float div_via_f64(float v, double inv) { return float(double(v) * inv); }

// Row buffer: the row's D elements followed by one zero element (index D) that padded compact
// positions point at, so the per-element loops need no bounds checks.
__host__ __device__ inline size_t row_lds_bytes(int D, int ES) { return (size_t(D) * ES + 15) / 16 * 16 + 16; }

// Issue the DMA of one row into `rowbuf` (VEC > 1) / copy it synchronously (scalar fallback).
// AUX = cache policy bits of the DMA (0 default, 2 = nt: streamed, not kept).
#ifndef VC2_AUX_S2
#define VC2_AUX_S2 0
#endif
#ifndef VC2_AUX_S3
#define VC2_AUX_S3 0
#endif
template <int DT, int VEC, int AUX = 0>
__device__ __forceinline__ void row_issue(const void* __restrict__ x, int64_t row, int D, int CV,
                                          unsigned char* rowbuf, int lane) {
  constexpr int ES = Tr<DT>::ES;
  const unsigned char* src = static_cast<const unsigned char*>(x) + row * int64_t(D) * ES;
  if constexpr (VEC == 1) {
    for (int c = lane; c < D; c += 64) {
      if constexpr (ES == 4) reinterpret_cast<float*>(rowbuf)[c] = reinterpret_cast<const float*>(src)[c];
      else reinterpret_cast<uint16_t*>(rowbuf)[c] = reinterpret_cast<const uint16_t*>(src)[c];
    }
  } else {
    const int nch = (CV + 63) >> 6;
    for (int j = 0; j < nch; ++j) {
      const int cv = j * 64 + lane;
      if (cv < CV)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + int64_t(cv) * 16), (lds_void_t*)(rowbuf + j * 1024),
                                         16, 0, AUX);
    }
  }
}
// Wait until every DMA this wave has issued has landed in LDS and is visible to all its lanes.
__device__ __forceinline__ void row_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Compact positions a lane owns.  PAIR = 0: p = i*64 + lane (k_norm_fix).  PAIR = 1 (sweep 2): ADJACENT positions two by
// two, p = 128*(i/2) + 2*lane + (i&1): a lane's pair (i, i+1) is two neighbours of the compacted row.
template <int PAIR> __device__ __forceinline__ int compact_pos(int i, int lane) {
  return PAIR ? 128 * (i >> 1) + 2 * lane + (i & 1) : i * 64 + lane;
}
// compact position of this lane -> element index inside the row buffer (D = zero pad)
template <int NPLB, int PAIR = 0>
__device__ __forceinline__ void load_col_offsets(const int* __restrict__ cols, int C, int D, int lane,
                                                 int (&coff)[NPLB]) {
  int t[NPLB];
#pragma unroll
  for (int i = 0; i < NPLB; ++i) {                      // unconditional (clamped) loads: all in flight together
    const uint32_t p = uint32_t(compact_pos<PAIR>(i, lane));
    t[i] = cols ? cols[p < uint32_t(C) ? p : uint32_t(C - 1)] : int(p);
  }
#pragma unroll
  for (int i = 0; i < NPLB; ++i) coff[i] = compact_pos<PAIR>(i, lane) < C ? t[i] : D;
}

// ---- strict mode: torch's own fp32 accumulation order for the tokens where it matters -----------------
// The reference's CPU kernels accumulate ||x||^2 and the squared-distance sums in fp32, in an order fixed
// by torch's 8-lane vector code over the variance-SORTED channel order (order[p]; ReduceOpsKernel.cpp
// norm_kernel_tensor_iterator_impl, SumKernel.cpp cascade_sum; pinned in oracle/vc2_oracle.cpp mode 1).
// Their fp32 noise (<~1e-6 relative) changes the T-rounded result only when the exact value sits that
// close to a T rounding boundary.  So: compute exactly (fp64), and only when the exact value is within
// kFragileUlps fp32-ulps of a boundary replay torch's order for that token -- a few tokens per thousand.
constexpr int kFragileUlpsNorm = 128;   // >= worst-case bound of the 8-chain FMA norm (232 * 2^-24 on the sum)
constexpr int kFragileUlpsDist = 48;    // >= worst-case bound of the cascade sum (~42 fp32 adds per lane)

template <int DT> __device__ __forceinline__ bool near_T_boundary(float y, int margin) {
  if constexpr (DT == VC2_F32) {
    return false;
  } else {
    constexpr int DROP = (DT == VC2_BF16) ? 16 : 13;            // fp32 mantissa bits the cast drops
    const uint32_t bits = __float_as_uint(y);
    if (DT == VC2_F16 && (bits & 0x7F800000u) < (113u << 23)) return true;   // fp16-subnormal result: be safe
    const int low = int(bits & ((1u << DROP) - 1u));
    const int d = low - (1 << (DROP - 1));
    return (d < 0 ? -d : d) <= margin;
  }
}

// Centre means (vidcom2.py:51-52): for half precision torch's mean_out sums an fp32 copy with the outer-reduction
// cascade of SumKernel.cpp, divides by the count in fp32 and casts.  We hold the EXACT sum S (fp64 over T values) and
// q = RN_f32(RN_f32(S) / n); torch's m_t = RN_f32(S_t / n) with |S_t - S| <= k * u * A, where u = 2^-24, A = sum |x^|
// over the reduced rows and k = the roundings along the deepest path of the cascade (standard forward bound of a
// summation tree; A is what the error is relative to, NOT |S| -- under cancellation the two differ by orders of
// magnitude, which is why a margin in ulps of the mean cannot be sound).  So RN_T(m_t) == RN_T(q) unless a T rounding
// boundary lies within
//       delta = (k * A / n + 4 |q|) * u            (the 4: the two fp32 roundings of q, the one of m_t, slack)
// of q -- exactly those means are replayed in torch's order.  A is bounded from what the pass already has, at no
// cost in the sweeps: |x^| <= (1 + 2^-8) |x| / den >= ..., so per frame and column
//       A <= (1 + 2^-7) * sqrt(N * sum_r x[r,c]^2) / min_r den[r]        (Cauchy-Schwarz; sum x^2 from sweep 1's
// shifted partials, the frame's smallest denominator from den[]), and the video's A is the sum of its frames'.
// Cascade depth: level_power lp = max(4, ceil_log2(n) / 4) (SumKernel.cpp); every level adds at most 2^lp - 1 times
// before it dumps, three levels, the top level once per 2^(3 lp) rows, plus the final four adds.
// Default ("torch order", mode 1): an EMPIRICAL margin instead -- 16 fp32-ulps of the mean.  The bound above flags
// 2.5 % of the frame means and 2 % of the video-centre columns of the `drift` workloads (8 % / 49 % on `iid` data, where
// every mean is the residue of a cancellation), against 0.05 % with 16 ulps, and the replays that implies cost the
// target pass +70 us.  Mode 3 uses the bound; the parity suite runs every fixture in both and asserts that they agree
// -- which is the evidence for the empirical margin (the measured spread of torch's cascade around the exact sum is
// < 2 ulp at 25088 rows; the bound assumes every rounding errs the same way).
// Tried in round 3 and NOT adopted as the default: the bound's delta with k replaced by ceil(2 sqrt(k)) (independent
// rounding errors: 3-11 standard deviations).  Being relative to A it widens under cancellation and passes every
// `cancel` fixture, but it replays 11x as many frame means and 8x as many video-centre columns on `drift` data
// (+45 us per pass) and 18 % of the video-centre columns on `iid` data (2.5x the pass): a data-dependent cliff.
// Mode 3 evaluates its margin lazily: |x^| <= 1 gives A <= n, and only a mean with a boundary inside THAT delta goes
// on to fetch sweep 1's partials for its real A.
constexpr int kFragileUlpsMean = 16;
template <int DT> __device__ __forceinline__ bool mean_near_T_boundary(float q) {
  if constexpr (DT == VC2_F32) {
    return false;
  } else if constexpr (DT == VC2_BF16) {
    return near_T_boundary<DT>(q, kFragileUlpsMean);
  } else {
    const float a = fabsf(q);
    if (a < 6.103515625e-05f) {                  // fp16-subnormal result: values on the 2^-24 grid
      const float t = a * 16777216.f;            // exact scaling
      return fabsf((t - floorf(t)) - 0.5f) <= 0.00390625f;   // 2^-8 of a grid step >> 16 fp32 ulps of q
    }
    return near_T_boundary<DT>(q, kFragileUlpsMean);
  }
}
// SumKernel.cpp's level_power for a chain of n elements: 4 up to 2^19, 5 up to 2^23, 6 up to 2^27
__host__ __device__ inline int cascade_lp(int64_t n) {
  int lg = 0;
  while ((int64_t(1) << lg) < n) ++lg;
  return lg / 4 > 4 ? lg / 4 : 4;
}
__host__ __device__ inline int cascade_depth(int64_t n) {
  const int lp = cascade_lp(n);
  return 3 * ((1 << lp) - 1) + int((n >> (3 * lp)) + 1) + 4 + 4;      // (+4: row_sum's interleaved chains, C % 32 tail)
}
// the k of mean_delta (mode 3)
inline double margin_depth(int64_t n, int) { return double(cascade_depth(n)); }
// is a T rounding boundary within `delta` (absolute, >= 0) of q?
template <int DT> __device__ __forceinline__ bool T_boundary_within(float q, float delta) {
  if constexpr (DT == VC2_F32) {
    return false;
  } else {
    if (!(delta == delta) || !(fabsf(q) <= 3.0e38f) || delta >= 3.0e38f) return q == q;   // inf / nan bound: replay (NaN means stay)
    constexpr int DROP = (DT == VC2_BF16) ? 16 : 13;            // fp32 mantissa bits the cast drops
    const float a = fabsf(q);
    if (DT == VC2_F16 && a < 6.103515625e-05f) {                 // fp16-subnormal result: values on the 2^-24 grid
      const float t = a * 16777216.f;                            // exact scaling
      return fabsf((t - floorf(t)) - 0.5f) <= delta * 16777216.f + 1e-6f;
    }
    const uint32_t bits = __float_as_uint(a);
    if ((bits & 0x7F800000u) == 0u) return true;                 // fp32-subnormal mean: be safe
    const float ulp = __uint_as_float(bits & 0x7F800000u) * 1.1920928955078125e-07f;   // 2^(e - 23)
    const int low = int(bits & ((1u << DROP) - 1u));
    const int d = low - (1 << (DROP - 1));
    const float dist = float(d < 0 ? -d : d) * ulp;              // distance to the midpoint above trunc_T(q)
    // (the next midpoint of this binade is 2^DROP ulps further ...
    // ... and next to a power of two the midpoint BELOW is only a quarter of this binade's T-ulp away)
    return dist <= delta || delta >= float(1 << (DROP - 2)) * ulp;
  }
}
template <int DT> __device__ __forceinline__ float mean_delta(float q, double A, int64_t n, double kk) {
  return float((kk * A / double(n) + 4.0 * double(fabsf(q))) * 5.9604644775390625e-08 * 1.01);
}
// per-frame bound of A = sum_r |x^[r, c]| from sweep 1's partials of the frame's groups: s = sum (x - K), q = sum (x - K)^2
// with K the first row of the frame's stat block -> sum x^2 = q + 2 K s + n K^2
__device__ __forceinline__ double abs_sum_bound(double sumsq, int64_t n, float den_min) {
  if (!(sumsq >= 0.0)) sumsq = sumsq != sumsq ? sumsq : 0.0;      // (tiny negative from cancellation -> 0; NaN stays)
  return 1.0078125 * sqrt(double(n) * sumsq * (1.0 + 1e-9)) / (double(den_min) * 0.9921875);   // (den may move by an ulp: k_norm_fix)
}

// sqrt(sum x^2) over the sorted channel order exactly as torch accumulates it (whole wave; same result on
// all lanes).  sv[p] = the row's selected values as fp32, ALREADY in sorted order (p = 0..C-1) in LDS.
template <int DT>
__device__ float norm_torch_order(const float* sv, int C, int lane) {
  float s = 0.f;
  if constexpr (DT == VC2_F16) {                                // generic path: ONE sequential fp32 chain
    if (lane == 0) {
      int p = 0;
      // (round 6) 32 values per step as eight 16-byte LDS reads, the NEXT step's reads in flight under this step's 32 dependent
      // fused multiply-adds: the chain is 1792-2048 FMAs whatever is done, but with 32 single reads and a wait per step the
      // replay of one norm took 12.4 us (fix_riders.py), 942 of them per fp16 pass, and they set k_frame_centres' time
      const float4* sv4 = reinterpret_cast<const float4*>(sv);        // (sv: 16-byte aligned row buffer)
      float4 cur[8], nxt[8];
      if (C >= 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = sv4[u];
      }
      for (; p + 32 <= C; p += 32) {
        const int pn = p + 64 <= C ? p + 32 : p;                   // (last step: re-read this one, unused)
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = sv4[(pn >> 2) + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                             // (fp16 x fp16 is exact in fp32: the fused form rounds like s + v * v, at half the dependent latency)
          s = __builtin_fmaf(cur[u].x, cur[u].x, s); s = __builtin_fmaf(cur[u].y, cur[u].y, s);
          s = __builtin_fmaf(cur[u].z, cur[u].z, s); s = __builtin_fmaf(cur[u].w, cur[u].w, s);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
      }
      for (; p + 8 <= C; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sv[p + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = __builtin_fmaf(v[u], v[u], s);
      }
      for (; p < C; ++p) { const float v = sv[p]; s = __builtin_fmaf(v, v, s); }
    }
    s = __shfl(s, 0, 64);
  } else {                                                      // 8 interleaved FMA chains, then lanes 0..7, then the tail
    constexpr int CH = (DT == VC2_F32) ? 8 : 16;
    const int nv = (C / CH) * CH;
    float acc = 0.f;
    if (lane < 8) {
      int p = lane;
      for (; p + 8 * 31 < nv; p += 8 * 32) {                     // 32 LDS reads in flight per round trip
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = sv[p + 8 * u];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = fmaf(v[u], v[u], acc);
      }
      for (; p + 56 < nv; p += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sv[p + 8 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(v[u], v[u], acc);
      }
      for (; p < nv; p += 8) { const float v = sv[p]; acc = fmaf(v, v, acc); }
    }
    s = __shfl(acc, 0, 64);
#pragma unroll
    for (int j = 1; j < 8; ++j) s = s + __shfl(acc, j, 64);
    for (int p = nv; p < C; ++p) { const float v = sv[p]; s = fmaf(v, v, s); }
  }
  return float(sqrt(double(s)));     // correctly rounded fp32 sqrt (53 >= 2*24+2 bits: no double rounding)
}

// sum_p sq[p] over the sorted channel order exactly as torch's cascade_sum adds it (8 lanes x 4 interleaved
// vectors, 4 cascade levels, halves of each 16-element chunk added first).  sq[p] = RN_T(RN_T(x^_p - c_p)^2)
// as fp32, already in sorted order in LDS.
template <int DT>
__device__ float sum_torch_order(const float* sq, int C, int lane) {
  constexpr int CH = (DT == VC2_F32) ? 8 : 16;
  const int l = lane & 7, k = (lane >> 3) & 3;
  auto load = [&](int c) -> float {
    if constexpr (DT == VC2_F32) return sq[CH * c + l];
    else return sq[CH * c + l] + sq[CH * c + 8 + l];
  };
  const int vec_size = C / CH, size_ilp = vec_size / 4;
  int lg = 0;
  while ((1 << lg) < size_ilp) ++lg;
  const int level_power = max(4, lg / 4);
  const int level_step = 1 << level_power, level_mask = level_step - 1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = 0;
  for (; i + level_step <= size_ilp;) {
    for (int j = 0; j < level_step; j += 4, i += 4) {           // level_step >= 16
      const float t0 = load(i * 4 + k), t1 = load((i + 1) * 4 + k), t2 = load((i + 2) * 4 + k), t3 = load((i + 3) * 4 + k);
      a0 += t0; a0 += t1; a0 += t2; a0 += t3;
    }
    a1 += a0; a0 = 0.f;
    if ((i & (level_mask << level_power)) == 0) {
      a2 += a1; a1 = 0.f;
      if ((i & (level_mask << (2 * level_power))) == 0) { a3 += a2; a2 = 0.f; }
    }
  }
  for (; i < size_ilp; ++i) a0 += load(i * 4 + k);
  a0 += a1; a0 += a2; a0 += a3;
  for (int c = size_ilp * 4; c < vec_size; ++c) { const float t = load(c); if (k == 0) a0 += t; }
  {
    const float t1 = __shfl(a0, l + 8, 64), t2 = __shfl(a0, l + 16, 64), t3 = __shfl(a0, l + 24, 64);
    a0 += t1; a0 += t2; a0 += t3;                                // meaningful on lanes 0..7
  }
  float fin = 0.f;
  for (int p = vec_size * CH; p < C; ++p) fin += sq[p];
#pragma unroll
  for (int j = 0; j < 8; ++j) fin += __shfl(a0, j, 64);
  return fin;
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- "torch order" mode, half precision: fp32 accumulation with a proven bound (ACC = 1) -----------------
// In "torch order" mode every sum whose exact value lies within kFragileUlps* fp32-ulps of a T rounding boundary
// is replayed in torch's own order anyway, so the sweeps need not be exact: any fp32 accumulation whose error is
// bounded by E ulps gives identical bits once the replay margin is widened by E (the value torch computes, the
// exact value and ours all lie on the same side of every boundary farther than kFragile + E from ours).
//   norm:  per lane NPLB fused multiply-adds (x*x is exact in fp32 for 8- / 11-bit significands) in two chains,
//          one add, six tree levels -> relative error <= (NPLB + 7) * 2^-24 on the sum of squares, half of that
//          plus one rounding after the (correctly rounded) square root: <= (NPLB + 7) / 2 + 1 fp32-ulps;
//   dist:  per lane NPLB - 1 adds of non-negative terms and six tree levels: <= NPLB + 5 fp32-ulps.
__host__ __device__ constexpr int acc_norm_ulps(int nplb) { return (nplb + 7) / 2 + 4; }
__host__ __device__ constexpr int acc_dist_ulps(int nplb) { return nplb + 8; }

// bf16 only: x^ = RN_T(RN_f32(x / dn)) through ONE fp32 multiply by r = v_rcp_f32(dn).  The quotient of two bf16
// numbers (8-bit significands mx, md) is never a bf16 rounding midpoint (odd 9-bit M): mx * 2^t == md * M would need
// M | mx; so it stays >= 1 / (md * M) > 2^-17 (relative) away from every midpoint, and a product within 2^-22 of the
// quotient (1 ulp of v_rcp_f32 + the multiply's rounding) rounds to the same bf16 number.  That argument needs a
// NORMAL bf16 result; a row takes this path only when every selected element is zero-free and within
// 2^-63 <= |x| <= 2^50 (then 2^-63 <= dn <= 2^56 and |x^| >= 2^-119); other rows divide exactly (fp64 reciprocal).
constexpr uint32_t kBf16SpanLo = 0x2000u;     // bf16 bits of 2^-63
constexpr uint32_t kBf16SpanLen = 0x3880u;    // ... up to 2^50 (0x5880)

// Strict-mode fix-up of sweep 2: for every queued row replay torch's norm accumulation (the row's selected
// values scattered to their SORTED positions, then the 8-chain / sequential fp32 sum).  Almost always the
// T-rounded norm equals the exactly-rounded one already in den[]; when it does not, den[] is corrected and
// the row is recorded so that k_centres can correct the column sums of its frame.  One wave per entry.
struct NormCorr { int row; float den_old; float den_new; int frame; };   // capacity: one per row (cannot overflow)
// Fused centre launch (k_frame_centres with the fix-ups as rider workgroups): the frame sums are formed WITHOUT the
// corrections, so a frame with a corrected row is put on the pass's replay list as correction entries, one per block of
// 64 columns (frame_replay_wave redoes those means with the corrections); fmark[frame] makes that happen once per
// frame.  list == nullptr: off.
struct FixPush { uint32_t* list; int* count; int cap; int* fmark; };

// The queue of rows whose norm sits next to a T rounding boundary: 8-byte granules (row + 1) | den bits << 32 (den =
// the denominator the sweep computed from the exactly rounded norm), each written by ONE store; the list is zeroed at
// the start of every pass.  (Round 3 tried to let the ORDER riders consume this queue inside the sweep's own launch:
// correct, but never faster than the separate kernel -- NOTES_r03.md.)
__device__ __forceinline__ unsigned long long fixq_pack(int64_t row, float dn) {
  return (static_cast<unsigned long long>(__float_as_uint(dn)) << 32) | static_cast<unsigned long long>(uint32_t(row) + 1u);
}
// ticket words (ints at Plan::o_ticket)
constexpr int kTkFixCount = 2, kTkCorrCount = 3, kTkVcFragile = 5, kTkStatus = 6, kTkFrameReplays = 7;

// One queued row, by one wave: replay torch's norm accumulation (the row's selected values scattered to their SORTED
// positions, then the 8-chain / sequential fp32 sum).  Almost always the T-rounded norm equals the exactly rounded one;
// when it does not, the row is recorded so that k_frame_centres can correct the column sums of its frame.
template <int DT, int VEC, int NPLB>
struct NormFixer {
  int coff[NPLB], sp[NPLB];
  __device__ __forceinline__ void init(const int* __restrict__ cols, const int* __restrict__ spos, int C, int pad_elem, int lane) {
    load_col_offsets<NPLB>(cols, C, pad_elem, lane, coff);
#pragma unroll
    for (int i = 0; i < NPLB; ++i) { const int p = i * 64 + lane; sp[i] = p < C ? (spos ? spos[p] : p) : -1; }
  }
  // the row's DMA into buf0 must have been issued (row_issue); buf0's zero pad element must be in place
  __device__ __forceinline__ void row(unsigned char* buf0, size_t rowb, int64_t row, float dn_old, int C, int N,
                                      float* __restrict__ den, int* __restrict__ corr_count,
                                      NormCorr* __restrict__ corr, int max_entries, int lane,
                                      const FixPush& push = FixPush{nullptr, nullptr, 0, nullptr}) {
    row_wait();
    float xv[NPLB];
#pragma unroll
    for (int i = 0; i < NPLB; ++i) xv[i] = lds_elem<DT>(buf0, coff[i]);
    wave_lds_fence();
    float* sv = reinterpret_cast<float*>(buf0);
#pragma unroll
    for (int i = 0; i < NPLB; ++i) if (sp[i] >= 0) sv[sp[i]] = xv[i];
    wave_lds_fence();
    const float norm = rnT<DT>(norm_torch_order<DT>(sv, C, lane));
    float dn = rnT<DT>(fmaxf(norm, 1e-12f));
    if (norm != norm) dn = norm;
    const bool changed = !(dn == dn_old) && !(dn != dn && dn_old != dn_old);
    if (lane == 0 && changed) {
      den[row] = dn;
      const int j = atomicAdd(corr_count, 1);
      if (j < max_entries) { corr[j].row = int(row); corr[j].den_old = dn_old; corr[j].den_new = dn; corr[j].frame = int(row / N); }
    }
    if (changed && push.list) {                                  // (wave-uniform)
      const int fr = int(row / N);
      const int nbx = (C + 63) / 64;
      int base = -1;
      if (lane == 0 && atomicExch(push.fmark + fr, 1) == 0) base = atomicAdd(push.count, nbx);
      base = __shfl(base, 0, 64);
      if (base >= 0)                                              // one correction entry per block of 64 columns (frame_replay_wave)
        for (int bx = lane; bx < nbx; bx += 64)
          if (base + bx < push.cap) push.list[base + bx] = uint32_t(fr) * uint32_t(nbx) + uint32_t(bx);
    }
    wave_lds_fence();
    if (lane < 4) reinterpret_cast<uint32_t*>(buf0 + rowb - 16)[lane] = 0u;
  }
};

// sweep 2: denominators den[r] = RN_T(max(RN_T(||x_r||), 1e-12f)) (F.normalize, vidcom2.py:48) and the
// per-(frame,split) column sums of x^ = RN_T(x / den) over the selected channels (compact order).
// NPLB = compile-time bound on compact positions per lane (ceil(C/64) <= NPLB).
// ACC = 0: norms accumulated in fp64 (exactly rounded; "exact" mode and fp32 inputs); ACC = 1: see above.
// rflag[r] = 1 marks the rows that divide exactly (consumed by sweep 3; ACC = 1, bf16).
template <int DT, int VEC, int NPLB, int ACC, int RIDER>
__global__ __launch_bounds__(kRowWaves * 64) void k_norm_colsum(const void* __restrict__ x, int N, int D, int CV,
                                                                int C, const int* __restrict__ cols,
                                                                int strict, int S, int q, int64_t R,
                                                                float* __restrict__ den_out, double* __restrict__ part,
                                                                int* __restrict__ tk, unsigned long long* __restrict__ fixq,
                                                                int nfix_cap, uint8_t* __restrict__ rflag,
                                                                OrderArgs rider) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Rider: when an ORDER job is attached, workgroup 0 replays torch.topk's sort of the kept channels (needed only by
  // the kernels AFTER this sweep) while the other workgroups stream -- no side stream, no extra kernel boundary.
  // (RIDER = 0 instantiations carry no ORDER code: it is only ever attached in "torch order" mode.)
  const int nrider = (RIDER && rider.perm) ? (rider.parts & 0xFF) : 0;
  VC2_WGTIME(1, 0);
#ifdef VC2_RIDER_PROBE
  if (int(blockIdx.x) >= nrider && (rider.parts >> 8) == 1) return;     // probe: riders alone
  if (int(blockIdx.x) < nrider && (rider.parts >> 8) == 2) return;      // probe: sweep alone
#endif
  if constexpr (RIDER != 0) {
    if (int(blockIdx.x) < nrider) {
      chan_order_body<uint32_t, kRowWaves, 4, 4>(smem, rider.var_f32, rider.D, rider.k, rider.perm, rider.cols,
                                                 rider.order, rider.opos, rider.spos,    // (16-bit variances: 32-bit words)
                                                 int(blockIdx.x), nrider, rider.wperm, rider.wcpos, rider.status);
      VC2_WGTIME(1, 1);
      return;
    }
  }
  const int bid = int(blockIdx.x) - nrider;
  constexpr int ES = Tr<DT>::ES;
  const size_t rowb = row_lds_bytes(D, ES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // chunk `bid` = rows [bid * q, (bid + 1) * q) of this rank (make_plan); the frames it meets are swept one after the
  // other: segment = the chunk's rows inside one frame, slot = the chunk's number among those that meet the frame
  const int64_t row_a = int64_t(bid) * q, row_b = min(R, row_a + q);
  unsigned char* buf0 = smem + size_t(2 * wave) * rowb;          // [kRowWaves][2][rowb]; later double sacc[C]
  unsigned char* buf1 = buf0 + rowb;
  int coff[NPLB];
  load_col_offsets<NPLB, 1>(cols, C, int((rowb - 16) / ES), lane, coff);
  constexpr bool kFastBf16 = ACC == 1 && DT == VC2_BF16;
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  for (int64_t seg_a = row_a; seg_a < row_b;) {
  const int f = int(seg_a / N);
  const int64_t seg_b = min(row_b, int64_t(f + 1) * N);
  const int sp = bid - int((int64_t(f) * N) / q);
  const int n0 = int(seg_a - int64_t(f) * N), n1 = int(seg_b - int64_t(f) * N);
  if (lane < 4) {                                                // zero pad element of both buffers
    reinterpret_cast<uint32_t*>(buf0 + rowb - 16)[lane] = 0u;
    reinterpret_cast<uint32_t*>(buf1 + rowb - 16)[lane] = 0u;
  }
  double acc[NPLB];
#pragma unroll
  for (int i = 0; i < NPLB; ++i) acc[i] = 0.0;
  int n = n0 + wave;
  if (n < n1) row_issue<DT, VEC, VC2_AUX_S2>(x, int64_t(f) * N + n, D, CV, buf0, lane);
  for (; n < n1; n += kRowWaves) {
    const int64_t row = int64_t(f) * N + n;
    row_wait();
    if (n + kRowWaves < n1) row_issue<DT, VEC, VC2_AUX_S2>(x, row + kRowWaves, D, CV, buf1, lane);
    // the norm rounded to T, the "torch order" queue, the denominator (F.normalize's clamp_min), den[row]
    auto finish_norm = [&](float nrm32, int margin) -> float {
      const float norm = rnT<DT>(nrm32);
      // clamp_min(1e-12) is evaluated in fp32 then cast to T (fp16: 1e-12 -> 0 => 0/0 = NaN, as torch)
      float dn = rnT<DT>(fmaxf(norm, 1e-12f));
      if (norm != norm) dn = norm;
      // strict mode: where the norm sits within a few fp32 ulps of a T rounding boundary, torch's own fp32
      // accumulation order decides the result -> queue the row (a few per thousand) for k_norm_fix
      const bool flagged = strict && (strict == 2 || near_T_boundary<DT>(nrm32, margin));
      if (flagged && lane == 0) {
        const int j = atomicAdd(tk + kTkFixCount, 1);
        if (j < nfix_cap) __hip_atomic_store(fixq + j, fixq_pack(row, dn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) den_out[row] = dn;
      return dn;
    };
    // rflag[row] = 1 marks the rows that divide exactly (consumed by sweep 3; bf16 in "torch order" mode)
    float xv[NPLB];
    bool done = false;
    if constexpr (kFastBf16) {
      // ---- bf16, "torch order" mode: the lane's elements two by two as packed pairs (lo = even slot) ----------
      // VALU issue is what this loop costs (scripts/ubench/valu_rates.hip), so: ONE v_dot2c_f32_bf16 squares and
      // adds a pair, the range test runs on both halves at once (packed 16-bit ops), and the rounded quotients are
      // stored as the v_cvt_pk_bf16_f32 result they come out of.
      typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
      uint32_t P[NPLB / 2];
#pragma unroll
      for (int k = 0; k < NPLB / 2; ++k)
        P[k] = uint32_t(reinterpret_cast<const uint16_t*>(buf0)[coff[2 * k]]) |
               (uint32_t(reinterpret_cast<const uint16_t*>(buf0)[coff[2 * k + 1]]) << 16);
      float ssq = 0.f;
      union { uint32_t u; us2_t h; } span;
      span.u = 0u;
#pragma unroll
      for (int k = 0; k < NPLB / 2; ++k) {
        union { uint32_t u; b2_t b; us2_t h; } pk, lo, t;
        pk.u = P[k];
        ssq = __builtin_amdgcn_fdot2_f32_bf16(pk.b, pk.b, ssq, false);
        // |x| outside [2^-63, 2^50] (zero included) -> the row divides exactly (see kBf16SpanLo)
        pk.u &= 0x7FFF7FFFu;
        lo.u = kBf16SpanLo * 0x10001u;
        t.h = pk.h - lo.h;
        if (128 * k + 127 >= C) {                                 // (padded positions read the zero pad element:
          const int p0 = 128 * k + 2 * lane;                      //  they must not force the exact path)
          t.u &= (p0 < C ? 0xFFFFu : 0u) | (p0 + 1 < C ? 0xFFFF0000u : 0u);
        }
        span.h = __builtin_elementwise_max(span.h, t.h);
      }
      const bool exact_div = __any(span.h.x > kBf16SpanLen || span.h.y > kBf16SpanLen) != 0;
      if (!exact_div) {
        const float nrm32 = float(sqrt(double(wave_sum_bcast_f32(ssq))));   // correctly rounded fp32 square root
        const float dn = finish_norm(nrm32, kFragileUlpsNorm + acc_norm_ulps(NPLB));
        const float r = __builtin_amdgcn_rcpf(dn);
        if (lane == 0 && rflag) rflag[row] = 0;
#pragma unroll
        for (int k = 0; k < NPLB / 2; ++k) {
          union { b2_t b; uint32_t u; } q;
          q.b = __builtin_convertvector((f2_t){__uint_as_float(P[k] << 16) * r, __uint_as_float(P[k] & 0xFFFF0000u) * r}, b2_t);
          acc[2 * k] += double(__uint_as_float(q.u << 16));
          acc[2 * k + 1] += double(__uint_as_float(q.u & 0xFFFF0000u));
        }
        done = true;
      } else {
#pragma unroll
        for (int k = 0; k < NPLB / 2; ++k) { xv[2 * k] = __uint_as_float(P[k] << 16); xv[2 * k + 1] = __uint_as_float(P[k] & 0xFFFF0000u); }
      }
    }
    if (!done) {
      float nrm32;
      int margin = kFragileUlpsNorm;
      if constexpr (ACC == 1 && DT != VC2_BF16) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NPLB; ++i) {
          xv[i] = lds_elem<DT>(buf0, coff[i]);
          if (i & 1) s1 = fmaf(xv[i], xv[i], s1); else s0 = fmaf(xv[i], xv[i], s0);
        }
        nrm32 = float(sqrt(double(wave_sum_bcast_f32(s0 + s1))));
        margin = kFragileUlpsNorm + acc_norm_ulps(NPLB);
      } else {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NPLB; ++i) {
          if constexpr (!kFastBf16) xv[i] = lds_elem<DT>(buf0, coff[i]);
          t = fma(double(xv[i]), double(xv[i]), t);
        }
        nrm32 = float(sqrt(wave_sum_bcast(t)));
      }
      const float dn = finish_norm(nrm32, margin);
      if (kFastBf16 && lane == 0 && rflag) rflag[row] = 1;
      const double inv = 1.0 / double(dn);
#pragma unroll
      for (int i = 0; i < NPLB; ++i) acc[i] += double(rnT<DT>(div_via_f64(xv[i], inv)));
    }
    unsigned char* tbuf = buf0; buf0 = buf1; buf1 = tbuf;
  }
  // combine the 4 waves' column sums in wave order (fixed order): every wave parks its sums in LDS (the row
  // buffers are free now), then each thread adds the 4 values of its columns -- two barriers instead of a
  // serial hand-over per wave
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double* sacc = reinterpret_cast<double*>(smem);                // [kRowWaves][NPLB * 64] doubles
#pragma unroll
  for (int i = 0; i < NPLB; ++i) sacc[wave * NPLB * 64 + compact_pos<1>(i, lane)] = acc[i];
  __syncthreads();
  for (int p = tid; p < C; p += kRowWaves * 64) {
    double t = sacc[p];
#pragma unroll
    for (int w = 1; w < kRowWaves; ++w) t += sacc[w * NPLB * 64 + p];
    part[(int64_t(f) * S + sp) * C + p] = t;
  }
  seg_a = seg_b;
  if (seg_a < row_b) __syncthreads();                            // (the next segment refills the row buffers)
  }
  VC2_WGTIME(1, 1);
}

// ---- sweep 2, streamlined ("v2") -------------------------------------------------------------------------------
// The same results as k_norm_colsum<.., ACC = 1>, for the shapes real models have -- 16-bit rows of NCH full 1 KiB
// chunks (D = 512 NCH), the reference's ratio of one half (C = D / 2 = 64 * 4 NCH: no padded compact positions) -- with
// the row loop rebuilt around what round 4's profiles showed it spends its time on:
//   * the row pipeline.  The row's 4 NCH selected elements per lane are read from LDS into registers FIRST (bf16:
//     ds_read_u16_d16_hi -- the element lands in the high half, i.e. as its fp32 value, no unpack), which frees the row
//     buffer at once: the DMA of the wave's NEXT row is issued into the same buffer before this row is computed -- ONE
//     buffer per wave, the next row on its way during the whole computation, and the pipeline runs THROUGH frame
//     boundaries (the next row may belong to the chunk's next segment: the cross-wave combine of a segment goes through a
//     scratch area, not through the row buffers).  Measured (NOTES_r05.md): two rows in flight per wave, or three
//     workgroups per CU, are no faster -- the sweep is bound by the memory side of its structure, not by rows in flight.
//     Nothing else sits in the vector-memory queue inside the loop: the per-row stores (den, rflag) and the "torch
//     order" queue pushes are parked in lanes (lane j keeps the wave's j-th row) and written once per 64 rows / segment.
//   * instruction issue.  DMA: NCH global_load_lds with immediate offsets from one base (the general kernel runs a
//     14-instruction loop per chunk); LDS addresses: one register per element, computed once (the general kernel
//     recomputes 4 NCH addresses per row); no padding masks; the
//     element range test of the bf16 quotient is ONE v_min3_f32 per pair on the products (a nonzero normal bf16 result is
//     all the midpoint argument above kBf16SpanLo needs; overflow / underflow of the squares shows in the norm itself,
//     which is tested once per row); the correctly rounded square root in fp32.
//   * fp16 gets a fast quotient too: q0 = x * r, e = fma(-dn, q0, x), q = fma(e, r, q0) with r = v_rcp_f32(dn) IS the IEEE
//     fp32 quotient for every nonzero finite fp16 x and dn -- proved by exhaustion (all 1.0e9 pairs, r off by up to +-4
//     ulps: tests/tools/check_f16_quotient.c) -- so RN_f16 of it is what torch's fp16 division returns; the squares are
//     exact fp32 fused multiply-adds in four chains (within the bound of acc_norm_ulps).
// Rows the fast path cannot take (norm outside the range where the fp32 sum of squares is safe, a zero / subnormal
// quotient in bf16, non-finite data) are redone on the spot exactly as the general kernel does them (fp64 sum of squares,
// exactly rounded division; rflag[row] = 1 tells sweep 3).
// ORD geometry (round 6: UNEQUAL pieces).  A wave sweeps whole 16-row blocks of ONE frame, so a frame of nblk = N / 16
// blocks is cut into S_f workgroups of nblk / S_f blocks (+ 1 for the first nblk % S_f): with one S for every frame the
// target shape (12 blocks, 448 workgroup slots for 128 frames) had 3 x 128 = 384 workgroups of 4 blocks -- 64 / 68 rows,
// the busiest CU 132 rows against the row-interleaved form's 112: 34.7 against 30.6 us.  Now the first nA frames take S - 1
// workgroups and the others S: 64 frames x 3 workgroups of 4 blocks + 64 frames x 4 workgroups of 3 blocks = 448 -- 7 blocks
// per CU.  The launch lists the LARGER workgroups first (riders, then the nA frames' workgroups, then the others'): the
// dispatcher hands workgroup i and workgroup i + 256 to the same CU, a large one and a small one.
// The frame's N % 16 leftover rows (their own chain) go to wave 3 of the frame's last workgroup, whether it has blocks or not.
struct OrdGeo { int S, nA; };                                     // S: workgroups of the frames >= nA (the stride of `part`)
__host__ __device__ inline int ord_frame_wgs(const OrdGeo& g, int f) { return f < g.nA ? g.S - 1 : g.S; }
__host__ __device__ inline int ord_total_wgs(const OrdGeo& g, int F) { return g.nA * (g.S - 1) + (F - g.nA) * g.S; }
struct OrdPiece { int f, j, Sf, b0, nb; };                        // workgroup -> frame, piece, its blocks [b0, b0 + nb)
__host__ __device__ inline OrdPiece ord_piece(const OrdGeo& g, int bid, int nblk) {
  OrdPiece o;
  const int na = g.nA * (g.S - 1);
  if (bid < na) { o.Sf = g.S - 1; o.f = bid / o.Sf; o.j = bid - o.f * o.Sf; }
  else { const int t = bid - na; o.Sf = g.S; o.f = g.nA + t / g.S; o.j = t - (o.f - g.nA) * g.S; }
  const int per = nblk / o.Sf, ext = nblk - per * o.Sf;
  o.nb = per + (o.j < ext ? 1 : 0);
  o.b0 = o.j * per + (o.j < ext ? o.j : ext);
  return o;
}

template <int J, int NCH>
__device__ __forceinline__ void s2_issue_chunks(const unsigned char* __restrict__ src_lane, unsigned char* lds_uniform) {
  if constexpr (J < NCH) {                                        // (the immediate offset applies to BOTH addresses; 13 bits signed)
    constexpr int hop = (J >> 2) * 4096, off = (J & 3) * 1024;
    __builtin_amdgcn_global_load_lds((glb_void_t*)(src_lane + hop), (lds_void_t*)(lds_uniform + hop), 16, off, 0);
    s2_issue_chunks<J + 1, NCH>(src_lane, lds_uniform);
  }
}
template <int NCH>
__device__ __forceinline__ void s2_issue_row(const unsigned char* __restrict__ src_lane, unsigned char* lds_uniform) {
  s2_issue_chunks<0, NCH>(src_lane, lds_uniform);
}
__device__ __forceinline__ f2_t pk_fma_f32(f2_t a, f2_t b, f2_t c) {
  f2_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f2_t pk_fnma_f32(f2_t a, f2_t b, f2_t c) {   // c - a * b
  f2_t r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f2_t pk_add_f32(f2_t a, f2_t b) {
  f2_t r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2_t pk_mul_f32(f2_t a, f2_t b) {
  f2_t r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr float kS2NormLoBf16 = 1.8189894035458565e-12f;   // 2^-39 (> 1e-12: clamp_min is the identity above it)
constexpr float kS2NormHiBf16 = 1.099511627776e12f;        // 2^40
constexpr float kS2QuotMinBf16 = 2.3509887016445750e-38f;  // 2^-125: products below it may round to a subnormal bf16
constexpr float kS2NormLoF16 = 6.103515625e-05f;           // 2^-14: dn a normal fp16 number
constexpr float kS2NormHiF16 = 32768.f;

#ifndef VC2_S2_PROBE
#define VC2_S2_PROBE 0       // timing probes (results invalid): 1 memory side only (DMA + LDS reads), 2 compute only (no DMA), 3 DMA only
#endif
#ifndef VC2_S2_WAVES
#define VC2_S2_WAVES 2      // waves per SIMD the register allocation aims at (3: three workgroups per CU)
#endif
constexpr int kS2CombineRounds = VC2_S2_WAVES >= 3 ? 4 : 2;      // (3 workgroups per CU: 4 NCH + 2 NCH KiB of LDS each)
__host__ inline size_t s2v2_lds(int nch) { return size_t(kRowWaves) * nch * 1024 + size_t(kRowWaves) * (4 * nch / kS2CombineRounds) * 64 * 8; }
template <int DT, int NCH, int RIDER, int ORD>
__global__ __launch_bounds__(kRowWaves * 64, VC2_S2_WAVES) void k_norm_colsum2(const void* __restrict__ x, int N, const int* __restrict__ cols,
                                                                 int strict, int S, int q, int64_t R,
                                                                 float* __restrict__ den_out, double* __restrict__ part,
                                                                 int* __restrict__ tk, unsigned long long* __restrict__ fixq,
                                                                 int nfix_cap, uint8_t* __restrict__ rflag, OrderArgs rider,
                                                                 float* __restrict__ bsum, int ord_m) {
  static_assert(DT == VC2_BF16 || DT == VC2_F16, "16-bit inputs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = NCH * 512, C = D / 2, NPLB = NCH * 4, NP = NCH * 2, ROWB = NCH * 1024;
  const int nrider = (RIDER && rider.perm) ? (rider.parts & 0xFF) : 0;
  VC2_WGTIME(1, 0);
  if constexpr (RIDER != 0) {
    if (int(blockIdx.x) < nrider) {
      chan_order_body<uint32_t, kRowWaves, 4, 4>(smem, rider.var_f32, rider.D, rider.k, rider.perm, rider.cols,
                                                 rider.order, rider.opos, rider.spos,
                                                 int(blockIdx.x), nrider, rider.wperm, rider.wcpos, rider.status);
      VC2_WGTIME(1, 1);
      return;
    }
  }
  const int bid = int(blockIdx.x) - nrider;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (rows fit 31 bits: the launch checks R < 2^31; 32-bit uniform arithmetic keeps the index math in a few SALU instructions)
  const int Rr = int(R);
  // ORD (round 5, frames of N <= 512 tokens): the frame means in TORCH'S OWN ORDER, natively.  Workgroup `bid` = piece
  // bid % S of frame bid / S; a wave takes ord_m CONSECUTIVE 16-row blocks of the frame (torch's outer-sum cascade adds
  // a frame's rows in blocks of 16: SumKernel.cpp, level_power 4) and sweeps their rows in order, adding every x^ to a
  // second, fp32, accumulator per column -- sequentially, as torch does -- that is stored per block (bsum[f][block][c];
  // index N / 16: the chain of the N % 16 leftover rows, taken by the wave that owns the last block).
  // k_frame_centres combines the block sums the way torch combines them: EVERY frame mean is torch's bit pattern by
  // construction -- no margins, no list of boundary-near means, no replays (bf16 replayed ~400 of them per pass, fp16
  // ~3300: k_video_centre 39 us at the cfg5 shape).  The fp64 column sums stay: the video centre is formed from them.
  const int nblk = N >> 4, ntail = N & 15;
  int row_a, row_b, ord_f = 0, ord_j = 0, ord_b0 = 0, ord_b1 = 0, ord_r0 = 0, ord_cnt = 0;
  if constexpr (ORD != 0) {
    // (ORD: `q` carries OrdGeo::nA and `ord_m` is unused -- the pieces are unequal, see OrdGeo)
    const OrdPiece pc = ord_piece(OrdGeo{S, q}, bid, nblk);
    ord_f = pc.f; ord_j = pc.j;
    const int m = (pc.nb + kRowWaves - 1) / kRowWaves;            // consecutive blocks per wave
    ord_b0 = min(pc.b0 + pc.nb, pc.b0 + wave * m);
    ord_b1 = min(pc.b0 + pc.nb, ord_b0 + m);
    const bool owns_tail = ntail != 0 && pc.j == pc.Sf - 1 && wave == kRowWaves - 1;
    if (ord_b1 == ord_b0) ord_b0 = ord_b1 = (owns_tail ? nblk : ord_b0);      // (no block: the leftover chain alone, or nothing)
    ord_r0 = ord_f * N + 16 * ord_b0;
    ord_cnt = 16 * (ord_b1 - ord_b0) + (owns_tail ? ntail : 0);
    row_a = ord_f * N; row_b = min(Rr, row_a + N);
    (void)ord_m;
  } else {
    row_a = bid * q; row_b = min(Rr, row_a + q);
  }
  // LDS: [kRowWaves] row buffers (ONE per wave: the row is in registers before the next one is asked for), then the
  // scratch of the cross-wave combine, [kRowWaves][NPLB / 2][64] doubles (= kRowWaves * ROWB bytes), used in two rounds
  unsigned char* wbuf = smem + size_t(wave) * ROWB;
  double* scratch = reinterpret_cast<double*>(smem + size_t(kRowWaves) * ROWB);        // [kRowWaves][NPLB / rounds][64]
  const unsigned char* xl = static_cast<const unsigned char*>(x) + size_t(lane) * 16;
  auto row_src = [&](int row) -> const unsigned char* { return xl + size_t(uint32_t(row)) * (size_t(D) * 2); };
  // this wave's first row at or behind `sa`, the start of a segment of frame `fr` (the wave's rows in a segment [sa, sb)
  // are sa + wave, + 4, ...); -1: none
  auto first_row_from = [&](int sa, int fr) -> int {
    while (sa < row_b) {
      const int sb = min(row_b, (fr + 1) * N);
      if (sa + wave < sb) return sa + wave;
      sa = sb; ++fr;
    }
    return -1;
  };
  const int f_a = ORD ? ord_f : row_a / N;                        // the chunk's first frame
  // the wave's first row is on its way before anything else is fetched
  {
    const int first = ORD ? (ord_cnt > 0 ? ord_r0 : -1) : first_row_from(row_a, f_a);
    if (VC2_S2_PROBE != 2 && first >= 0) s2_issue_row<NCH>(row_src(first), wbuf);
  }
  // LDS byte address of the lane's elements: pair k = compact positions 128 k + 2 lane, + 1
  uint32_t addr[NPLB];
  {
    const uint32_t base = uint32_t(uintptr_t((lds_void_t*)wbuf));
    int2 t[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) t[k] = *reinterpret_cast<const int2*>(cols + 128 * k + 2 * lane);
#pragma unroll
    for (int k = 0; k < NP; ++k) { addr[2 * k] = base + uint32_t(t[k].x) * 2u; addr[2 * k + 1] = base + uint32_t(t[k].y) * 2u; }
  }
  typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const int margin_fast = kFragileUlpsNorm + acc_norm_ulps(NPLB);

  constexpr int RS = ORD ? 1 : kRowWaves;                          // the wave's rows: r0 + RS i, i < cnt
  f2_t chain[ORD ? NP : 1];                                       // (ORD) the fp32 chains of the block under way
  int f = f_a;
  for (int seg_a = row_a; seg_a < row_b; ++f) {
    const int seg_b = min(row_b, (f + 1) * N);
    const int sp = ORD ? ord_j : bid - (f * N) / q;
    const int r0 = ORD ? ord_r0 : seg_a + wave;
    const int cnt = ORD ? ord_cnt : (r0 < seg_b ? (seg_b - r0 + kRowWaves - 1) / kRowWaves : 0);
    double acc[NPLB];
#pragma unroll
    for (int i = 0; i < NPLB; ++i) acc[i] = 0.0;
    // per-row results parked in lanes: lane j keeps row (base + j) of the wave until flush()
    float pk_dn = 0.f;
    uint32_t pk_meta = 0u;                                        // bit 0: norm next to a T rounding boundary, bit 1: exact division
    auto flush = [&](int first, int count) {                      // rows r0 + 4 (first + j), j < count
      const bool mine = lane < count;
      const int row = r0 + RS * (first + lane);
      if (mine) {
        den_out[row] = pk_dn;
        if (rflag) rflag[row] = uint8_t((pk_meta >> 1) & 1u);
      }
      const unsigned long long fm = __ballot(mine && (pk_meta & 1u));
      if (fm) {                                                   // (wave-uniform) one queue reservation for all of them
        int base = 0;
        if (lane == 0) base = atomicAdd(tk + kTkFixCount, __popcll(fm));
        base = __shfl(base, 0, 64);
        const int j = base + __popcll(fm & ((1ull << lane) - 1ull));
        if (mine && (pk_meta & 1u) && j < nfix_cap)
          __hip_atomic_store(fixq + j, fixq_pack(row, pk_dn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    for (int i = 0; i < cnt; ++i) {
      if (VC2_S2_PROBE != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef VC2_DEBUG_TIMING
      if (i == 0 && seg_a == row_a) VC2_WGTIME(4, 0);
#endif
      // (gfx950 runs with SRAM ECC: a d16 load ZEROES the other half of its register instead of keeping it, so a pair
      // cannot be assembled by two loads -- but a bf16 element loaded into the HIGH half IS its fp32 value: no unpack)
      float XA[NP], XB[NP];                                       // the lane's pairs as fp32 values
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (VC2_S2_PROBE == 3) { XA[k] = 1.f; XB[k] = 1.f; continue; }
        if constexpr (DT == VC2_BF16) {
          asm volatile("ds_read_u16_d16_hi %0, %2\n\tds_read_u16_d16_hi %1, %3"
                       : "=&v"(XA[k]), "=&v"(XB[k]) : "v"(addr[2 * k]), "v"(addr[2 * k + 1]) : "memory");
        } else {
          asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %3"
                       : "=&v"(XA[k]), "=&v"(XB[k]) : "v"(addr[2 * k]), "v"(addr[2 * k + 1]) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(XA[k]), "+v"(XB[k]));   // (the loads' results: not before the wait)
      // the row is in registers: its buffer takes the wave's NEXT row (of this segment, else of a later one) right away
      if (VC2_S2_PROBE != 2) {
        const int nr = i + 1 < cnt ? r0 + RS * (i + 1) : (ORD ? -1 : first_row_from(seg_b, f + 1));
        if (nr >= 0) s2_issue_row<NCH>(row_src(nr), wbuf);
      }
      if (VC2_S2_PROBE == 1 || VC2_S2_PROBE == 3) {               // (keep the loads alive, skip the arithmetic)
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) t += XA[k] + XB[k];
        if (t == 12345.678f) pk_dn = t;
        continue;
      }
      if constexpr (DT == VC2_F16) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          union { uint16_t u; _Float16 h; } ca, cb;
          ca.u = uint16_t(__float_as_uint(XA[k])); cb.u = uint16_t(__float_as_uint(XB[k]));
          XA[k] = float(ca.h); XB[k] = float(cb.h);
        }
      }
      // ---- the fast path: sum of squares as exact products in four fp32 chains (the bound of acc_norm_ulps) --------
      f2_t s2 = {0.f, 0.f}, s3 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NP; k += 2) {
        s2 = pk_fma_f32((f2_t){XA[k], XB[k]}, (f2_t){XA[k], XB[k]}, s2);
        s3 = pk_fma_f32((f2_t){XA[k + 1], XB[k + 1]}, (f2_t){XA[k + 1], XB[k + 1]}, s3);
      }
      const float tot = wave_sum_bcast_f32((s2.x + s2.y) + (s3.x + s3.y));
      const float nrm32 = __builtin_sqrtf(tot);                   // correctly rounded (HIP's default for fp32 sqrt)
      constexpr float lo = DT == VC2_BF16 ? kS2NormLoBf16 : kS2NormLoF16, hi = DT == VC2_BF16 ? kS2NormHiBf16 : kS2NormHiF16;
      bool ok = nrm32 >= lo && nrm32 <= hi;                       // (NaN: false)
      float dn = 0.f;
      uint32_t Q[NP];
      if (ok) {
        const float norm = rnT<DT>(nrm32);
        dn = rnT<DT>(fmaxf(norm, 1e-12f));
        const float r = __builtin_amdgcn_rcpf(dn);
        const f2_t r2 = {r, r};
        if constexpr (DT == VC2_BF16) {
          float m = 1.f;
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const f2_t pr = pk_mul_f32((f2_t){XA[k], XB[k]}, r2);
            m = fminf(fminf(fabsf(pr.x), fabsf(pr.y)), m);
            union { b2_t b; uint32_t u; } c;
            c.b = __builtin_convertvector(pr, b2_t);
            Q[k] = c.u;
          }
          ok = !__any(m < kS2QuotMinBf16);
        } else {
          const f2_t d2 = {dn, dn};
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const f2_t xx = {XA[k], XB[k]};
            const f2_t q0 = pk_mul_f32(xx, r2);
            const f2_t e = pk_fnma_f32(d2, q0, xx);
            const f2_t qq = pk_fma_f32(e, r2, q0);
            union { h2_t h; uint32_t u; } c;
            c.h = __builtin_convertvector(qq, h2_t);
            Q[k] = c.u;
          }
        }
      }
      // (ORD) a block's chain starts as an ASSIGNMENT of its first row in torch (a = p[0]; a += p[1] ...): -0.f + v == v
      // for every v, signed zeros included; the leftover rows' chain starts from +0.f (r = 0; r += ...)
      if constexpr (ORD != 0) {
        const int ib = i - 16 * (ord_b1 - ord_b0);                 // >= 0: a leftover row
        if ((ib < 0 && (i & 15) == 0) || ib == 0) {
          const float z = ib == 0 ? 0.f : -0.f;
#pragma unroll
          for (int k = 0; k < NP; ++k) chain[k] = (f2_t){z, z};
        }
      }
      uint32_t meta;
      if (ok) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          f2_t w;
          if constexpr (DT == VC2_BF16) {
            w = (f2_t){__uint_as_float(Q[k] << 16), __uint_as_float(Q[k] & 0xFFFF0000u)};
          } else {
            union { uint32_t u; h2_t h; } c;
            c.u = Q[k];
            w = __builtin_convertvector(c.h, f2_t);
          }
          acc[2 * k] += double(w.x);
          acc[2 * k + 1] += double(w.y);
          if constexpr (ORD != 0) chain[k] = pk_add_f32(chain[k], w);
        }
        meta = (strict == 2 || near_T_boundary<DT>(nrm32, margin_fast)) ? 1u : 0u;
      } else {
        // ---- the row as the general kernel does it (rare) ------------------------------------------------------
        // (one pair at a time -- the empty asm statements pin the order: left to its own devices the scheduler starts all
        // 4 NCH conversions at once and this rare path becomes the kernel's register peak)
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const double a = double(XA[k]), b = double(XB[k]);
          t = fma(a, a, t);
          t = fma(b, b, t);
          if (k + 1 < NP) asm volatile("" : "+v"(XA[k + 1]), "+v"(XB[k + 1]), "+v"(t));
        }
        const float n32 = float(sqrt(wave_sum_bcast(t)));
        const float norm = rnT<DT>(n32);
        dn = rnT<DT>(fmaxf(norm, 1e-12f));
        if (norm != norm) dn = norm;
        const double inv = 1.0 / double(dn);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const float fa = rnT<DT>(div_via_f64(XA[k], inv)), fb = rnT<DT>(div_via_f64(XB[k], inv));
          const double va = double(fa), vb = double(fb);
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[2 * k]) : "v"(va));     // (asm: keeps the two paths' adds apart --
          asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[2 * k + 1]) : "v"(vb)); //  merged, they cost 4 NCH register pairs)
          if constexpr (ORD != 0) chain[k] = pk_add_f32(chain[k], (f2_t){fa, fb});
          if (k + 1 < NP) asm volatile("" : "+v"(XA[k + 1]), "+v"(XB[k + 1]));
        }
        meta = ((strict == 2 || near_T_boundary<DT>(n32, kFragileUlpsNorm)) ? 1u : 0u) | 2u;
      }
      if (lane == (i & 63)) { pk_dn = dn; pk_meta = meta; }
      if ((i & 63) == 63) flush(i - 63, 64);
      if constexpr (ORD != 0) {
        const int ib = i - 16 * (ord_b1 - ord_b0);
        if ((ib < 0 && (i & 15) == 15) || i == cnt - 1) {          // a block (or the leftover chain) is complete
          const int blk = ib < 0 ? ord_b0 + (i >> 4) : nblk;
          float* dst = bsum + (size_t(uint32_t(f * (nblk + 1) + blk))) * C + 2 * lane;
#ifndef VC2_ORD_PROBE_NOSTORE
#pragma unroll
          for (int k = 0; k < NP; ++k) *reinterpret_cast<f2_t*>(dst + 128 * k) = chain[k];
#else
          if (chain[0].x == 12345.678f) *dst = chain[1].x;
#endif
        }
      }
    }
    if (cnt & 63) flush(cnt & ~63, cnt & 63);
#ifdef VC2_DEBUG_TIMING
    if (seg_b == row_b) VC2_WGTIME(4, 1);
#endif
    // Combine the 4 waves' column sums in wave order (fixed order) through the scratch area, 1 / NR of the lane's sums per
    // round; the row buffers are not touched -- the waves' next rows (of the next segment) are landing in them meanwhile.
    constexpr int NR = kS2CombineRounds, PER = NPLB / NR;
    static_assert(NPLB % NR == 0, "combine rounds");
#pragma unroll
    for (int h = 0; h < NR; ++h) {
      if (h) __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) scratch[(wave * PER + i) * 64 + lane] = acc[h * PER + i];
      __syncthreads();
      // scratch[w][i][l] = wave w's sum of the lane-l compact position of slot h PER + i
      for (int t = tid; t < PER * 64; t += kRowWaves * 64) {
        const int i = t >> 6, l = t & 63;
        double v = scratch[t];
#pragma unroll
        for (int w = 1; w < kRowWaves; ++w) v += scratch[w * PER * 64 + t];
        part[size_t(uint32_t(f * S + sp)) * C + compact_pos<1>(h * PER + i, l)] = v;
      }
    }
    seg_a = seg_b;
    if (seg_a < row_b) __syncthreads();                            // (the next segment's combine reuses the scratch)
  }
  VC2_WGTIME(1, 1);
}

// Fix-up of sweep 2: one wave per queued row.
template <int DT, int VEC, int NPLB>
__global__ __launch_bounds__(64) void k_norm_fix(const void* __restrict__ x, int D, int CV, int C,
                                                 const int* __restrict__ cols, const int* __restrict__ spos,
                                                 float* __restrict__ den, const int* __restrict__ nfix_count,
                                                 const unsigned long long* __restrict__ fixq, int max_entries,
                                                 int* __restrict__ corr_count, NormCorr* __restrict__ corr,
                                                 int N, FixPush push) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = Tr<DT>::ES;
  const size_t rowb = row_lds_bytes(D, ES);
  const int lane = threadIdx.x;
  unsigned char* buf0 = smem;
  if (blockIdx.x == 0 && lane == 0) VC2_STAMP(400);
  const int count = min(*nfix_count, max_entries);
  if (blockIdx.x == 0 && lane == 0) VC2_STAMP(401);
  if (int(blockIdx.x) >= count) return;
  if (lane < 4) reinterpret_cast<uint32_t*>(buf0 + rowb - 16)[lane] = 0u;
  unsigned long long g = fixq[blockIdx.x];
  row_issue<DT, VEC>(x, int64_t(uint32_t(g)) - 1, D, CV, buf0, lane);   // first row's DMA overlaps the index loads below
  NormFixer<DT, VEC, NPLB> fx;
  fx.init(cols, spos, C, int((rowb - 16) / ES), lane);
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    if (e != int(blockIdx.x)) { g = fixq[e]; row_issue<DT, VEC>(x, int64_t(uint32_t(g)) - 1, D, CV, buf0, lane); }
    fx.row(buf0, rowb, int64_t(uint32_t(g)) - 1, __uint_as_float(uint32_t(g >> 32)), C, N, den, corr_count, corr,
           max_entries, lane, push);
  }
  if (blockIdx.x == 0 && lane == 0) VC2_STAMP(409);
}

constexpr int kCentreFL = 16;           // frames per centre group (csum_part rows; exchange 2 of the sharded pass)

// ---- "torch order" mode, half precision: torch's fp32 outer-sum cascade for boundary-near centre means -----------
// SumKernel.cpp multi_row_sum over n elements: blocks of 16 added sequentially (acc0), block sums added into
// acc1, acc1 dumped into acc2 every 256 elements, acc2 into acc3 every 4096; finally
// ((tail + acc1) + acc2) + acc3.  That is level_power lp = 4; in general lp = max(4, ceil_log2(n) / 4) (cascade_lp:
// 5 beyond 2^19 elements, 6 beyond 2^23) and blocks, level-1 and level-2 groups all hold B = 2^lp members.
// Every block / group sum is an independent sequential chain.  Work item = one level-1 group (B blocks = B^2
// rows) of one column, done by B lanes: each adds one block's B values in order, then the B block sums are
// added in order.  Modelled: lp <= 6 (B <= 64 = one wave) and at most kL1Cap level-1 groups per chain.
constexpr int kCFixSolo = 512;       // level-1 groups one wave of k_frame_centres may hold (its LDS share)
constexpr int kL1Cap = 8192;         // level-1 groups of a video-centre chain: 2^23 rows at lp = 5, 2^25 at lp = 6
__host__ __device__ inline bool cascade_modelled(int64_t n) {
  const int lp = cascade_lp(n);
  return lp <= 6 && (((n >> lp) + (int64_t(1) << lp) - 1) >> lp) <= kL1Cap;
}

template <int DT>
__device__ __forceinline__ float xhat_at(const void* __restrict__ x, int64_t row, int D, int col,
                                         const float* __restrict__ den) {
  // one quotient per row here, so the IEEE fp32 division itself (same value as div_via_f64, see there)
  return rnT<DT>(ldT<DT>(x, row * D + col) / den[row]);
}

// one level-0 block: the 2^lp elements e0 .. at rows r0 + e*rs added in order (sixteen loads in flight at a time)
template <int DT>
__device__ __forceinline__ float block_sum_rows(const void* __restrict__ x, int D, int col,
                                                const float* __restrict__ den, int64_t r0, int rs, int64_t e0, int lp) {
  float a = 0.f;
  for (int c0 = 0; c0 < (1 << lp); c0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = xhat_at<DT>(x, r0 + (e0 + c0 + u) * rs, D, col, den);
    a = c0 == 0 ? v[0] : a + v[0];
#pragma unroll
    for (int u = 1; u < 16; ++u) a += v[u];
  }
  return a;
}

// ... of NC columns at once: one round of loads (the denominators once) instead of NC rounds; per column the same
// quotients added in the same order as block_sum_rows
template <int DT, int NC>
__device__ __forceinline__ void block_sum_rows_n(const void* __restrict__ x, int D, const int (&col)[NC],
                                                 const float* __restrict__ den, int64_t e0, int lp, float (&a)[NC]) {
  for (int c0 = 0; c0 < (1 << lp); c0 += 16) {
    float dn[16], v[NC][16];
#pragma unroll
    for (int u = 0; u < 16; ++u) dn[u] = den[e0 + c0 + u];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
      for (int u = 0; u < 16; ++u) v[j][u] = ldT<DT>(x, (e0 + c0 + u) * D + col[j]);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
#pragma unroll
      for (int u = 0; u < 16; ++u) v[j][u] = rnT<DT>(v[j][u] / dn[u]);
      a[j] = c0 == 0 ? v[j][0] : a[j] + v[j][0];
#pragma unroll
      for (int u = 1; u < 16; ++u) a[j] += v[j][u];
    }
  }
}

// sum of the level-0 sums of blocks [first_block, first_block + nbl), nbl <= 2^lp, elements at rows r0 + e*rs
template <int DT>
__device__ float wave_l1_group(const void* __restrict__ x, int D, int col, const float* __restrict__ den,
                               int64_t r0, int rs, int64_t first_block, int nbl, int lane, int lp) {
  float a = 0.f;
  if (lane < nbl) a = block_sum_rows<DT>(x, D, col, den, r0, rs, (first_block + lane) << lp, lp);
  float s = __shfl(a, 0, 64);
  for (int u = 1; u < nbl; ++u) s += __shfl(a, u, 64);
  return s;
}

// ((tail + acc1) + acc2) + acc3 from the level-1 values l1[0 .. ceil(nb/16)) (same result on every lane)
template <int DT>
__device__ float wave_cascade_final(const float* l1, int64_t nb, const void* __restrict__ x, int D, int col,
                                    const float* __restrict__ den, int64_t r0, int rs, int64_t n, int lane,
                                    int lp = 4, const float* __restrict__ tail_raw = nullptr) {
  const int B = 1 << lp;
  const int n1c = int(nb >> lp);                              // complete level-1 groups
  const int n2 = n1c >> lp;                                   // complete level-2 groups
  float acc3 = 0.f;
  for (int h0 = 0; h0 < n2; h0 += 64) {
    const int h = h0 + lane;
    float a = 0.f;
    if (h < n2) {
      a = l1[B * h];
      for (int u = 1; u < B; ++u) a += l1[B * h + u];
    }
    const int cnt = min(64, n2 - h0);
    for (int u = 0; u < cnt; ++u) acc3 += __shfl(a, u, 64);
  }
  float acc2 = 0.f;
  for (int g = B * n2; g < n1c; ++g) acc2 += l1[g];
  const float acc1 = (nb & (B - 1)) ? l1[n1c] : 0.f;
  // the < B leftover rows: fetched by as many lanes at once, added in row order
  const int ntail = int(n - (nb << lp));
  const float tv = lane < ntail ? (tail_raw ? tail_raw[lane] : xhat_at<DT>(x, r0 + ((nb << lp) + lane) * rs, D, col, den)) : 0.f;
  float r = 0.f;
  for (int u = 0; u < ntail; ++u) r += __shfl(tv, u, 64);
  r += acc1; r += acc2; r += acc3;
  return r;
}

// one chain (n elements at rows r0 + e*rs) by ONE wave; needs cascade_modelled(n) and as many level-1 groups as l1s holds
template <int DT>
__device__ float wave_cascade_solo(float* l1s, const void* __restrict__ x, int D, int col,
                                   const float* __restrict__ den, int64_t r0, int rs, int64_t n, int lane) {
  const int lp = cascade_lp(n), B = 1 << lp;
  const int64_t nb = n >> lp;
  const int G = int((nb + B - 1) >> lp);
  for (int g = 0; g < G; ++g) {
    const float v = wave_l1_group<DT>(x, D, col, den, r0, rs, int64_t(g) << lp,
                                      int(min<int64_t>(B, nb - (int64_t(g) << lp))), lane, lp);
    if (lane == 0) l1s[g] = v;
  }
  wave_lds_fence();
  const float r = wave_cascade_final<DT>(l1s, nb, x, D, col, den, r0, rs, n, lane, lp);
  wave_lds_fence();
  return r;
}

// a whole column sum in torch's order by one wave: simple cascade, or row_sum's four interleaved chains
template <int DT>
__device__ float wave_column_solo(float* l1s, bool simple, const void* __restrict__ x, int D, int col,
                                  const float* __restrict__ den, int64_t r0, int64_t rows, int lane) {
  if (simple) return wave_cascade_solo<DT>(l1s, x, D, col, den, r0, 1, rows, lane);
  const int64_t q4 = rows >> 2;
  float part[4];
  for (int k = 0; k < 4; ++k) part[k] = wave_cascade_solo<DT>(l1s, x, D, col, den, r0 + k, 4, q4, lane);
  float s = part[0];
  for (int64_t i = q4 << 2; i < rows; ++i) s += xhat_at<DT>(x, r0 + i, D, col, den);
  s += part[1]; s += part[2]; s += part[3];
  return s;
}

// Short chains (n <= kCFixSolo elements, e.g. the N rows of one frame): the whole chain is fetched in ONE round of
// loads into LDS (xs, by the caller) and the cascade runs from there -- blocks of 16 by one lane each, the (at most
// two) level-1 groups through lane reads -- instead of one memory round trip per level-1 group.
// Element e of the chain = xs[off + rs * e].  Same value on every lane.
__device__ __forceinline__ float lds_cascade_short(const float* xs, int off, int rs, int n, int lane) {
  const int nb = n >> 4;                                           // <= 32 blocks
  float a = 0.f;
  if (lane < nb) {
    const float* p = xs + off + rs * 16 * lane;
    a = p[0];
#pragma unroll
    for (int u = 1; u < 16; ++u) a += p[rs * u];
  }
  const int n1c = nb >> 4;                                         // complete level-1 groups (0 .. 2)
  float l1[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int cnt = min(16, nb - 16 * g);
    if (cnt > 0) {
      float t = __shfl(a, 16 * g, 64);
      for (int u = 1; u < cnt; ++u) t += __shfl(a, 16 * g + u, 64);
      l1[g] = t;
    }
  }
  float acc2 = 0.f;
  for (int g = 0; g < n1c; ++g) acc2 += l1[g];
  const float acc1 = (nb & 15) ? l1[n1c] : 0.f;
  float r = 0.f;
  for (int e = nb << 4; e < n; ++e) r += xs[off + rs * e];         // the < 16 leftover elements, in order
  r += acc1; r += acc2; r += 0.f;                                  // (acc3: no complete level-2 group below 4096)
  return r;
}
// a frame's column sum in torch's order from its N <= kCFixSolo values in LDS (see wave_column_solo)
__device__ __forceinline__ float lds_column_short(const float* xs, bool simple, int n, int lane) {
  if (simple) return lds_cascade_short(xs, 0, 1, n, lane);
  const int q4 = n >> 2;
  float part[4];
  for (int k = 0; k < 4; ++k) part[k] = lds_cascade_short(xs, k, 4, q4, lane);
  float s = part[0];
  for (int i = q4 << 2; i < n; ++i) s += xs[i];
  s += part[1]; s += part[2]; s += part[3];
  return s;
}

// ---- centres (vidcom2.py:51-52): two kernels between sweep 2 and sweep 3 ---------------------------------------
// k_frame_centres   workgroup = 64 columns (compact channel space) x 16 frames: frame sums (the S partials of
//                   sweep 2 in order, norm corrections applied) -> frame_center[f][c] = mean_T; the 16 frame sums
//                   added in frame order -> csum_part[g][c] (exchange 2 of the frame-sharded pass);
// k_video_centre    one wave = 64 columns: the group sums added in group order -> vid_center[c] = mean_T.
// "torch order" mode, half precision: a mean with a T rounding boundary inside its error bound (mean_delta) goes on a
// workgroup-local list and is replayed in torch's outer-sum cascade by the same workgroup -- frame entries one wave
// each (video-centre columns: see k_video_centre).
// No global queues, no separate fix-up kernel.
constexpr int kCen2List = 1024;

// ORD (round 5): sweep 2 left, per frame and 16-row block, the fp32 sum of the block's x^ added in row order, and the chain
// of the N % 16 leftover rows (k_norm_colsum2<.., ORD = 1>; bsum[f][N / 16 + 1][C]).  Torch's outer-sum cascade of a frame
// (N <= 512: at most two level-1 groups) from them, exactly as lds_cascade_short combines the same quantities: groups of
// 16 block sums in order (the first one assigned), the complete groups' sums added to acc2 from zero, the partial group
// = acc1, then ((leftover chain + acc1) + acc2) + 0.  A block that holds a row whose denominator a norm fix-up changed
// (corr, nc entries: normally none) is recomputed from x and the final den[].
struct OrdSrc { const float* bsum; int on; int nA; };            // nA: OrdGeo::nA (frames below it have one piece less)
template <int DT>
__device__ __attribute__((noinline)) float ord_frame_sum(const float* __restrict__ bsum, int f, int c, int C, int N, int nc,
                                               const NormCorr* __restrict__ corr, const void* __restrict__ x, int D, int col,
                                               const float* __restrict__ den) {
  const int nb = N >> 4, ntail = N & 15;                          // nb <= 32
  const float* __restrict__ bs = bsum + size_t(uint32_t(f * (nb + 1))) * C + c;
  float v[33];
#pragma unroll
  for (int b = 0; b < 33; ++b) v[b] = bs[size_t(b <= nb ? b : nb) * C];       // (all in flight; entry nb: the leftover chain)
  if (ntail == 0) v[32] = 0.f;                                   // (read below only through index nb)
  unsigned long long dirty = 0ull;
  for (int e0 = 0; e0 < nc; e0 += 8) {                            // (eight entries' loads in flight: one by one they cost a round
    int fr[8], rw[8];                                             //  trip each -- fp16 has ~30 corrected norms per pass, and this scan
#pragma unroll                                                    //  alone made its correction riders 17-33 us long, round 6)
    for (int u = 0; u < 8; ++u) { const int e = e0 + u < nc ? e0 + u : nc - 1; fr[u] = corr[e].frame; rw[u] = corr[e].row; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u < nc && fr[u] == f) { const int b = (rw[u] - f * N) >> 4; dirty |= 1ull << (b < nb ? b : nb); }
  }
  float tail = 0.f;                                              // v[nb] without a run-time register index
#pragma unroll
  for (int b = 0; b < 33; ++b) if (b == nb) tail = ntail ? v[b] : 0.f;
  // (wave-uniform: f is.)  One dirty block at a time; its new sum replaces v[b] through constant-index selects -- a 33-fold
  // unrolled body with sixteen loads each is not unrolled by the compiler, and the run-time index that is left put v[] in scratch
  for (unsigned long long mk = dirty; mk; mk &= mk - 1ull) {
    const int b = int(__builtin_ctzll(mk));
    const int64_t r0 = int64_t(f) * N + 16 * b;
    float xv[16];                                                 // (sixteen loads in flight, then the adds in row order)
    const int nel = b < nb ? 16 : ntail;
#pragma unroll
    for (int u = 0; u < 16; ++u) xv[u] = xhat_at<DT>(x, r0 + (u < nel ? u : nel - 1), D, col, den);
    if (b < nb) {
      float a = xv[0];
#pragma unroll
      for (int u = 1; u < 16; ++u) a += xv[u];
#pragma unroll
      for (int bb = 0; bb < 33; ++bb) v[bb] = bb == b ? a : v[bb];
    } else {
      float r = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) if (u < ntail) r += xv[u];
      tail = r;
    }
  }
  float acc2 = 0.f, acc1 = 0.f;
  const int n1c = nb >> 4;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int cnt = nb - 16 * g < 16 ? nb - 16 * g : 16;
    if (cnt > 0) {
      float t = v[16 * g];
#pragma unroll
      for (int u = 1; u < 16; ++u) if (u < cnt) t += v[16 * g + u];
      if (g < n1c) acc2 += t; else acc1 = t;
    }
  }
  float r = tail;
  r += acc1; r += acc2; r += 0.f;
  return r;
}

// ... the same cascade when no row of the frame had its norm corrected (nc == 0: always in the fused centre launch, whose
// corrections arrive as entries of the next launch) -- INLINED and sized by the frame: the general form above is a real call
// whose callee-saved registers go to scratch (32 VGPRs stored and reloaded per thread: 58 MB of scratch traffic per pass,
// +8 us on k_frame_centres -- round 6), and it keeps 33 values live where a 196-token frame has 13.
template <int NB>      // NB >= nb + 1: 16 (N < 256) or 33
__device__ __forceinline__ float ord_frame_sum_clean(const float* __restrict__ bsum, int f, int c, int C, int N) {
  const int nb = N >> 4, ntail = N & 15;
  const float* __restrict__ bs = bsum + size_t(uint32_t(f * (nb + 1))) * C + c;
  float v[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) v[b] = bs[size_t(b <= nb ? b : nb) * C];        // (all in flight; entry nb: the leftover chain)
  float tail = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) if (b == nb) tail = ntail ? v[b] : 0.f;
  float acc2 = 0.f, acc1 = 0.f;
  const int n1c = nb >> 4;
#pragma unroll
  for (int g = 0; g < (NB > 16 ? 2 : 1); ++g) {
    const int cnt = nb - 16 * g < 16 ? nb - 16 * g : 16;
    if (cnt > 0) {
      float t = v[16 * g];
#pragma unroll
      for (int u = 1; u < 16; ++u) if (16 * g + u < NB && u < cnt) t += v[16 * g + u];
      if (g < n1c) acc2 += t; else acc1 = t;
    }
  }
  float r = tail;
  r += acc1; r += acc2; r += 0.f;
  return r;
}

// where sweep 1's partials of this rank's frames live (part == nullptr: none): group g = piece g % splits of frame
// g / splits holds (sum (x - K), sum (x - K)^2) per channel, K = the first row of the frame's stat block
struct FrameStatSrc {
  const double* part; int splits; int block_frames; int N; int* diag;
  template <int DT> __device__ __forceinline__ double sumsq(int f, int col, const void* __restrict__ x, int D) const {
    const float Kf = ldT<DT>(x, int64_t(f / block_frames) * block_frames * N * D + col);   // (issued with the first partials)
    double s1 = 0.0, s2 = 0.0;
    for (int g = f * splits; g < (f + 1) * splits; ++g) {
      const double a = part[(int64_t(g) * 2 + 0) * D + col], b = part[(int64_t(g) * 2 + 1) * D + col];
      s1 += a;
      s2 += b;
    }
    const double K = double(Kf);
    return s2 + 2.0 * K * s1 + double(N) * K * K;
  }
};

// Does the frame mean q = RN_f32(RN_f32(exact sum) / N) of (frame f, channel col) have to be replayed in torch's order?
// (see kFragileUlpsMean / mean_delta; ab: the bound of sum |x^| used, for exchange 2 of the frame-sharded pass)
template <int DT>
__device__ __forceinline__ bool frame_mean_near(float q, bool bounded, bool all, int strict, double kk, double kk_a,
                                                bool want_bounds, const FrameStatSrc& fs, int f, int col,
                                                const void* __restrict__ x, int D, int N, float dmin, double& ab,
                                                const double* ssq_pre = nullptr) {
  if (!bounded) return all || mean_near_T_boundary<DT>(q);       // the empirical margin alone
  // A >= sum_r |x^[r, c]| over the frame (see mean_delta).  |x^| <= 1 gives A <= N: only a mean with a boundary
  // inside that margin fetches sweep 1's partials for the real one (want_bounds: the frame-sharded pass ships
  // every group's bound with exchange 2, so all of them)
  bool near = all || (strict == 3 ? T_boundary_within<DT>(q, mean_delta<DT>(q, double(N), N, kk))
                                  : mean_near_T_boundary<DT>(q));
  const bool pre_a = !near && strict != 3 && kk_a > 0.0 && T_boundary_within<DT>(q, mean_delta<DT>(q, double(N), N, kk_a));
  if ((want_bounds || (near && strict == 3) || pre_a) && fs.part) {
    const double b = abs_sum_bound(ssq_pre ? *ssq_pre : fs.sumsq<DT>(f, col, x, D), N, dmin);
    ab = b < double(N) ? b : double(N);            // (NaN: N)
    if (near && !all && strict == 3) near = T_boundary_within<DT>(q, mean_delta<DT>(q, ab, N, kk));
    if (pre_a) near = T_boundary_within<DT>(q, mean_delta<DT>(q, ab, N, kk_a));
  } else {
    ab = double(N);
  }
  return near;
}

// FUSED centre launch (round 4; single rank, 16-bit inputs, vector path): the norm fix-ups of sweep 2 (k_norm_fix's work)
// run as RIDER workgroups of this launch -- its first fy grid rows, kFixWaves waves each, one queue entry per wave, row
// buffers in dynamic LDS -- next to the frame workgroups, and NOBODY WAITS for anybody: the frame sums are formed without
// the corrections; a rider that corrects a norm puts all means of that frame on the replay list (FixPush) and the
// next launch (k_video_centre) adds the corrections to the video-centre sums.  One kernel boundary and the fix-ups'
// latency chain (~7.5 us) off the critical path.  (An earlier form of this round had the frame workgroups WAIT for the
// riders through an agent-scope counter: 26.7 us against 20.3 for the two kernels back to back -- every waiting
// workgroup's acquire empties its XCD's L2.)
struct FixRiders {
  int fy;                                   // grid rows of rider workgroups (0: none -- k_norm_fix ran before this launch)
  int waves;                                // waves of a rider workgroup that take queue entries (the others leave at once)
  int CV, max_entries;
  const int* nfix_count; const unsigned long long* fixq;
  int* corr_count; NormCorr* corr; float* den;
  FixPush push;
};
#ifndef VC2_FIX_WAVES_DEFAULT
#define VC2_FIX_WAVES_DEFAULT 8      // (8 = two chains per SIMD: a fp16 norm replay is ONE sequential 2048-step chain)
#endif
constexpr int kFixWaves = VC2_FIX_WAVES_DEFAULT;               // (default; FixRiders::waves)
constexpr int kTkFixEntries = 8;             // ticket word: correction entries of the pass (FixPush)
constexpr int kTkSelArrive = 9;              // ticket word: k_select workgroups that have finished their selection (host mirror)

template <int DT, int VEC, int NPLB>        // NPLB = 0: no rider code compiled in
__global__ __launch_bounds__(64 * kCentreFL) void k_frame_centres(const double* __restrict__ part, int F, int S, int S_q, int N,
                                                                   int C, float* __restrict__ fc,
                                                                   double* __restrict__ csum_part,
                                                                   const void* __restrict__ x, int D,
                                                                   const int* __restrict__ cols,
                                                                   const int* __restrict__ spos,
                                                                   const float* den,
                                                                   const int* __restrict__ corr_count,
                                                                   const NormCorr* __restrict__ corr, int strict,
                                                                   int* __restrict__ vtick, FrameStatSrc fs,
                                                                   double kk, int want_bounds,
                                                                   float* __restrict__ dmin_out, double kk_a,
                                                                   uint32_t* __restrict__ rlist, int rcap, FixRiders fr,
                                                                   OrdSrc ord = OrdSrc{nullptr, 0, 0}) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fix_rows[];     // riders: kFixWaves row buffers
  __shared__ double sm[kCentreFL][64];
  __shared__ double sb[kCentreFL][64];
  __shared__ uint32_t flist[kCen2List];            // local frame * 64 + local column
  __shared__ int count, rbase;
  const int tid = threadIdx.x, cl = tid & 63, fl = tid >> 6, lane = cl, wave = fl;
  const int fy = NPLB > 0 ? fr.fy : 0;
  if constexpr (NPLB > 0) {
    if (int(blockIdx.y) < fy) {                                  // ---- a rider workgroup: one queue entry per wave
      constexpr int ES = Tr<DT>::ES;
      const size_t rowb = row_lds_bytes(D, ES);
      const size_t stride = (std::max(rowb, size_t(C) * 4 + 16) + 15) / 16 * 16;
      if (wave >= fr.waves) return;
      const int nslots = fy * int(gridDim.x) * fr.waves;
      const int first = (int(blockIdx.y) * int(gridDim.x) + int(blockIdx.x)) * fr.waves + wave;
      // (the wave's first entry is asked for TOGETHER with the count: behind the early return it was a second round trip --
      //  1.5 us of a replay that is ~8 us and sets this launch's time)
      unsigned long long q_first = fr.fixq[first < fr.max_entries ? first : 0];
      const int cnt = min(*fr.nfix_count, fr.max_entries);
      asm volatile("" : "+v"(q_first));
      if (first >= cnt) return;
#ifdef VC2_DEBUG_TIMING
      if (lane == 0 && first < 4096) g_dbg_wg[5][0][first] = wall_clock64();
#endif
      unsigned char* buf0 = fix_rows + size_t(wave) * stride;
      if (lane < 4) reinterpret_cast<uint32_t*>(buf0 + rowb - 16)[lane] = 0u;
      unsigned long long q = q_first;
      row_issue<DT, VEC>(x, int64_t(uint32_t(q)) - 1, D, fr.CV, buf0, lane);       // the first row's DMA overlaps the index loads
      NormFixer<DT, VEC, NPLB> fx;
      fx.init(cols, spos, C, int((rowb - 16) / ES), lane);
      for (int e = first; e < cnt; e += nslots) {
        if (e != first) { q = fr.fixq[e]; row_issue<DT, VEC>(x, int64_t(uint32_t(q)) - 1, D, fr.CV, buf0, lane); }
        fx.row(buf0, rowb, int64_t(uint32_t(q)) - 1, __uint_as_float(uint32_t(q >> 32)), C, N, fr.den, fr.corr_count, fr.corr,
               fr.max_entries, lane, fr.push);
      }
#ifdef VC2_DEBUG_TIMING
      if (lane == 0 && first < 4096) g_dbg_wg[5][1][first] = wall_clock64();
#endif
      return;
    }
  }
#ifdef VC2_DEBUG_TIMING
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == fy) g_dbg_wg[5][0][4095] = wall_clock64();     // (a frame workgroup's begin: the launch's clock zero)
#endif
  const int c = blockIdx.x * 64 + cl;
  const int g = int(blockIdx.y) - fy;
  const int FGn = int(gridDim.y) - fy;                           // frame groups of this launch
  const int f = g * kCentreFL + fl;
  const bool replay = strict != 0 && DT != VC2_F32;
  const bool all = strict == 2;
  if (tid == 0) {
    count = 0;
    if (g == 0) vtick[blockIdx.x] = 0;             // k_video_centre's arrival ticket of this column block
    if (g == 0 && blockIdx.x == 0) VC2_STAMP(500);
  }
  // the frame's smallest denominator (wave fl = frame f): |x^| <= |x| / it.  (The loads are in flight with the ones below.)
  // kk_a > 0 (mode 4 "robust", the pass's own sweep-1 partials at hand): the empirical margin OR a boundary within
  // (kk_a A / n + 4 |q|) u -- the term that grows under cancellation (see the note at kFragileUlpsMean)
  const bool bounded = replay && (strict == 3 || want_bounds || kk_a > 0.0);   // margins relative to sum |x^|
  // (wave fl reads its frame's N denominators -- in flight with the partial sums below.  A denominator that a rider of
  //  this launch is correcting by an ulp may be read either way: abs_sum_bound allows for that.  Rounds 3 / early 4 had
  //  sweep 2 leave the minimum behind through atomics: +2 us on the sweep.)
  // Every load this workgroup's results hang on is issued up front -- the channel index (the head of the only
  // dependent chain), the frame's partial sums, sweep 1's partials for the A-relative margin (18 bytes per lane that
  // more than half of the waves would otherwise fetch three dependent round trips later, when one of their lanes meets
  // a candidate), the denominators -- two round trips instead of five.
  // (UNCONDITIONAL loads at clamped addresses: a load inside a divergent conditional is waited for on the spot)
  const bool active = c < C && f < F;
  const int cq = min(c, C - 1), fq = min(f, F - 1);
  const int col = cols ? cols[cq] : cq;
  // this frame's segments: one per sweep-2 chunk that meets it (make_plan)
  const int Sf = ord.on ? (fq < ord.nA ? S - 1 : S) : int((int64_t(fq + 1) * N - 1) / S_q) - int((int64_t(fq) * N) / S_q) + 1;   // (ORD: OrdGeo)
  double v0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v0[u] = part[(int64_t(fq) * S + min(u, Sf - 1)) * C + cq];
  float dv[4] = {INFINITY, INFINITY, INFINITY, INFINITY};       // (the frame's first 256 denominators: wave fl = frame f)
  if (bounded) {                                                // (kernel-uniform)
#pragma unroll
    for (int i = 0; i < 4; ++i) dv[i] = den[int64_t(fq) * N + min(lane + 64 * i, N - 1)];
  }
  const bool pre_stats = bounded && fs.part != nullptr;         // (kernel-uniform)
  double ssq_pre = 0.0;
  if (pre_stats) ssq_pre = fs.template sumsq<DT>(fq, col, x, D);  // (the first use of `col`: everything above is in flight)
  float dmin = INFINITY;
  if (bounded && f < F) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dmin = fminf(dmin, fabsf(dv[i]));                                    // (NaN does not enter)
    for (int r = lane + 256; r < N; r += 64) dmin = fminf(dmin, fabsf(den[int64_t(f) * N + r]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o, 64));
  }
  __syncthreads();
  double sf = 0.0, ab = 0.0;
  float q = 0.f;
  if (active) {
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < Sf) sf += v0[u];
    for (int s0 = 8; s0 < Sf; s0 += 8) {                        // (more than 8 partials: further batches, added in split order)
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t(f) * S + min(s0 + u, Sf - 1)) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (s0 + u < Sf) sf += v[u];
    }
    const int nc = corr_count ? *corr_count : 0;
    for (int e = 0; e < nc; ++e) {                              // rows whose norm k_norm_fix corrected (normally none)
      if (corr[e].frame == f) {
        const float v = ldT<DT>(x, int64_t(corr[e].row) * D + col);
        const float xo = rnT<DT>(div_via_f64(v, 1.0 / double(corr[e].den_old)));
        const float xn = rnT<DT>(div_via_f64(v, 1.0 / double(corr[e].den_new)));
        sf += double(xn) - double(xo);
      }
    }
    if (ord.on) {
      float s;
      if (nc == 0) s = N < 256 ? ord_frame_sum_clean<16>(ord.bsum, f, c, C, N) : ord_frame_sum_clean<33>(ord.bsum, f, c, C, N);
      else s = ord_frame_sum<DT>(ord.bsum, f, c, C, N, nc, corr, x, D, col, den);
      fc[int64_t(f) * C + c] = rnT<DT>(s / float(N));
    }
    else fc[int64_t(f) * C + c] = mean_T<DT>(sf, N);
    q = float(sf) / float(N);
  }
  if (tid == 0 && g == 0 && blockIdx.x == 0) VC2_STAMP(501);
  if (replay && !(ord.on && !bounded)) {                          // (ORD: the means are final; only the bounds of sum |x^| may be wanted)
    if (bounded && dmin_out && blockIdx.x == 0 && cl == 0 && f < F) dmin_out[f] = dmin;     // (k_video_centre's lazy bounds)
    if (active && frame_mean_near<DT>(q, bounded, all, strict, kk, kk_a, want_bounds != 0, fs, f, col, x, D, N,
                                      dmin, ab, pre_stats ? &ssq_pre : nullptr)) {
      if (!ord.on) {
        const int j = atomicAdd(&count, 1);
        if (j < kCen2List) flist[j] = uint32_t(fl) * 64u + uint32_t(cl);
      }
    }
  }
  if (tid == 0 && g == 0 && blockIdx.x == 0) VC2_STAMP(503);
  sm[fl][cl] = sf;
  sb[fl][cl] = ab;
  __syncthreads();
  if (fl == 0 && c < C) {
    double t = 0.0, tb = 0.0;
#pragma unroll
    for (int i = 0; i < kCentreFL; ++i) { t += sm[i][cl]; tb += sb[i][cl]; }
    csum_part[int64_t(g) * C + c] = t;
    csum_part[(int64_t(FGn) + g) * C + c] = tb;    // second half of the buffer: the groups' bounds (zeros unless want_bounds:
                                                   // exchange 2 of the frame-sharded pass ships both halves)
  }
  if (tid == 0 && g == 0 && blockIdx.x == 0) VC2_STAMP(505);
  if (!replay || (((N >> 4) + 15) >> 4) > kCFixSolo) return;
  // ---- the boundary-near frame means are replayed in torch's cascade order by the NEXT launch (rider waves of
  //      k_video_centre, frame_replay_wave): this workgroup only appends its entries (frame * C + column) to the pass's
  //      list.  Replaying them here, one wave per entry, left the launch waiting for the few workgroups whose 64
  //      columns hold most of the near-zero means (two or three rounds of ~2.5 us) while 3000 other waves idled.
  const int nf = min(count, kCen2List);
  if (tid == 0) rbase = nf ? atomicAdd(fs.diag, nf) : 0;        // (the list's length = replayed frame means of this pass)
  __syncthreads();
  for (int e = tid; e < nf; e += 64 * kCentreFL) {
    const int ff = g * kCentreFL + int(flist[e] >> 6), cc = blockIdx.x * 64 + int(flist[e] & 63u);
    if (rbase + e < rcap) rlist[rbase + e] = uint32_t(ff) * uint32_t(C) + uint32_t(cc);
  }
  if (tid == 0 && g == 0 && blockIdx.x == 0) VC2_STAMP(509);
}

// One boundary-near frame mean per wave (see k_frame_centres): the frame's N values of x^ in ONE round of loads, then
// torch's outer-sum cascade from LDS (N <= kCFixSolo), else level-1 group by level-1 group.  xs: kCFixSolo floats of LDS.
// The second list (fused centre launch only) holds CORRECTION entries: (frame f, block of 64 columns) of a frame
// that holds a row whose norm a fix-up rider corrected -- k_frame_centres formed that frame's sums without the
// correction.  The wave redoes k_frame_centres' arithmetic for those 64 means with the corrections added (same sum,
// same margins: frame_mean_near) and replays the ones that need it on the spot.
struct FrameFix {
  const double* part; int S, S_q, strict; double kk, kk_a; FrameStatSrc fs;
  const int* corr_count; const NormCorr* corr;
  OrdSrc ord;
};
struct FrameReplay {
  const uint32_t* list; const int* count; int cap;   // entries frame * C + column (count may exceed cap: never written beyond)
  float* fc; int N;
  FrameFix fix;
  const uint32_t* list2; const int* count2; int cap2;   // correction entries frame * nbx + column block: taken FIRST (the
                                                        // longest items -- a block's sums, margins, then its replays one by one)
  const int* fmark;    // per frame: non-zero = the frame has correction entries, which redo ALL of its means (with the corrected
};                     //   sums) -- its regular entries are skipped: two waves would otherwise store the same fc word in either order
template <int DT>
__device__ __forceinline__ void frame_replay_wave(const FrameReplay& r, int rid, int nrid, float* xs,
                                                  const void* __restrict__ x, int D, int C,
                                                  const int* __restrict__ cols, const int* __restrict__ spos,
                                                  const float* __restrict__ den, int lane) {
  const int N = r.N;
  const int group = C >= 8 ? 32 : 4;
  const int simple_end = (C / group) * group;
  // (round 6: the wave's first entry of either list, the number of corrected norms and its column index are asked for TOGETHER
  //  with the two counts -- behind them they were a second and a third round trip of a chain of five)
  uint32_t ent2_first = r.list2 ? r.list2[rid < r.cap2 ? rid : 0] : 0u;
  uint32_t ent1_first = r.list[rid < r.cap ? rid : 0];
  int nc_all = r.fix.corr_count ? *r.fix.corr_count : 0;
  const int cnt1 = min(*r.count, r.cap), cnt2 = r.list2 ? min(*r.count2, r.cap2) : 0;
  asm volatile("" : "+v"(ent2_first), "+v"(ent1_first), "+v"(nc_all));
  auto replay_one = [&](int ff, int cc) {                          // (wave-uniform arguments)
    const int col = cols ? cols[cc] : cc, sp = spos ? spos[cc] : cc;
    float s;
    if (N <= kCFixSolo) {                                         // the frame's values in one round of loads, then LDS
      float v[kCFixSolo / 64];
#pragma unroll
      for (int i = 0; i < kCFixSolo / 64; ++i) {
        const int rr = min(lane + 64 * i, N - 1);
        v[i] = xhat_at<DT>(x, int64_t(ff) * N + rr, D, col, den);
      }
#pragma unroll
      for (int i = 0; i < kCFixSolo / 64; ++i) if (lane + 64 * i < N) xs[lane + 64 * i] = v[i];
      wave_lds_fence();
      s = lds_column_short(xs, sp < simple_end, N, lane);
      wave_lds_fence();
    } else {
      s = wave_column_solo<DT>(xs, sp < simple_end, x, D, col, den, int64_t(ff) * N, N, lane);
    }
    if (lane == 0) r.fc[int64_t(ff) * C + cc] = rnT<DT>(s / float(N));
  };
#ifdef VC2_DEBUG_TIMING
  int dbg_entries = 0;
  if (lane == 0 && rid < 4096) { g_dbg_vc[3][rid] = 0ull; g_dbg_vc[4][rid] = (unsigned long long)cnt1 | ((unsigned long long)cnt2 << 32); g_dbg_vc[5][rid] = 0ull; }
#endif
  for (int e = rid; e < cnt1 + cnt2; e += nrid) {
#ifdef VC2_DEBUG_TIMING
    ++dbg_entries;
    if (lane == 0 && rid < 4096) { g_dbg_vc[3][rid] = (unsigned long long)dbg_entries; if (dbg_entries == 2) g_dbg_vc[5][rid] = wall_clock64(); }
#endif
    if (e >= cnt2) {
      const uint32_t ent = (e == rid && cnt2 == 0) ? ent1_first : r.list[e - cnt2];
      const int ff = int(ent / uint32_t(C));
      if (r.fmark && cnt2 > 0 && r.fmark[ff] != 0) continue;      // (its correction entries answer for the whole frame)
      replay_one(ff, int(ent - uint32_t(ff) * uint32_t(C)));
      continue;
    }
    // ---- a correction entry: lane = column bx * 64 + lane of frame f
    const uint32_t ent = e == rid ? ent2_first : r.list2[e];
    const FrameFix& m = r.fix;
    const int nbx = (C + 63) / 64;
    const int f = int(ent / uint32_t(nbx)), bx = int(ent - uint32_t(f) * uint32_t(nbx));
    const int c = bx * 64 + lane;
    const bool active = c < C, all = m.strict == 2;
    if (m.ord.on) {                                               // ORD: the block sums, the corrected rows' blocks redone
      if (active) {
        const int col = cols ? cols[c] : c;
        r.fc[int64_t(f) * C + c] = rnT<DT>(ord_frame_sum<DT>(m.ord.bsum, f, c, C, N, nc_all, m.corr, x, D, col, den) / float(N));
      }
      continue;
    }
    const bool bounded = m.strict == 3 || m.kk_a > 0.0;
    float dmin = INFINITY;
    if (bounded) {
      for (int rr = lane; rr < N; rr += 64) dmin = fminf(dmin, fabsf(den[int64_t(f) * N + rr]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o, 64));
    }
    bool near = false;
    if (active) {
      double sf = 0.0, ab = 0.0;
      const int Sf = int((int64_t(f + 1) * N - 1) / m.S_q) - int((int64_t(f) * N) / m.S_q) + 1;
      for (int s0 = 0; s0 < Sf; s0 += 8) {                        // (k_frame_centres' sum: split order)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = m.part[(int64_t(f) * m.S + min(s0 + u, Sf - 1)) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (s0 + u < Sf) sf += v[u];
      }
      const int col = cols ? cols[c] : c;
      const int nc = *m.corr_count;
      for (int e0 = 0; e0 < nc; e0 += 8) {                        // (eight entries' loads in flight; added in entry order)
        NormCorr nc8[8];
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nc8[u] = m.corr[min(e0 + u, nc - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = ldT<DT>(x, int64_t(nc8[u].row) * D + col);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (e0 + u < nc && nc8[u].frame == f) {
            const float xo = rnT<DT>(div_via_f64(v8[u], 1.0 / double(nc8[u].den_old)));
            const float xn = rnT<DT>(div_via_f64(v8[u], 1.0 / double(nc8[u].den_new)));
            sf += double(xn) - double(xo);
          }
        }
      }
      r.fc[int64_t(f) * C + c] = mean_T<DT>(sf, N);
      near = frame_mean_near<DT>(float(sf) / float(N), bounded, all, m.strict, m.kk, m.kk_a, false, m.fs, f, col, x, D, N, dmin, ab);
    }
    wave_lds_fence();
    for (uint64_t mk = __ballot(near); mk; mk &= mk - 1) replay_one(f, bx * 64 + int(__builtin_ctzll(mk)));
  }
}
// the replays as their own launch (frame-sharded pass: its video centre is computed later, from the all-gathered sums)
template <int DT>
__global__ __launch_bounds__(64) void k_frame_replay(FrameReplay r, const void* __restrict__ x, int D, int C,
                                                      const int* __restrict__ cols, const int* __restrict__ spos,
                                                      const float* __restrict__ den) {
  __shared__ float xs[kCFixSolo];
  frame_replay_wave<DT>(r, int(blockIdx.x), int(gridDim.x), xs, x, D, C, cols, spos, den, int(threadIdx.x));
}

// parts[NP][stride]: the 16-frame group sums of one rank, or the all-gathered ones of every rank (frame order).
// rpr > 0: every rank's block of rpr rows is [rpr / 2 group sums | rpr / 2 group bounds of sum |x^|] (k_frame_centres);
// rpr = 0: sums only (then |x^| <= 1 bounds the replay margin: sound, but flags far more columns).
// Workgroup (bx, y) = one wave; lane = column bx*64 + lane.  Every y computes the same means and the same set of
// boundary-near columns; y = 0 stores the means.  replay_rows = 1 (single rank: x / den hold ALL R rows): the ~R/256
// level-1 groups of a flagged column are spread over the gridDim.y waves of its column block -- one CU ingests a
// strided column at ~50 GB/s, 25088 rows would take it ~35 us -- which leave them in l1g[column][group]; the last
// wave to arrive (agent-scope release / acquire around the block's ticket) finishes the cascade and stores the
// replayed mean.  replay_rows = 0 (frame-sharded pass): the flagged columns are only counted (fragile_count).
__host__ __device__ inline int64_t cascade_l1_groups(int64_t n) {
  const int lp = cascade_lp(n);
  return (((n >> lp) + (int64_t(1) << lp) - 1) >> lp);
}
__host__ inline size_t vc_lds_bytes(int64_t R, int64_t N) {      // k_video_centre: main waves hold a column's level-1 groups,
  const int64_t main_f = std::min<int64_t>(kL1Cap, std::max(cascade_l1_groups(R), cascade_l1_groups(R >> 2)));   // riders a frame's
  const int64_t rider_f = N <= kCFixSolo ? N : std::max(cascade_l1_groups(N), cascade_l1_groups(N >> 2));           // values / groups
  return size_t(std::max<int64_t>(std::max(main_f, rider_f), 64) + 4) * 4;
}
template <int DT>
__global__ __launch_bounds__(64) void k_video_centre(const double* __restrict__ parts, int NP, int64_t stride, int C,
                                                      int64_t R, float* __restrict__ vc, const void* __restrict__ x,
                                                      int D, const int* __restrict__ cols,
                                                      const int* __restrict__ spos, const float* __restrict__ den,
                                                      int strict, int replay_rows, int* __restrict__ fragile_count,
                                                      float* __restrict__ l1g, int vstride, int* __restrict__ vtick,
                                                      uint8_t* __restrict__ vflag, int rpr, double kk,
                                                      FrameStatSrc fs, const float* __restrict__ dmin, int F,
                                                      FrameReplay frp = FrameReplay{}, int Ymain = 0,
                                                      const int* __restrict__ vcorr_count = nullptr,
                                                      const NormCorr* __restrict__ vcorr = nullptr) {
  // level-1 groups of one column / a rider's frame chain: vc_lds_bytes(R, N) of dynamic LDS.  (Round 5: this was a static
  // float[kL1Cap + 4] -- 32 KiB for every 64-thread workgroup, i.e. FIVE waves per CU: with 700 main + 1024 rider waves
  // in the launch the riders only started when main waves ended, and fp16's ~4000 replays queued for 30 us.)
  extern __shared__ __attribute__((aligned(16))) float l1s[];
  // grid rows [Ymain, gridDim.y) (Ymain > 0) are RIDER waves: they replay the boundary-near frame means k_frame_centres
  // listed -- work that is independent of the video centre and hides under this kernel's own latency chain
  const int lane = threadIdx.x, bx = blockIdx.x, y = blockIdx.y, Y = Ymain > 0 ? Ymain : int(gridDim.y);
  if (y >= Y) {
#ifdef VC2_DEBUG_TIMING
    const int rid_ = (y - Y) * int(gridDim.x) + bx;
    if (lane == 0 && rid_ < 4096) g_dbg_wg[6][0][rid_] = wall_clock64();
#endif
    frame_replay_wave<DT>(frp, (y - Y) * int(gridDim.x) + bx, (int(gridDim.y) - Y) * int(gridDim.x), l1s, x, D, C, cols, spos,
                          den, lane);
#ifdef VC2_DEBUG_TIMING
    if (lane == 0 && rid_ < 4096) g_dbg_wg[6][1][rid_] = wall_clock64();
#endif
    return;
  }
#ifdef VC2_DEBUG_TIMING
  struct EndStamp { int i; __device__ ~EndStamp() { if (threadIdx.x == 0 && i < 4096) g_dbg_wg[7][1][i] = wall_clock64(); } } end_stamp_{bx * Y + y};
  if (lane == 0 && bx * Y + y < 4096) g_dbg_wg[7][0][bx * Y + y] = wall_clock64();
#endif
  const bool replay = strict != 0 && DT != VC2_F32;
  const bool all = strict == 2;
  const int c = bx * 64 + lane;
  if (bx == 0 && y == 0 && lane == 0) VC2_STAMP(600);
  bool flag = false, pre = false;
  const int cl = c < C ? c : C - 1;
  const int my_col = cols ? cols[cl] : cl, my_sp = spos ? spos[cl] : cl;   // (in flight with the partial sums)
  const int nvc = vcorr_count ? *vcorr_count : 0;                          // (likewise: not a round trip of its own)
  double ab = 0.0;
  float q = 0.f;
  if (c < C) {
    double t = 0.0;
    const int half = rpr > 0 ? rpr / 2 : NP;        // rows of sums in a rank's block
    const int nblocks = rpr > 0 ? NP / rpr : 1;
    for (int b = 0; b < nblocks; ++b) {
      const double* __restrict__ blk = parts + int64_t(b) * (rpr > 0 ? rpr : 0) * stride;
      for (int p0 = 0; p0 < half; p0 += 8) {        // eight loads in flight, added in part order
        double v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = blk[int64_t(min(p0 + u, half - 1)) * stride + c];
          w[u] = rpr > 0 ? blk[int64_t(half + min(p0 + u, half - 1)) * stride + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (p0 + u < half) { t += v[u]; ab += w[u]; }
      }
    }
    // fused centre launch: the group sums were formed before the norm fix-ups -> their corrections are added here
    for (int e0 = 0; e0 < nvc; e0 += 8) {                         // (bf16: normally none; fp16: ~16 at the cfg5 shape -- eight
      NormCorr nc8[8];                                            //  entries' loads in flight: one by one they cost 0.5 us each)
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) nc8[u] = vcorr[min(e0 + u, nvc - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = ldT<DT>(x, int64_t(nc8[u].row) * D + my_col);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (e0 + u < nvc) {
          const float xo = rnT<DT>(div_via_f64(v8[u], 1.0 / double(nc8[u].den_old)));
          const float xn = rnT<DT>(div_via_f64(v8[u], 1.0 / double(nc8[u].den_new)));
          t += double(xn) - double(xo);
        }
      }
    }
    if (y == 0) vc[c] = mean_T<DT>(t, R);
    q = float(t) / float(R);
    if (replay && cascade_modelled(R)) {
      if (all) flag = true;
      else if (strict != 3) flag = mean_near_T_boundary<DT>(q);                    // default: the empirical margin
      else if (rpr > 0) flag = T_boundary_within<DT>(q, mean_delta<DT>(q, ab < double(R) ? ab : double(R), R, kk));
      else pre = T_boundary_within<DT>(q, mean_delta<DT>(q, double(R), R, kk));      // |x^| <= 1: A <= R
    }
  }
  // single rank: the real A of the (few) columns that pass with A = R -- the sum over the frames of the frame bounds
  // (k_frame_centres), from sweep 1's partials and the frames' smallest denominators; the lanes share the frames.
  // (Every y evaluates the same expression in the same order: the same flags in all workgroups of a column block.)
  if (replay) {
    const uint64_t pm = __ballot(pre);
    if (pm && fs.part && dmin) {
      for (uint64_t m = pm; m; m &= m - 1) {
        const int vl = __builtin_ctzll(m);
        const int col = __shfl(my_col, vl, 64);
        double a = 0.0;
        for (int f = lane; f < F; f += 64) {
          const double b = abs_sum_bound(fs.sumsq<DT>(f, col, x, D), fs.N, dmin[f]);
          a += b < double(fs.N) ? b : double(fs.N);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == vl) ab = a;
      }
      if (pre) flag = T_boundary_within<DT>(q, mean_delta<DT>(q, ab < double(R) ? ab : double(R), R, kk));
    } else {
      flag = flag || pre;
    }
  }
  if (bx == 0 && y == 0 && lane == 0) VC2_STAMP(605);
#ifdef VC2_DEBUG_TIMING
  { const unsigned long long dbg_nf = (unsigned long long)__popcll(__ballot(flag));
    if (lane == 0 && bx * Y + y < 4096) { g_dbg_vc[0][bx * Y + y] = wall_clock64(); g_dbg_vc[1][bx * Y + y] = 0ull; g_dbg_vc[2][bx * Y + y] = dbg_nf | ((unsigned long long)nvc << 32); } }
#endif
  if (vflag && y == 0 && c < C) vflag[c] = flag ? 1 : 0;         // (frame-sharded pass: which columns to replay)
  if (!replay) return;
  const uint64_t flagged = __ballot(flag);
  if (!flagged) return;
  if (y == 0 && lane == 0 && fragile_count) atomicAdd(fragile_count, __popcll(flagged));
  if (!replay_rows) return;
  const int group = C >= 8 ? 32 : 4;
  const int simple_end = (C / group) * group;
  const int lp = cascade_lp(R), B = 1 << lp, gpw = 64 >> lp;       // gpw level-1 groups per wave, B lanes each
  const int64_t nbv = R >> lp;
  const int G1v = int((nbv + B - 1) >> lp);
  const int sub = lane >> lp, li = lane & (B - 1);
  uint64_t plain = 0ull;                            // flagged columns that take the plain cascade
  for (uint64_t m = flagged; m; m &= m - 1) {
    const int vl = __builtin_ctzll(m), cc = bx * 64 + vl;
    const int col = __shfl(my_col, vl, 64), sp = __shfl(my_sp, vl, 64);
    if (sp >= simple_end) {                       // row_sum's four interleaved chains (C % 32 tail): y = 0 alone
      if (y == 0) {                                 // (cascade_modelled(R) covers the chains of R / 4 too)
        const float s = wave_column_solo<DT>(l1s, false, x, D, col, den, 0, R, lane);
        if (lane == 0) vc[cc] = rnT<DT>(s / float(R));
      }
      continue;
    }
    plain |= 1ull << vl;
  }
  // up to four flagged columns of the block per round of loads (fp16 flags 1 .. 4 columns per block, each a round trip
  // of 16 strided rows per lane: one after the other they took 7-8 us each and set this launch's time)
  auto batch = [&](auto nc_tag, uint64_t& m) {
    constexpr int NC = decltype(nc_tag)::value;
    int col[NC], cc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int vl = __builtin_ctzll(m);
      m &= m - 1;
      cc[j] = bx * 64 + vl;
      col[j] = __shfl(my_col, vl, 64);
    }
    for (int g4 = y * gpw; g4 < G1v; g4 += Y * gpw) {   // gpw level-1 groups at once (four at lp = 4)
      const int g1 = g4 + sub;
      const int nbl = g1 < G1v ? int(min<int64_t>(B, nbv - (int64_t(g1) << lp))) : 0;
      float a[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) a[j] = 0.f;
      if (li < nbl) block_sum_rows_n<DT, NC>(x, D, col, den, ((int64_t(g1) << lp) + li) << lp, lp, a);
      const int l0 = lane & ~(B - 1);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float sgrp = __shfl(a[j], l0, 64);           // the block sums of my group, added in block order
        for (int u = 1; u < B; ++u) { const float t = __shfl(a[j], l0 + u, 64); if (u < nbl) sgrp += t; }
        if (li == 0 && g1 < G1v) l1g[int64_t(cc[j]) * vstride + g1] = sgrp;
      }
    }
  };
  for (uint64_t m = plain; m;) {
    const int left = __popcll(m);
    if (left >= 4) batch(std::integral_constant<int, 4>{}, m);
    else if (left == 3) batch(std::integral_constant<int, 3>{}, m);
    else if (left == 2) batch(std::integral_constant<int, 2>{}, m);
    else batch(std::integral_constant<int, 1>{}, m);
  }
  // Release / acquire FENCES around the ticket.  (A fence-free variant -- the groups as relaxed agent-scope atomic
  // stores, s_waitcnt vmcnt(0), a relaxed ticket, coherent loads in the last arriver -- was 0.8 us faster and WRONG:
  // the soak caught a stale group in 1 of 972 cases, 5 of 40 repeats of that case.  vmcnt(0) does not mean the
  // write-through has reached the point the other XCDs read from.)
#ifdef VC2_DEBUG_TIMING
  if (lane == 0 && bx * Y + y < 4096) g_dbg_vc[1][bx * Y + y] = wall_clock64();
#endif
  __threadfence();                                   // release my groups ...
  int last = 0;
  if (lane == 0) last = atomicAdd(&vtick[bx], 1) == Y - 1;
  if (!__shfl(last, 0, 64)) return;
  __threadfence();                                   // ... acquire everyone else's
  for (uint64_t m = flagged; m; m &= m - 1) {
    const int vl = __builtin_ctzll(m), cc = bx * 64 + vl;
    const int col = __shfl(my_col, vl, 64), sp = __shfl(my_sp, vl, 64);
    if (sp >= simple_end) continue;
    for (int g1 = lane; g1 < G1v; g1 += 64) l1s[g1] = l1g[int64_t(cc) * vstride + g1];
    wave_lds_fence();
    const float s = wave_cascade_final<DT>(l1s, nbv, x, D, col, den, 0, 1, R, lane, lp);
    wave_lds_fence();
    if (lane == 0) vc[cc] = rnT<DT>(s / float(R));
  }
}

// ---- frame-sharded pass: the video-centre replay across ranks ------------------------------------------------
// Every rank flags the same columns (the flags come from the all-gathered group sums).  Slot j = the j-th flagged
// column in ascending order.  The cascade's level-0 blocks are B = 2^lp consecutive rows OF THE VIDEO (lp =
// cascade_lp(R_total): 16 rows up to 2^19 tokens, 32 up to 2^23, 64 beyond), so a rank whose row count is not a
// multiple of B holds the tail of a block that started in the previous rank and the head of one that ends in the next.
// k_vc_blocks writes, per slot, a record of kVcRec(R_local) floats:
//     [0, 64)   the x^ values of my first h rows (the end of a block begun by the previous rank), in row order
//     [64, 128) the x^ values of my last t rows (the beginning of a block the next rank ends)
//     [128, ..) the sums of my complete blocks (rows added in order)
// The records are all-gathered (rank order = row order) and k_vc_finish runs the rest of torch's cascade over the
// whole video: a straddling block is its rows' raw values added one by one in row order -- from as many ranks as it
// meets (a rank with fewer rows than a block is all head); the video's last R_total % B rows (the cascade's tail) come
// the same way.  Needs equal row counts per rank and the cascade's plain form (column in a full group of 32).  More
// flagged columns than a record buffer holds go in further rounds (slot offset j0).
constexpr int kVcEdge = 64;
__host__ __device__ inline int64_t vc_rec_floats(int64_t R_local) { return 2 * kVcEdge + R_local / 16 + 1; }
__host__ __device__ inline int vc_head_rows(int64_t row0, int B) { return int((B - row0 % B) % B); }
__device__ __forceinline__ int nth_flagged_column(const uint8_t* __restrict__ vflag, int C, int j, int lane) {
  // lane-contiguous chunks, wave scan; returns the column of the j-th set flag or -1 (same value in every lane)
  const int E = (C + 63) / 64;
  const int b = lane * E, e = min(C, b + E);
  uint32_t cnt = 0;
  for (int c = b; c < e; ++c) cnt += vflag[c] ? 1u : 0u;
  const uint32_t incl = wave_incl_scan_u32(cnt);
  const uint32_t before = incl - cnt;
  int found = -1;
  if (uint32_t(j) >= before && uint32_t(j) < incl) {
    uint32_t k = before;
    for (int c = b; c < e; ++c) if (vflag[c]) { if (k == uint32_t(j)) { found = c; break; } ++k; }
  }
  const uint64_t m = __ballot(found >= 0);
  return m ? __shfl(found, __builtin_ctzll(m), 64) : -1;
}

template <int DT>
__global__ __launch_bounds__(64) void k_vc_blocks(const uint8_t* __restrict__ vflag, int C, const void* __restrict__ x,
                                                  int D, const int* __restrict__ cols, const int* __restrict__ spos,
                                                  const float* __restrict__ den, int64_t R_local, int64_t row0, int lp,
                                                  float* __restrict__ blocks_out, int j0) {
  const int lane = threadIdx.x, j = blockIdx.y;                  // record j of this round = flagged column number j0 + j
  const int cc = nth_flagged_column(vflag, C, j0 + j, lane);
  if (cc < 0) return;
  const int col = cols ? cols[cc] : cc;
  const int B = 1 << lp;
  const int h = int(min<int64_t>(R_local, vc_head_rows(row0, B)));
  const int64_t nb = (R_local - h) >> lp;
  const int t = int(R_local - h - (nb << lp));
  float* rec = blocks_out + int64_t(j) * vc_rec_floats(R_local);
  if (blockIdx.x == gridDim.x - 1) {                            // the raw edges
    if (lane < h) rec[lane] = xhat_at<DT>(x, lane, D, col, den);
    if (lane < t) rec[kVcEdge + lane] = xhat_at<DT>(x, h + (nb << lp) + lane, D, col, den);
    return;
  }
  const int64_t b = int64_t(blockIdx.x) * 64 + lane;
  if (b >= nb) return;
  rec[2 * kVcEdge + b] = block_sum_rows<DT>(x, D, col, den, h, 1, b << lp, lp);
}

template <int DT>
__global__ __launch_bounds__(256) void k_vc_finish(const uint8_t* __restrict__ vflag, int C,
                                                   const int* __restrict__ spos, const float* __restrict__ blocks_all,
                                                   int world, int cap, int64_t R_local, int64_t R_total,
                                                   float* __restrict__ vc, int* __restrict__ fragile_count, int j0) {
  __shared__ float l1[kL1Cap + 4];
  __shared__ float tailv[kVcEdge];
  const int tid = threadIdx.x, lane = tid & 63, j = blockIdx.x;  // record j of this round = flagged column number j0 + j
  const int cc = nth_flagged_column(vflag, C, j0 + j, lane);
  if (cc < 0) return;
  const int group = C >= 8 ? 32 : 4;
  const int sp = spos ? spos[cc] : cc;
  if (sp >= (C / group) * group) return;                        // row_sum's interleaved chains: not replayed here
  const int lp = cascade_lp(R_total), B = 1 << lp;
  const int64_t rec = vc_rec_floats(R_local);
  auto record = [&](int64_t w) { return blocks_all + (w * cap + j) * rec; };
  // what rank w holds: h head rows (raw, the end of a block begun before it), nb complete blocks, t tail rows (raw)
  auto head_of = [&](int64_t w) { return int(min<int64_t>(R_local, vc_head_rows(w * R_local, B))); };
  // the x^ value of video row r: raw in its rank's head or tail section (only asked for rows of straddling blocks)
  auto row_value = [&](int64_t r) -> float {
    const int64_t w = r / R_local, off = r - w * R_local;
    const int h = head_of(w);
    if (off < h) return record(w)[off];
    const int64_t nb = (R_local - h) >> lp;
    return record(w)[kVcEdge + (off - h - (nb << lp))];
  };
  // the level-0 sum of the video's block b: a rank's own complete block, or -- a block that meets two OR MORE ranks
  // (ranks with fewer rows than a block included) -- its rows one by one in row order
  auto block_value = [&](int64_t b) -> float {
    const int64_t g0 = b << lp, w = g0 / R_local, off = g0 - w * R_local;
    const int h = head_of(w);
    const int64_t nb = (R_local - h) >> lp;
    if (off >= h && ((off - h) & (B - 1)) == 0 && ((off - h) >> lp) < nb) return record(w)[2 * kVcEdge + ((off - h) >> lp)];
    float a = row_value(g0);
    for (int u = 1; u < B; ++u) a += row_value(g0 + u);
    return a;
  };
  const int64_t nbv = R_total >> lp;
  const int G1 = int((nbv + B - 1) >> lp);
  for (int g = tid; g < G1; g += 256) {                         // level 1: B block sums in block order
    const int64_t b0 = int64_t(g) << lp;
    const int nbl = int(min<int64_t>(B, nbv - b0));
    float a = 0.f;
    for (int u = 0; u < nbl; ++u) {
      const float t = block_value(b0 + u);
      a = u == 0 ? t : a + t;
    }
    l1[g] = a;
  }
  const int ntail = int(R_total - (nbv << lp));                  // the cascade's tail: the video's last R_total % B rows
  if (tid < ntail) tailv[tid] = row_value((nbv << lp) + tid);
  __syncthreads();
  if (tid < 64) {
    const float s = wave_cascade_final<DT>(l1, nbv, nullptr, 0, 0, nullptr, 0, 1, R_total, lane, lp, tailv);
    if (lane == 0) {
      vc[cc] = rnT<DT>(s / float(R_total));
      if (fragile_count) atomicSub(fragile_count, 1);            // one flagged column less that kept its exact mean
    }
  }
}

// 5-scale Gaussian sum of one squared distance (vidcom2.py:62): every op rounded to T, exp = fp64 exp rounded once
template <int DT>
__device__ __forceinline__ float gauss_term(float dist, int a) {
  const float two_a = a == 0 ? 0.25f : a == 1 ? 0.5f : a == 2 ? 1.0f : a == 3 ? 2.0f : 4.0f;   // 2*alpha, alpha = 2^-3 .. 2^1
  const float arg = rnT<DT>((-dist) / two_a);
  return rnT<DT>(float(exp(double(arg))));
}
// The same term through v_exp_f32 (half precision only): e = 2^(arg * log2 e) carries < 30 fp32-ulps of error for
// |arg| <= 16 (1 ulp of v_exp_f32; the product's rounding and the constant's error scale with |arg * log2 e| <= 23.1:
// 23.1 * 1.5 * 2^-24 * ln 2 relative), so RN_T(e) is the correctly rounded result unless e lies within
// kFragileUlpsExp of a T rounding boundary -- then (false) the caller takes the fp64 path.  dist <= 4 for unit
// vectors, i.e. arg >= -16 and e >= 1.1e-7: a normal fp32 number (v_exp_f32 flushes subnormals).
constexpr int kFragileUlpsExp = 64;
template <int DT>
__device__ __forceinline__ bool gauss_term_fast(float dist, int a, float& out) {
  if constexpr (DT == VC2_F32) {
    out = gauss_term<DT>(dist, a);
    return true;
  } else {
    const float two_a = a == 0 ? 0.25f : a == 1 ? 0.5f : a == 2 ? 1.0f : a == 3 ? 2.0f : 4.0f;
    const float arg = rnT<DT>((-dist) / two_a);
    const float e = __builtin_amdgcn_exp2f(arg * 1.44269504088896341f);
    out = rnT<DT>(e);
    return fabsf(arg) <= 16.f && !near_T_boundary<DT>(e, kFragileUlpsExp);     // (NaN: not decided here)
  }
}
template <int DT>
__device__ __forceinline__ float gauss_sum(float dist) {
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float e = gauss_term<DT>(dist, a);
    acc = (a == 0) ? e : rnT<DT>(acc + e);                   // Python sum(): 0 + t1 is exact
  }
  return acc;
}

// ---- sweep 3 arithmetic ---------------------------------------------------------------------------------------
// The sweep is bound by VALU ISSUE, not by memory (measured, scripts/ubench/valu_rates.hip: a wave64 shift, cvt_pk,
// packed or DPP op costs ~4.3 cycles per SIMD, and / sub / mul ~2.7), so the per-element sequence is what is optimised:
//   bf16  x^ unpacked once (and + shift); per centre: two subtractions, ONE v_cvt_pk_bf16_f32 rounding both
//         differences, unpack, two exact products, ONE cvt_pk rounding both squares, and -- "torch order" mode -- ONE
//         v_dot2c_f32_bf16 against (1, 1) adding both squares to the fp32 accumulator (keeps subnormals, rounds the
//         three-term sum once: scripts/ubench/valu_rates.hip) -- 20 instructions per two channels and two centres;
//   fp16  the hardware's packed fp16 arithmetic IS the reference's arithmetic here: RN_16(RN_32(a - b)) == RN_16(a - b)
//         for fp16 a, b (the fp32 difference is inexact only when the smaller operand is below 2^-13 of the larger,
//         where both roundings return the larger), and a product of two fp16 numbers is exact in fp32; v_pk_add_f16 /
//         v_pk_mul_f16 keep subnormals (IEEE, measured) -- v_pk_add_f16, v_pk_mul_f16, v_dot2c_f32_f16: 3 instructions
//         per two channels and centre, with the centres as packed fp16 pairs;
//   fp32 / "exact" mode: the plain sequence, fp64 accumulators.
template <int DT, int ACC> struct DistArith;
template <> struct DistArith<VC2_BF16, 1> {
  static __device__ __forceinline__ uint32_t pk(float a, float b) {                // (RN_bf16(a) | RN_bf16(b) << 16)
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    union { b2_t h; uint32_t u; } c;
    c.h = __builtin_convertvector((f2_t){a, b}, b2_t);
    return c.u;
  }
  static __device__ __forceinline__ float dot_ones(uint32_t q, float acc) {        // acc + q.lo + q.hi
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    union { b2_t h; uint32_t u; } a, one;
    a.u = q; one.u = 0x3F803F80u;
    return __builtin_amdgcn_fdot2_f32_bf16(a.h, one.h, acc, false);
  }
};

// K fp32 partial sums per lane (K = 4, 8 or 16) -> their K wave totals with a FIXED tree whose first two levels fold
// two values per instruction pair: v_permlane32_swap / v_permlane16_swap exchange halves / odd-even rows of two
// registers, one add folds both; four DPP steps then finish the four 16-lane rows of every register at once.
// On return register m (m < K/4) holds, in ALL lanes of row r (lanes 16r .. 16r+15), the total of value
// 4m + {0, 2, 1, 3}[r].  (~2.5 instructions per value instead of 8.)
template <int K>
__device__ __forceinline__ void wave_totals_f32(float (&v)[K]) {
  static_assert(K == 4 || K == 8 || K == 16, "K");
#pragma unroll
  for (int i = 0; i < K / 2; ++i) {                              // lanes l and l + 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);        // lanes 0-31: value 2i, lanes 32-63: value 2i + 1
  }
#pragma unroll
  for (int i = 0; i < K / 4; ++i) {                              // rows r and r + 1 (r even)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < K / 4; ++i) {
    float t = v[i];
    t = dpp_addf<0xB1, 0xF>(t);     // quad_perm [1,0,3,2]
    t = dpp_addf<0x4E, 0xF>(t);     // quad_perm [2,3,0,1]
    t = dpp_addf<0x141, 0xF>(t);    // row_half_mirror
    t = dpp_addf<0x140, 0xF>(t);    // row_mirror
    v[i] = t;
  }
}

// sweep 3 (vidcom2.py:61-62, :32-33): per token dist_v = RN_T(sum_c RN_T(RN_T(x^ - vc)^2)), dist_f likewise with
// the frame centre, x^ recomputed from X and den; then -- still inside the workgroup -- the 5-scale Gaussian sums
// v, f, total = RN_T(v + f) and the workgroup's partial sum of v (for the per-frame uniqueness score).
// Column offsets and both centres of the lane's compact positions live in registers for the whole workgroup; the
// row loop touches LDS only for the row itself.  Three phases per workgroup (one frame split, <= 64 rows):
//   1. row loop, one wave per row.  The sweep is bound by VALU ISSUE (see DistPair), so the lane's elements are
//      handled two by two; the totals of TWO rows (4 values) are reduced together (wave_totals_f32).  In "torch
//      order" mode a sum within the replay margin of a T rounding boundary is put on a workgroup-local list;
//   2. the (rare) listed sums are replayed in torch's cascade order by the whole workgroup: every thread recomputes
//      a few squares from X and scatters them to their SORTED positions (spos) in LDS, wave 0 adds them;
//   3. 10 Gaussian terms per token spread over the workgroup's threads (v_exp_f32, fp64 next to a rounding
//      boundary), the two running sums, outputs.
// (Round 3 also built the other data flow -- sweep 2 materialising x^ for a sweep 3 that streams it: same bytes per
// pass, measured 3 % SLOWER at the target shape, 90 MB more workspace; NOTES_r03.md.)
constexpr int kDistMaxRows = 64;
// first row of split j (of S) of an N-row frame: rows_j ~ N/S * (1 + a (1 - 2 j / (S - 1))), a = skew_q10 / 1024
__host__ __device__ inline int dist_split_cut(int j, int S, int N, int skew_q10) {
  if (j <= 0) return 0;
  if (j >= S) return N;
  const long long num = (long long)N * j * (1024 + skew_q10) * (S - 1) - (long long)N * skew_q10 * j * (j - 1);
  const long long den = 1024LL * S * (S > 1 ? S - 1 : 1);
  return int((num + den / 2) / den);
}

// V2 = 1 ("streamlined", round 5; 16-bit rows of full 1 KiB chunks, C = 64 NPLB = D / 2, ACC = 1, QUAD tables): the row
// loop of k_norm_colsum2 -- the row's elements are read into registers first (bf16 with ds_read_u16_d16_hi: the element
// arrives as its fp32 value), the wave's NEXT row is DMA'd into the same buffer at once and lands while this one is
// computed; LDS addresses are per-lane constants; the rows' denominators / exact-division flags are parked in lanes (no LDS
// access in the loop: the compiler would put a vmcnt(0) in front of it); the first row's DMA is issued before the tables
// are fetched.  fp16 divides by the fused-multiply-add quotient proved in tests/tools/check_f16_quotient.c.
#ifndef VC2_S3_WAVES
#define VC2_S3_WAVES 3      // (at 4 the streamlined loop spills inside the row loop)
#endif
template <int DT, int VEC, int NPLB, int ACC, int QUAD = 0, int V2 = 0>
__global__ __launch_bounds__(kRowWaves * 64, V2 ? VC2_S3_WAVES : 1) void k_dist(const void* __restrict__ x, int N, int D, int CV, int C,
                                                         const int* __restrict__ cols,
                                                         const int* __restrict__ spos, int strict, int S,
                                                         int rows_per_split, int skew_q10, const float* __restrict__ den,
                                                         const uint8_t* __restrict__ rflag,
                                                         const float* __restrict__ vc,
                                                         const float* __restrict__ fc,
                                                         void* __restrict__ v_T, void* __restrict__ f_T,
                                                         float* __restrict__ total, double* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = Tr<DT>::ES;
  constexpr bool kFast = ACC == 1 && DT == VC2_BF16;
  const size_t rowb = row_lds_bytes(D, ES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup b = split b / F of frame b % F, and the splits of a frame SHRINK with their number (skew_q10 / 1024 =
  // relative excess of the first over the mean; dist_split_cut): the hardware issues from the oldest wave first, so of
  // the four workgroups that share a CU for the whole sweep the first-dispatched one runs its rows 30 % faster than
  // the last (in-kernel stamps, scripts/dbg_wg.py: row loops of 25 rows ended at 21.8 / 23.5 / 25.6 / 28.6 us by
  // dispatch quartile) -- with equal splits the sweep waited for the youngest workgroups, and for their replays.
  const int F_ = int(gridDim.x) / S;
  const int sp = int(blockIdx.x) / F_, f = int(blockIdx.x) - sp * F_;
  const int n0 = dist_split_cut(sp, S, N, skew_q10), n1 = dist_split_cut(sp + 1, S, N, skew_q10);
  const int nrows = n1 - n0;
  (void)rows_per_split;
  // One row buffer per wave and no intra-wave prefetch: measured faster than double buffering here
  // (45 vs 50 us at 128x196x3584) because the smaller LDS footprint doubles the resident waves.
  unsigned char* buf0 = smem + size_t(wave) * rowb;               // [kRowWaves][rowb]; phase 2: float sq[C]
  const size_t area = (std::max(size_t(kRowWaves) * rowb, size_t(C) * 4 + 16) + 15) / 16 * 16;
  float* dens = reinterpret_cast<float*>(smem + area);             // [kDistMaxRows]
  float* dists = dens + kDistMaxRows;                             // [kDistMaxRows][2]  RN_T distances (v, f)
  float* ebuf = dists + 2 * kDistMaxRows;                         // [kDistMaxRows][10] Gaussian terms
  int* list = reinterpret_cast<int*>(ebuf + 10 * kDistMaxRows);   // [2 * kDistMaxRows] (local row) * 2 + centre
  int* lcount = list + 2 * kDistMaxRows;                          // [0] phase-2 list, [1] phase-3 list
  uint16_t* elist = reinterpret_cast<uint16_t*>(lcount + 4);      // [10 * kDistMaxRows] exp terms for the fp64 path
  uint8_t* rfl = reinterpret_cast<uint8_t*>(elist + 10 * kDistMaxRows);   // [kDistMaxRows]
  int n = n0 + wave;
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) VC2_STAMP(700);
  VC2_WGTIME(2, 0);
  static_assert(V2 == 0 || (QUAD != 0 && ACC == 1 && VEC > 1 && DT != VC2_F32 && NPLB % 4 == 0), "streamlined sweep 3");
  constexpr int NCH2 = NPLB / 4;                                    // (V2) 1 KiB chunks per row
  const unsigned char* xl2 = static_cast<const unsigned char*>(x) + (size_t(f) * N + n) * (size_t(D) * ES) + size_t(lane) * 16;   // (V2) the wave's first row, this lane's 16 bytes
  const int cnt2 = wave < nrows ? (nrows - wave + kRowWaves - 1) / kRowWaves : 0;                    // (V2) the wave's rows
  float my_dn = 1.f;                                                // (V2) lane j: the wave's row j
  uint32_t my_rf = 1u;
  if constexpr (V2 != 0) {
    if (cnt2 > 0) s2_issue_row<NCH2>(xl2, buf0);
    if (lane < cnt2) {
      const int64_t row = int64_t(f) * N + n + kRowWaves * lane;
      my_dn = den[row];
      my_rf = rflag ? uint32_t(rflag[row]) : 1u;
    }
  }
  if constexpr (V2 == 0) {
  if (lane < 4) reinterpret_cast<uint32_t*>(buf0 + rowb - 16)[lane] = 0u;
  if (tid < 2) lcount[tid] = 0;
  for (int r = tid; r < nrows; r += kRowWaves * 64) {
    dens[r] = den[int64_t(f) * N + n0 + r];
    rfl[r] = ((kFast || (ACC == 1 && DT == VC2_F16)) && rflag) ? rflag[int64_t(f) * N + n0 + r] : 1;   // (fp16: written by k_norm_colsum2 only -- else null)
  }
  }
  float den_mine = 0.f;                                             // (V2) dens[tid], stored to LDS behind the table loads
  if constexpr (V2 != 0) { if (tid < nrows) den_mine = den[int64_t(f) * N + n0 + tid]; }
  int coff[NPLB];
  float cv[NPLB], cf[NPLB];                                       // video / frame centre of the lane's compact positions
  // QUAD (ACC = 1 only: any ownership of the compact positions gives the same bits there -- the accumulation is bounded,
  // not ordered; the host picks it when cols is given and C % 4 == 0): a lane owns QUADS of adjacent positions,
  // p(i) = 256 (i / 4) + 4 lane + i % 4, so that the three tables come in 16-byte loads: 3 NPLB / 4 instead of 3 NPLB
  // load instructions per lane and 119 instead of 127 VGPRs; k_dist 36.5 -> 35.2 us.  (As a run-time branch next to the
  // other form the kernel needed 145 VGPRs -- a wave per SIMD less.)
  static_assert(QUAD == 0 || (ACC == 1 && NPLB % 4 == 0), "QUAD needs the bounded accumulation");
  if constexpr (QUAD != 0) {
    const int pad = int((rowb - 16) / ES);
    const float* __restrict__ fcf = fc + int64_t(f) * C;
    {                                                             // (table by table: the temporaries must not become the register peak)
      int4 tc[NPLB / 4];
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) { const int p = 256 * q + 4 * lane; tc[q] = *reinterpret_cast<const int4*>(cols + (p < C ? p : C - 4)); }
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) {
        const bool in = 256 * q + 4 * lane < C;
        coff[4 * q] = in ? tc[q].x : pad; coff[4 * q + 1] = in ? tc[q].y : pad; coff[4 * q + 2] = in ? tc[q].z : pad; coff[4 * q + 3] = in ? tc[q].w : pad;
      }
    }
    asm volatile("" ::: "memory");
    {
      float4 ta[NPLB / 4];
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) { const int p = 256 * q + 4 * lane; ta[q] = *reinterpret_cast<const float4*>(vc + (p < C ? p : C - 4)); }
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) {
        const bool in = 256 * q + 4 * lane < C;
        cv[4 * q] = in ? ta[q].x : 0.f; cv[4 * q + 1] = in ? ta[q].y : 0.f; cv[4 * q + 2] = in ? ta[q].z : 0.f; cv[4 * q + 3] = in ? ta[q].w : 0.f;
      }
    }
    asm volatile("" ::: "memory");
    {
      float4 tb[NPLB / 4];
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) { const int p = 256 * q + 4 * lane; tb[q] = *reinterpret_cast<const float4*>(fcf + (p < C ? p : C - 4)); }
#pragma unroll
      for (int q = 0; q < NPLB / 4; ++q) {
        const bool in = 256 * q + 4 * lane < C;
        cf[4 * q] = in ? tb[q].x : 0.f; cf[4 * q + 1] = in ? tb[q].y : 0.f; cf[4 * q + 2] = in ? tb[q].z : 0.f; cf[4 * q + 3] = in ? tb[q].w : 0.f;
      }
    }
  } else {
  load_col_offsets<NPLB>(cols, C, int((rowb - 16) / ES), lane, coff);
  {
    // unconditional loads (clamped index), batched; a load inside a conditional is waited for on the spot, and
    // 2 * NPLB serialised L2 round trips cost the whole workgroup ~10 us.  (Batches of 8: the temporaries must
    // not become the kernel's register peak.)
    constexpr int B = NPLB % 8 == 0 ? 8 : NPLB % 7 == 0 ? 7 : 2;
    const float* __restrict__ fcf = fc + int64_t(f) * C;
#pragma unroll
    for (int i0 = 0; i0 < NPLB; i0 += B) {
      float a[B], b[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const uint32_t p = uint32_t((i0 + j) * 64 + lane), pc = p < uint32_t(C) ? p : uint32_t(C - 1);
        a[j] = vc[pc];                                             // (32-bit offsets: scalar base + VGPR offset loads)
        b[j] = fcf[pc];
      }
      asm volatile("" ::: "memory");                               // keep the batches apart
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const bool in = (i0 + j) * 64 + lane < C;
        cv[i0 + j] = in ? a[j] : 0.f;
        cf[i0 + j] = in ? b[j] : 0.f;
      }
    }
  }
  }
  // (plain loads first: behind an in-flight global_load_lds the compiler waits for EVERY load separately)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (V2 != 0) {
    if (tid < 2) lcount[tid] = 0;
    if (tid < nrows) dens[tid] = den_mine;                          // (phase 2 reads them; nrows <= kDistMaxRows <= 256)
  } else {
  if (n < n1) row_issue<DT, VEC, VC2_AUX_S3>(x, int64_t(f) * N + n, D, CV, buf0, lane);
  }
  __syncthreads();
  // ---- phase 1 ---------------------------------------------------------------------------------------
  // The rows' results stay in registers (lane `it` holds row it of this wave) until the loop is over: the compiler
  // orders every LDS access behind an in-flight global_load_lds with vmcnt(0).
  using acc_t = typename std::conditional<ACC == 0, double, float>::type;
  float res_v = 0.f, res_f = 0.f;
  uint32_t res_flag = 0u;
  const int margin = ACC == 0 ? kFragileUlpsDist : kFragileUlpsDist + acc_dist_ulps(NPLB);
  auto settle = [&](float dvv, float dff, int it) {               // one row's two sums -> lane `it`
    // rare: a sum within a few fp32 ulps of a T rounding boundary, where torch's own fp32 accumulation order
    // decides the result -> phase 2
    const uint32_t fl = !strict ? 0u
                                : ((strict == 2 || near_T_boundary<DT>(dvv, margin)) ? 1u : 0u) |
                                      ((strict == 2 || near_T_boundary<DT>(dff, margin)) ? 2u : 0u);
    if (lane == it) { res_v = rnT<DT>(dvv); res_f = rnT<DT>(dff); res_flag = fl; }
  };
  float held_v = 0.f, held_f = 0.f;                               // (fast path) the previous row's lane partials
  int it = 0;
  if constexpr (V2 != 0) {
    constexpr int NP = NPLB / 2;
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    uint32_t addr[NPLB];                                            // LDS byte address of the lane's elements
    {
      const uint32_t base = uint32_t(uintptr_t((lds_void_t*)buf0));
#pragma unroll
      for (int i = 0; i < NPLB; ++i) addr[i] = base + uint32_t(coff[i]) * uint32_t(ES);
    }
    // fp16: the centres as packed pairs
    uint32_t cvp[DT == VC2_F16 ? NP : 1], cfp[DT == VC2_F16 ? NP : 1];
    if constexpr (DT == VC2_F16) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        union { h2_t h; uint32_t u; } a, b;
        a.h = __builtin_convertvector((f2_t){cv[2 * k], cv[2 * k + 1]}, h2_t);
        b.h = __builtin_convertvector((f2_t){cf[2 * k], cf[2 * k + 1]}, h2_t);
        cvp[k] = a.u; cfp[k] = b.u;
      }
    }
    const size_t row_step = size_t(kRowWaves) * size_t(D) * ES;
    for (; it < cnt2; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float XA[NP], XB[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if constexpr (DT == VC2_BF16)
          asm volatile("ds_read_u16_d16_hi %0, %2\n\tds_read_u16_d16_hi %1, %3"
                       : "=&v"(XA[k]), "=&v"(XB[k]) : "v"(addr[2 * k]), "v"(addr[2 * k + 1]) : "memory");
        else
          asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %3"
                       : "=&v"(XA[k]), "=&v"(XB[k]) : "v"(addr[2 * k]), "v"(addr[2 * k + 1]) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(XA[k]), "+v"(XB[k]));
      if (it + 1 < cnt2) s2_issue_row<NCH2>(xl2 + size_t(it + 1) * row_step, buf0);
      const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dn), it));
      const bool exact_div = __builtin_amdgcn_readlane(int(my_rf), it) != 0;
      float pv = 0.f, pf = 0.f;
      if constexpr (DT == VC2_BF16) {
        using AR = DistArith<VC2_BF16, 1>;
        auto body = [&](auto exact_tag) {
          constexpr bool kExact = decltype(exact_tag)::value;
          const double inv = kExact ? 1.0 / double(dn) : 0.0;
          const float r = kExact ? 0.f : __builtin_amdgcn_rcpf(dn);
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            uint32_t xh;
            if constexpr (kExact) xh = AR::pk(div_via_f64(XA[k], inv), div_via_f64(XB[k], inv));
            else { const f2_t pr = pk_mul_f32((f2_t){XA[k], XB[k]}, (f2_t){r, r}); xh = AR::pk(pr.x, pr.y); }
            const float xa = __uint_as_float(xh << 16), xb = __uint_as_float(xh & 0xFFFF0000u);
            const uint32_t dv = AR::pk(xa - cv[2 * k], xb - cv[2 * k + 1]);
            const uint32_t df = AR::pk(xa - cf[2 * k], xb - cf[2 * k + 1]);
            const float va = __uint_as_float(dv << 16), vb = __uint_as_float(dv & 0xFFFF0000u);
            const float fa = __uint_as_float(df << 16), fb = __uint_as_float(df & 0xFFFF0000u);
            pv = AR::dot_ones(AR::pk(va * va, vb * vb), pv);
            pf = AR::dot_ones(AR::pk(fa * fa, fb * fb), pf);
          }
        };
        if (exact_div) body(std::true_type{}); else body(std::false_type{});
      } else {
        const h2_t ones = {static_cast<_Float16>(1.0f), static_cast<_Float16>(1.0f)};
        auto body = [&](auto exact_tag) {
          constexpr bool kExact = decltype(exact_tag)::value;
          const double inv = kExact ? 1.0 / double(dn) : 0.0;
          const float r = kExact ? 0.f : __builtin_amdgcn_rcpf(dn);
          const f2_t r2 = {r, r}, d2 = {dn, dn};
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            union { uint16_t u; _Float16 h; } ca, cb;
            ca.u = uint16_t(__float_as_uint(XA[k])); cb.u = uint16_t(__float_as_uint(XB[k]));
            const f2_t xx = {float(ca.h), float(cb.h)};
            h2_t xh;
            if constexpr (kExact) xh = __builtin_convertvector((f2_t){div_via_f64(xx.x, inv), div_via_f64(xx.y, inv)}, h2_t);
            else {
              const f2_t q0 = pk_mul_f32(xx, r2);
              const f2_t e = pk_fnma_f32(d2, q0, xx);
              xh = __builtin_convertvector(pk_fma_f32(e, r2, q0), h2_t);
            }
            union { uint32_t u; h2_t h; } a, b;
            a.u = cvp[k]; b.u = cfp[k];
            const h2_t dv = xh - a.h, df = xh - b.h;
            pv = __builtin_amdgcn_fdot2(dv * dv, ones, pv, false);
            pf = __builtin_amdgcn_fdot2(df * df, ones, pf, false);
          }
        };
        if (exact_div) body(std::true_type{}); else body(std::false_type{});
      }
      // two rows' lane partials are reduced together (four values: one permlane32 / permlane16 fold each)
      if (it & 1) {
        float q[4] = {held_v, held_f, pv, pf};
        wave_totals_f32<4>(q);
        auto at = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[0]), l)); };
        const float t0 = at(0), t2 = at(16), t1 = at(32), t3 = at(48);
        settle(t0, t1, it - 1);
        settle(t2, t3, it);
      } else {
        held_v = pv; held_f = pf;
      }
    }
    if (it & 1) settle(wave_sum_bcast_f32(held_v), wave_sum_bcast_f32(held_f), it - 1);   // the odd row out
  } else
  for (; n < n1; n += kRowWaves, ++it) {
    // issue the row's DMA, wait, compute straight from LDS
    if (it) row_issue<DT, VEC, VC2_AUX_S3>(x, int64_t(f) * N + n, D, CV, buf0, lane);
    row_wait();
    const float dn = dens[n - n0];
    const bool exact_div = !(kFast || (ACC == 1 && DT == VC2_F16)) || rfl[n - n0] != 0;
    static_assert(NPLB % 2 == 0, "compact positions are processed in pairs");
    if constexpr (kFast) {
      typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
      using AR = DistArith<VC2_BF16, 1>;
      float pv = 0.f, pf = 0.f;
      auto body = [&](auto exact_tag) {
        constexpr bool kExact = decltype(exact_tag)::value;
        const double inv = kExact ? 1.0 / double(dn) : 0.0;
        const float r = kExact ? 0.f : __builtin_amdgcn_rcpf(dn);
#pragma unroll
        for (int k = 0; k < NPLB / 2; ++k) {
          const float a = __uint_as_float(uint32_t(reinterpret_cast<const uint16_t*>(buf0)[coff[2 * k]]) << 16);
          const float b = __uint_as_float(uint32_t(reinterpret_cast<const uint16_t*>(buf0)[coff[2 * k + 1]]) << 16);
          // x^ pair, rounded by ONE packed conversion; then the DistArith sequence on it
          const uint32_t xh = kExact ? AR::pk(div_via_f64(a, inv), div_via_f64(b, inv)) : AR::pk(a * r, b * r);
          const float xa = __uint_as_float(xh << 16), xb = __uint_as_float(xh & 0xFFFF0000u);
          const uint32_t dv = AR::pk(xa - cv[2 * k], xb - cv[2 * k + 1]);
          const uint32_t df = AR::pk(xa - cf[2 * k], xb - cf[2 * k + 1]);
          const float va = __uint_as_float(dv << 16), vb = __uint_as_float(dv & 0xFFFF0000u);
          const float fa = __uint_as_float(df << 16), fb = __uint_as_float(df & 0xFFFF0000u);
          pv = AR::dot_ones(AR::pk(va * va, vb * vb), pv);
          pf = AR::dot_ones(AR::pk(fa * fa, fb * fb), pf);
        }
      };
      if (exact_div) body(std::true_type{}); else body(std::false_type{});
      // two rows' lane partials are reduced together (four values: one permlane32 / permlane16 fold each)
      if (it & 1) {
        float q[4] = {held_v, held_f, pv, pf};
        wave_totals_f32<4>(q);                                    // row r of the wave's 16-lane rows: value {0, 2, 1, 3}[r]
        auto at = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[0]), l)); };
        const float t0 = at(0), t2 = at(16), t1 = at(32), t3 = at(48);
        settle(t0, t1, it - 1);
        settle(t2, t3, it);
      } else {
        held_v = pv; held_f = pf;
      }
    } else if constexpr (ACC == 1 && DT == VC2_F16) {
      // fp16: the hardware's packed fp16 subtract / multiply ARE the reference's roundings here (DistArith)
      typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
      const h2_t ones = {static_cast<_Float16>(1.0f), static_cast<_Float16>(1.0f)};
      float pv = 0.f, pf = 0.f;
      // (round 5) rows sweep 2 divided with the fused-multiply-add quotient (rflag = 0: x and dn finite, dn a normal fp16
      // number) get the same quotient here -- it IS the IEEE quotient (tests/tools/check_f16_quotient.c), at three packed
      // fp32 instructions per pair instead of six fp64-rate ones
      auto body16 = [&](auto exact_tag) {
        constexpr bool kExact = decltype(exact_tag)::value;
        const double inv = kExact ? 1.0 / double(dn) : 0.0;
        const float r = kExact ? 0.f : __builtin_amdgcn_rcpf(dn);
        const f2_t r2 = {r, r}, d2 = {dn, dn};
#pragma unroll
        for (int k = 0; k < NPLB / 2; ++k) {
          const float v0 = lds_elem<DT>(buf0, coff[2 * k]), v1 = lds_elem<DT>(buf0, coff[2 * k + 1]);
          h2_t xh;
          if constexpr (kExact) xh = __builtin_convertvector((f2_t){div_via_f64(v0, inv), div_via_f64(v1, inv)}, h2_t);
          else {
            const f2_t xx = {v0, v1};
            const f2_t q0 = pk_mul_f32(xx, r2);
            const f2_t e = pk_fnma_f32(d2, q0, xx);
            xh = __builtin_convertvector(pk_fma_f32(e, r2, q0), h2_t);
          }
          const h2_t cvp = __builtin_convertvector((f2_t){cv[2 * k], cv[2 * k + 1]}, h2_t);    // (loop-invariant: hoisted)
          const h2_t cfp = __builtin_convertvector((f2_t){cf[2 * k], cf[2 * k + 1]}, h2_t);
          const h2_t dv = xh - cvp, df = xh - cfp;
          pv = __builtin_amdgcn_fdot2(dv * dv, ones, pv, false);
          pf = __builtin_amdgcn_fdot2(df * df, ones, pf, false);
        }
      };
      if (exact_div) body16(std::true_type{}); else body16(std::false_type{});
      settle(wave_sum_bcast_f32(pv), wave_sum_bcast_f32(pf), it);
    } else {
      acc_t pv = 0, pf = 0;                                       // accumulation order: i, then i + 1
      const double inv = 1.0 / double(dn);
#pragma unroll
      for (int i = 0; i < NPLB; i += 2) {
        const float v0 = lds_elem<DT>(buf0, coff[i]), v1 = lds_elem<DT>(buf0, coff[i + 1]);
        float xh0, xh1;
        rnT2<DT>(div_via_f64(v0, inv), div_via_f64(v1, inv), xh0, xh1);
        const f2_t d0 = rnT2v<DT>((f2_t){xh0 - cv[i], xh0 - cf[i]});
        const f2_t d1 = rnT2v<DT>((f2_t){xh1 - cv[i + 1], xh1 - cf[i + 1]});
        const f2_t q0 = rnT2v<DT>(pk_square(d0)), q1 = rnT2v<DT>(pk_square(d1));
        pv += acc_t(q0.x); pf += acc_t(q0.y);
        pv += acc_t(q1.x); pf += acc_t(q1.y);
      }
      if constexpr (ACC == 0) settle(float(wave_sum_bcast(pv)), float(wave_sum_bcast(pf)), it);
      else settle(wave_sum_bcast_f32(pv), wave_sum_bcast_f32(pf), it);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // LDS reads done before the buffer is refilled
  }
  if constexpr (kFast && V2 == 0) {
    if (it & 1) settle(wave_sum_bcast_f32(held_v), wave_sum_bcast_f32(held_f), it - 1);   // the odd row out
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane < it) {
    const int nl = wave + lane * kRowWaves;
    dists[2 * nl] = res_v;
    dists[2 * nl + 1] = res_f;
    if (res_flag & 1u) list[atomicAdd(lcount, 1)] = nl * 2;
    if (res_flag & 2u) list[atomicAdd(lcount, 1)] = nl * 2 + 1;
  }
  __syncthreads();
  // ---- phase 2: replay torch's cascade sum for the listed (row, centre) pairs ---------------------------
  VC2_WGTIME(0, 0);                                               // (debug builds: phase 2 begins / ends)
  if (strict) {
    const int cnt = *lcount;
    // The row comes back by DMA into wave 0's (idle) row buffer while every thread fetches its column indices, sorted
    // positions and centre values -- ONE round of global loads per entry instead of two dependent ones (cols -> x) at
    // a time when 900 other workgroups are streaming rows (in-kernel stamps: 4.1 us of an entry's 5.2 were spent
    // getting the squares into LDS).  The squares go to wave 1's buffer.
    constexpr int kP2 = (NPLB * 64 + kRowWaves * 64 - 1) / (kRowWaves * 64);        // positions per thread
    const bool dma = DT != VC2_F32 && VEC > 1 && size_t(C) * 4 + 16 <= rowb && C <= NPLB * 64;
    float* sq = reinterpret_cast<float*>(dma ? smem + rowb : smem);
    for (int e = 0; e < cnt; ++e) {
      const int ent = list[e];
      const int nl = ent >> 1, which = ent & 1;
      const int64_t row = int64_t(f) * N + n0 + nl;
      const double inv = 1.0 / double(dens[nl]);
      const float* cen = which ? fc + int64_t(f) * C : vc;
      if (dma) {
        if (wave == 0) row_issue<DT, VEC, VC2_AUX_S3>(x, row, D, CV, smem, lane);
        int colv[kP2], sppv[kP2];
        float cev[kP2];
#pragma unroll
        for (int i = 0; i < kP2; ++i) {
          const int p = tid + i * kRowWaves * 64, pc = p < C ? p : C - 1;
          colv[i] = cols ? cols[pc] : pc;
          sppv[i] = spos ? spos[pc] : pc;
          cev[i] = cen[pc];
        }
        if (wave == 0) row_wait();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kP2; ++i) {
          const int p = tid + i * kRowWaves * 64;
          const float xh = rnT<DT>(div_via_f64(lds_elem<DT>(smem, colv[i]), inv));
          const float a = rnT<DT>(xh - cev[i]);
          if (p < C) sq[sppv[i]] = rnT<DT>(a * a);
        }
      } else {
        for (int p = tid; p < C; p += kRowWaves * 64) {
          const int col = cols ? cols[p] : p, spp = spos ? spos[p] : p;
          const float xh = rnT<DT>(div_via_f64(ldT<DT>(x, row * D + col), inv));
          const float a = rnT<DT>(xh - cen[p]);
          sq[spp] = rnT<DT>(a * a);
        }
      }
      __syncthreads();
      if (e == 0) VC2_WGTIME(3, 0);                              // (debug builds: first entry's squares are in LDS)
      if (wave == 0) {
        const float r = sum_torch_order<DT>(sq, C, lane);
        if (lane == 0) dists[ent] = rnT<DT>(r);
      }
      __syncthreads();
      if (e == 0) VC2_WGTIME(3, 1);
    }
  }
  // ---- phase 3: Gaussian sums, total, partial frame sum ---------------------------------------------
  VC2_WGTIME(0, 1);
  for (int t = tid; t < nrows * 10; t += kRowWaves * 64) {
    const int nl = t / 10, j = t - nl * 10;
    float e;
    if (!gauss_term_fast<DT>(dists[2 * nl + (j >= 5 ? 1 : 0)], j >= 5 ? j - 5 : j, e)) elist[atomicAdd(lcount + 1, 1)] = uint16_t(t);
    ebuf[t] = e;
  }
  __syncthreads();
  for (int k = tid; k < lcount[1]; k += kRowWaves * 64) {         // the few terms next to a rounding boundary: fp64 exp
    const int t = elist[k], nl = t / 10, j = t - nl * 10;
    ebuf[t] = gauss_term<DT>(dists[2 * nl + (j >= 5 ? 1 : 0)], j >= 5 ? j - 5 : j);
  }
  __syncthreads();
  if (wave == 0) {
    double vd = 0.0;
    if (lane < nrows) {
      const float* e = ebuf + lane * 10;
      float v = e[0], g = e[5];
#pragma unroll
      for (int a = 1; a < 5; ++a) { v = rnT<DT>(v + e[a]); g = rnT<DT>(g + e[5 + a]); }
      const int64_t row = int64_t(f) * N + n0 + lane;
      if (v_T) stT<DT>(v_T, row, v);
      if (f_T) stT<DT>(f_T, row, g);
      total[row] = rnT<DT>(v + g);
      vd = double(v);
    }
    vd = wave_sum(vd);                                            // fixed tree over T values: exact in fp64
    if (lane == 0) vpart[f * S + sp] = vd;
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) VC2_STAMP(709);
    VC2_WGTIME(2, 1);
  }
}

// s[f] = -mean_T(sum of the frame's v) from the S workgroup partials of k_dist (vidcom2.py:32)
template <int DT>
__global__ void k_frame_scores(const double* __restrict__ vpart, int F, int S, int N, float* __restrict__ s_out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double t = 0.0;
  for (int i = 0; i < S; ++i) t += vpart[int64_t(f) * S + i];
  s_out[f] = -mean_T<DT>(t, N);
}

// ======================================================================================
// budgets (vidcom2.py:64-68, :72): single workgroup over the F frame scores
// ======================================================================================
constexpr int kBudNT = 256;

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sm) {
  v = wave_max_nanprop(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int i = 1; i < 4; ++i) {
    const float t = sm[i];
    r = (r != r) ? r : ((t != t) ? t : fmaxf(r, t));
  }
  return r;
}

// scales = clamp(base * (1 + softmax((s - max s)/temp) - mean(softmax)), max=1) with every op in T.
template <int DT>
__device__ __forceinline__ void scales_body(const float* __restrict__ s, int F, float base, float temp,
                                            float* __restrict__ zbuf, float* __restrict__ scales_f32,
                                            void* __restrict__ scales_T) {
  __shared__ double smd[4];
  __shared__ float smf[4];
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  bool first = true;
  for (int i = tid; i < F; i += kBudNT) {
    const float v = s[i];
    mx = first ? v : ((mx != mx) ? mx : ((v != v) ? v : fmaxf(mx, v)));
    first = false;
  }
  mx = block_max_256(mx, smf);
  float zmax = -INFINITY;
  for (int i = tid; i < F; i += kBudNT) {
    const float d = rnT<DT>(s[i] - mx);
    const float z = rnT<DT>(d / temp);
    zbuf[i] = z;
    zmax = (zmax != zmax) ? zmax : ((z != z) ? z : fmaxf(zmax, z));
  }
  zmax = block_max_256(zmax, smf);
  double es = 0.0;
  for (int i = tid; i < F; i += kBudNT) es += double(float(exp(double(zbuf[i] - zmax))));
  const double esum = block_sum_256(es, smd);
  double ps = 0.0;
  for (int i = tid; i < F; i += kBudNT) {
    const double e = double(float(exp(double(zbuf[i] - zmax))));
    const float p = rnT<DT>(float(e / esum));
    zbuf[i] = p;
    ps += double(p);
  }
  const double psum = block_sum_256(ps, smd);
  const float pmean = mean_T<DT>(psum, F);
  for (int i = tid; i < F; i += kBudNT) {
    float t = rnT<DT>(1.0f + zbuf[i]);
    t = rnT<DT>(t - pmean);
    t = rnT<DT>(base * t);
    if (t > 1.0f) t = 1.0f;
    if (scales_f32) scales_f32[i] = t;
    if (scales_T) stT<DT>(scales_T, i, t);
  }
}

template <int DT>
__global__ __launch_bounds__(kBudNT) void k_scales(const float* __restrict__ s, int F, float base, float temp,
                                                   float* __restrict__ zbuf, float* __restrict__ scales_f32,
                                                   void* __restrict__ scales_T) {
  scales_body<DT>(s, F, base, temp, zbuf, scales_f32, scales_T);
}

// _multi_scale_gaussian(x, center, alphas) as a standalone call (vidcom2.py:59-62): x T[R, C] is taken
// as given (already normalised / channel-selected by the caller), centre T[1 | R/N, C].  One wave per
// row: RN_T(RN_T(x - c)^2) per element, the row sum exact (fp64) or -- `strict` -- in torch's cascade
// order over the columns as they are stored, then the Gaussian sum over the caller's scales.
constexpr int kMsgWaves = 4;
constexpr int kMaxAlphas = 16;
struct MsgAlphas { float two_a[kMaxAlphas]; int n; };

template <int DT>
__global__ __launch_bounds__(kMsgWaves * 64) void k_multi_scale_gaussian(const void* __restrict__ x, int64_t R,
                                                                         int C, const void* __restrict__ centre,
                                                                         int per_frame, int N, MsgAlphas al,
                                                                         int strict, void* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sq = reinterpret_cast<float*>(smem) + size_t(wave) * C;      // strict only
  for (int64_t r = int64_t(blockIdx.x) * kMsgWaves + wave; r < R; r += int64_t(gridDim.x) * kMsgWaves) {
    const int64_t cb = per_frame ? (r / N) * C : 0;
    double acc = 0.0;
    for (int p = lane; p < C; p += 64) {
      const float a = rnT<DT>(ldT<DT>(x, r * C + p) - ldT<DT>(centre, cb + p));
      const float q = rnT<DT>(a * a);
      if (strict) sq[p] = q; else acc += double(q);
    }
    float dist;
    if (strict) {
      wave_lds_fence();
      dist = rnT<DT>(sum_torch_order<DT>(sq, C, lane));
      wave_lds_fence();
    } else {
      dist = rnT<DT>(float(wave_sum(acc)));
    }
    if (lane == 0) {
      float g = 0.f;
      for (int a = 0; a < al.n; ++a) {
        const float arg = rnT<DT>((-dist) / al.two_a[a]);
        const float e = rnT<DT>(float(exp(double(arg))));
        g = (a == 0) ? e : rnT<DT>(g + e);
      }
      stT<DT>(out, r, g);
    }
  }
}

// k_f = clamp_min(long(round(RN_T(scale_f * tpf))), 1)   (vidcom2.py:72)
template <int DT> __device__ __forceinline__ int budget_k(float scale, int tpf) {
  float t = rnT<DT>(scale * float(tpf));
  t = rintf(t);                                    // round-half-even, like torch.round
  // NaN.long() clamps to 1 here (clamp(min=1) of the minimum int64); not clamped from above: torch.topk raises when
  // k exceeds the row length (the callers take min(k, N) for the work and report the raw k)
  return (t != t) ? 1 : (t < 1.f ? 1 : (t > 1.0e9f ? 1000000000 : int(t)));
}

template <int DT>
__global__ void k_widen(const void* __restrict__ in, int64_t n, float* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ldT<DT>(in, i);
}

// ======================================================================================
// budgets + per-frame selection + index mapping (vidcom2.py:64-77, :99-115): one workgroup per frame
// ======================================================================================
// Every workgroup (4 waves) first derives the budgets of ALL frames -- compute_scales over the F frame scores, one
// frame per thread, five block reductions; the scores come straight from sweep 3's workgroup partials (vpart) --
// and with them its own k and output offset; then replays torch.topk's selection on its N scores (one wave for
// N <= 1020, all four beyond; vc2_select2.h) and writes the kept token indices ascending, already mapped
// (linear / grid_vid / local).  No host round trip, no separate budget kernel.
constexpr int kFrameNT = 256;
constexpr int kFusedScalesMaxF = 1024;      // up to this many frames every k_select workgroup derives the budgets itself

// NaN-propagating block maximum (256 threads).  sm: exchange cells nobody has written before in this launch -- no barrier in
// front of the write
__device__ __forceinline__ float block_max_nanprop_256_once(float v, float* sm) {
  v = wave_max_nanprop_bcast(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int i = 1; i < 4; ++i) { const float t = sm[i]; r = (r != r) ? r : ((t != t) ? t : fmaxf(r, t)); }
  return r;
}

// the frame's (score, index) words into LDS: issued at the top of k_select, so the loads' latency runs under the
// budget arithmetic (which only needs the static shared arrays)
template <typename W>
__device__ __forceinline__ void select_frame_load(unsigned char* smem, const float* __restrict__ total, int f, int N) {
  using T = WordTr<W>;
  Sel2<W> S = sel2_carve<W>(smem, N);
  for (int i = threadIdx.x; i < N; i += kFrameNT) S.w[i] = T::pack(topk_key(total[int64_t(f) * N + i]), i);
}

template <int DT, typename W>
__device__ __forceinline__ void select_frame_body(unsigned char* smem, const float* __restrict__ total, int f, int N, int k,
                                  int64_t o0, int map_mode, int grid_h, int64_t stride, int64_t cap,
                                  int64_t* __restrict__ idx_out, int* status) {
  using T = WordTr<W>;
  const int tid = threadIdx.x;
  Sel2<W> S = sel2_carve<W>(smem, N, status);
#if defined(VC2_DEBUG_TIMING)
  S.dbg_slot = f == 0 ? 1 : -1;
#endif
  // (the frame's words were packed into S.w by select_frame_load before the budgets were derived)
  __syncthreads();
  if (tid == 0 && f == 0) VC2_STAMP(802);
  if (tid == 0) VC2_ROUND(S, 802, N);
  if (N > sel2_capacity(1, 4)) topk_smallest2<W, 4, 4, 8>(S, N, k, tid);
  else if (tid < 64) {
    // round 6: 32-bit words, frames of <= 512 tokens: barrier-free rounds on registers (sel4_round<1, true>; the LDS rounds
    // cost 1.2-1.8 us each above 64 elements, these 0.7-0.8).  -DVC2_NO_SEL4_SOLO: the LDS rounds of rounds 3-5;
    // -DVC2_SEL3_SOLO: the ballot form (sel3_rounds; measured no faster than the LDS rounds)
#if !defined(VC2_NO_SEL4_SOLO) && !defined(VC2_SEL3_SOLO)
    if constexpr (sizeof(W) == 4) {
      if (N <= 256) topk_smallest4_solo<4>(S, N, k, tid);
      else if (N <= 512) topk_smallest4_solo<8>(S, N, k, tid);
      else topk_smallest2<W, 1, 4, 4>(S, N, k, tid);
    } else
#elif defined(VC2_SEL3_SOLO)
    if constexpr (sizeof(W) == 4) {
      if (N <= 256) topk_smallest3_solo<4>(S, N, k, tid);
      else if (N <= 512) topk_smallest3_solo<8>(S, N, k, tid);
      else topk_smallest2<W, 1, 4, 4>(S, N, k, tid);
    } else
#endif
    topk_smallest2<W, 1, 4, 4>(S, N, k, tid);
  }
  __syncthreads();
  if (tid == 0 && f == 0) VC2_STAMP(803);
  if (tid == 0) VC2_ROUND(S, 803, k);
  // kept flags (la is free now), then ordered compaction = idx.sort().values.  S.w holds every token exactly once, the
  // kept ones in [0, k): one pass writes every flag
  for (int i = tid; i < N; i += kFrameNT) S.la[T::idx(S.w[i])] = (i < k || k >= N) ? 1 : 0;
  __syncthreads();
  const int Ept = (N + kFrameNT - 1) / kFrameNT;
  const int b = tid * Ept, e = min(N, b + Ept);
  uint32_t cnt = 0;
  for (int p = b; p < e; ++p) cnt += S.la[p];
  uint32_t tot;
  int64_t o = o0 + block_excl_scan<kFrameNT / 64>(cnt, S.xch, tot);
  for (int p = b; p < e; ++p) {
    if (S.la[p]) {
      int64_t g;
      if (map_mode == VC2_MAP_LINEAR) g = int64_t(p) + int64_t(f) * stride;
      else if (map_mode == VC2_MAP_GRID_VID)
        g = int64_t(f) * grid_h * (grid_h + 1) + int64_t(p / grid_h) * (grid_h + 1) + (p % grid_h);
      else g = p;
      if (o < cap) idx_out[o] = g;
      ++o;
    }
  }
  if (map_mode == VC2_MAP_GRID_VID) {   // the frame's h newline rows follow its kept tokens
    for (int a = tid; a < grid_h; a += kFrameNT) {
      const int64_t oo = o0 + k + a;
      if (oo < cap) idx_out[oo] = int64_t(f) * grid_h * (grid_h + 1) + int64_t(a) * (grid_h + 1) + grid_h;
    }
  }
  if (tid == 0) VC2_ROUND(S, 809, k);
}

// Budget sources, in order of preference: vpart (sweep-3 workgroup partials, S2 per frame) or frame_scores
// (fp32-widened s[F]) -> compute_scales here; else scales_f32[F] as given (the stage API and F > 1024).
// F_sel / f0: the frames this launch selects ([f0, f0 + gridDim.x) of the F budget frames; the frame-sharded pass
// budgets over the whole video and selects its own frames).
template <int DT>
__global__ __launch_bounds__(kFrameNT) void k_select(const float* __restrict__ total,
                                                     const float* __restrict__ scales_f32, int F, int f0, int N,
                                                     int tpf, int map_mode, int grid_h, int64_t stride, int64_t cap,
                                                     int64_t* __restrict__ ks, int64_t* __restrict__ offs,
                                                     int64_t* __restrict__ idx_out, int64_t* __restrict__ K_out,
                                                     const double* __restrict__ vpart, int S2,
                                                     const float* __restrict__ frame_scores, float base,
                                                     float temp, float* __restrict__ scales_out,
                                                     int* status = nullptr, long long* khost = nullptr,
                                                     int* arrive = nullptr, int force_guard = -1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // every block-wide reduction of the budget phase has exchange cells of its own: written once per launch, so ONE barrier
  // per reduction (write, barrier, read) instead of two -- 5 barriers on the budget path instead of 12 (round 5)
  __shared__ float smf[2][4];
  __shared__ double smd[2][4];
  __shared__ long long smi[8];
  __shared__ long long smk;
  __shared__ float smsc;
  const int fl = blockIdx.x, FS = gridDim.x;                   // local frame, frames selected by this launch
  const int f = f0 + fl;                                        // its index among the F budget frames
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int extra = map_mode == VC2_MAP_GRID_VID ? grid_h : 0;
  if (tid == 0 && (fl == 0 || fl == FS - 1)) VC2_STAMP(800 + (fl ? 50 : 0));
  if (tid == 0 && fl == 0) VC2_ROUND_RAW(1, 120, 800);
  if constexpr (DT == VC2_F32) select_frame_load<uint64_t>(smem, total, fl, N);
  else select_frame_load<uint32_t>(smem, total, fl, N);
  // ---- budgets of all frames; thread t holds frames t, t + 256, ...
  constexpr int FPT = kFusedScalesMaxF / kFrameNT;
  long long before = 0, all = 0;
  int kmine = 1;
  float scmine = 0.f;
  if (vpart || frame_scores) {
    float sv[FPT], zz[FPT], pp[FPT];
    float m = -INFINITY;
    bool first = true;
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
      const int i = tid + j * kFrameNT;
      sv[j] = 0.f;
      if (i < F) {
        if (vpart) {
          double t = 0.0;
          for (int q0 = 0; q0 < S2; q0 += 8) {                  // eight loads in flight, added in split order
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = vpart[int64_t(i) * S2 + min(q0 + u, S2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (q0 + u < S2) t += pv[u];
          }
          sv[j] = -mean_T<DT>(t, N);                              // vidcom2.py:32
        } else {
          sv[j] = frame_scores[i];
        }
        const float v = sv[j];
        m = first ? v : ((m != m) ? m : ((v != v) ? v : fmaxf(m, v)));
        first = false;
      }
    }
    // a thread without frames must not inject -inf into a NaN-propagating max of finite values: -inf is the
    // identity there as well (max(-inf, v) = v), and NaN still wins
    const float mx = block_max_nanprop_256_once(m, smf[0]);
    float zm = -INFINITY;
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
      const int i = tid + j * kFrameNT;
      zz[j] = 0.f;
      if (i < F) {
        zz[j] = rnT<DT>(rnT<DT>(sv[j] - mx) / temp);
        const float z = zz[j];
        zm = (zm != zm) ? zm : ((z != z) ? z : fmaxf(zm, z));
      }
    }
    const float zmax = block_max_nanprop_256_once(zm, smf[1]);
    double es = 0.0;
    float ee[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
      const int i = tid + j * kFrameNT;
      ee[j] = 0.f;
      if (i < F) { ee[j] = float(exp(double(zz[j] - zmax))); es += double(ee[j]); }
    }
    es = wave_sum_bcast(es);
    if (lane == 0) smd[0][wave] = es;
    __syncthreads();
    const double esum = smd[0][0] + smd[0][1] + smd[0][2] + smd[0][3];
    double ps = 0.0;
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
      const int i = tid + j * kFrameNT;
      pp[j] = 0.f;
      if (i < F) { pp[j] = rnT<DT>(float(double(ee[j]) / esum)); ps += double(pp[j]); }
    }
    ps = wave_sum_bcast(ps);
    if (lane == 0) smd[1][wave] = ps;
    __syncthreads();
    const float pmean = mean_T<DT>(smd[1][0] + smd[1][1] + smd[1][2] + smd[1][3], F);
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
      const int i = tid + j * kFrameNT;
      if (i < F) {
        float t = rnT<DT>(1.0f + pp[j]);
        t = rnT<DT>(t - pmean);
        t = rnT<DT>(base * t);
        if (t > 1.0f) t = 1.0f;
        const int kr = budget_k<DT>(t, tpf), kk = kr < N ? kr : N;
        if (i >= f0 && i < f) before += kk + extra;
        if (i >= f0 && i < f0 + FS) all += kk + extra;
        if (i == f) { kmine = kr; scmine = t; }
      }
    }
  } else {
    for (int i = f0 + tid; i < f0 + FS; i += kFrameNT) {
      const int kr = budget_k<DT>(scales_f32[i], tpf), kk = kr < N ? kr : N;
      if (i < f) before += kk + extra;
      all += kk + extra;
      if (i == f) { kmine = kr; scmine = scales_f32[i]; }
    }
  }
  // block sums of (before, all) and broadcast of this frame's k (exactly one thread holds it)
  // (Budgets by ONE wave with wave-level reductions only -- no barriers -- were built twice, rounds 3 and 4: 1.2 us
  //  SLOWER than this block-wide form, whose ten barriers cost less than one wave's serial exp / load slots.)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); all += __shfl_xor(all, o, 64); }
  if (lane == 0) { smi[wave] = before; smi[4 + wave] = all; }
  // the thread that holds frame f publishes k and the scale (cells of their own: the same barrier serves both)
  {
    const bool holder = (vpart || frame_scores) ? (tid == f % kFrameNT) : (tid == (f - f0) % kFrameNT);
    if (holder) { smk = kmine; smsc = scmine; }
  }
  __syncthreads();
  const int64_t o0 = smi[0] + smi[1] + smi[2] + smi[3];
  const int64_t Ktot = smi[4] + smi[5] + smi[6] + smi[7];
  const int kraw = int(smk);                                    // round(scale * tpf): may exceed N when tpf != N
  const float scale_f = smsc;
  if (tid == 0 && (fl == 0 || fl == FS - 1)) VC2_STAMP(801 + (fl ? 50 : 0));
  if (tid == 0 && fl == 0) VC2_ROUND_RAW(1, 121, 801);
  const int k = kraw < N ? kraw : N;
  if (tid == 0) {
    ks[fl] = kraw;                                              // (the caller turns k > N into torch.topk's error)
    offs[fl] = o0;
    if (scales_out) scales_out[fl] = scale_f;
    if (fl == FS - 1) {
      offs[FS] = Ktot;
      K_out[0] = Ktot;
      // status word: 1 = capacity exceeded, 2 = a bounded wait inside a launch of this pass expired (kTkStatus),
      // 4 = a loop bound of the selection engine expired since the last pass that reported (vc2_select2.h)
      // (bit 4 covers the selection replays of the EARLIER kernels of the pass -- channel selection, ORDER riders; a hit
      //  inside this launch's own replays is ORed in by the workgroup that sees it, below)
      const int stw = status ? __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      const long long wv = (Ktot > cap ? 1 : 0) | ((stw & kStatusSpinExpired) ? 2 : 0) | ((stw & kSel2StatusGuard) ? 4 : 0);
      // (the word is zero when the launch begins -- the pass's first kernel or launch_select's memset -- and every
      // workgroup ORs its bits in: order-free)
      atomicOr(reinterpret_cast<unsigned long long*>(K_out + 1), (unsigned long long)wv);
      if (khost) {
        // the host's mirror of (K, status) in pinned memory: the caller only needs K to slice its outputs, so it can go
        // on while the gather launch is still running (everything after is stream-ordered).  Status first, then the
        // count the host polls for.  This is the EARLY status: a selection-guard hit of another workgroup may land
        // later -- khost[2], written by the last workgroup to finish its selection (below), is the final word.
        __hip_atomic_store(khost + 1, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(khost, (long long)Ktot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __shared__ int st_local;
  if (tid == 0) st_local = 0;
  if constexpr (DT == VC2_F32) select_frame_body<DT, uint64_t>(smem, total, fl, N, k, o0, map_mode, grid_h, stride, cap, idx_out, &st_local);
  else select_frame_body<DT, uint32_t>(smem, total, fl, N, k, o0, map_mode, grid_h, stride, cap, idx_out, &st_local);
  __syncthreads();
  if (tid == 0 && fl == force_guard) st_local = 1;               // (test hook: vc2_selftest_force_guard)
  if (tid == 0 && st_local) {                                    // (cannot happen; reported, never swallowed)
    if (status) atomicOr(status, kSel2StatusGuard);
    atomicOr(reinterpret_cast<unsigned long long*>(K_out + 1), 4ull);
  }
  // The FINAL status for the host mirror: the last workgroup to get here (all selections of the launch are over, their
  // bits are in K_out[1]) publishes khost[2] = status | 2^62.  A host that returned on the early words re-reads it
  // before it trusts the next pass of the plan (CompressPlan._settle): a late guard hit is raised, never lost.
  if (tid == 0 && khost && arrive) {
    __threadfence();                                             // release my bits ...
    if (atomicAdd(arrive, 1) == FS - 1) {
      __threadfence();                                           // ... acquire everybody's
      const unsigned long long fin = atomicOr(reinterpret_cast<unsigned long long*>(K_out + 1), 0ull);
      __hip_atomic_store(khost + 2, (long long)(fin | (1ull << 62)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (tid == 0 && (fl == 0 || fl == FS - 1)) VC2_STAMP(809 + (fl ? 50 : 0));
}

// standalone mappers on already-selected local indices
__global__ void k_map_indices(const int64_t* __restrict__ local, const int64_t* __restrict__ ks,
                              const int64_t* __restrict__ offs, int map_mode, int64_t stride_or_h,
                              int64_t* __restrict__ out) {
  const int f = blockIdx.x;
  const int64_t k = ks[f], o0 = offs[f];
  if (map_mode == VC2_MAP_GRID_VID) {
    const int64_t h = stride_or_h, w = h + 1;
    const int64_t d0 = o0 + int64_t(f) * h;
    for (int64_t j = threadIdx.x; j < k; j += blockDim.x) {
      const int64_t p = local[o0 + j];
      out[d0 + j] = int64_t(f) * h * w + (p / h) * w + (p % h);
    }
    for (int64_t a = threadIdx.x; a < h; a += blockDim.x) out[d0 + k + a] = int64_t(f) * h * w + a * w + h;
  } else {
    for (int64_t j = threadIdx.x; j < k; j += blockDim.x)
      out[o0 + j] = local[o0 + j] + int64_t(f) * stride_or_h;
  }
}

// Row gather / scatter (vidcom2.py:91,96 and the splice that follows it in the hooks): up to kGSMaxSrc tensors with
// the SAME row length share one index list -- dst_t[dst_pos[j]] = src_t[idx[j]] for j < n -- and `tail_rows` extra
// rows (the LLaVA newline embedding) land right behind the gathered ones in dst_0, all in one launch, so every kept
// row is written once, at its final position.  One workgroup per (row, tensor), 16 B per lane; n is read on the
// device when the selection kernel of the same stream produced it.  A row outside its tensor is skipped and reported
// (bit 1 of *status).
constexpr int kGSMaxSrc = 8;
struct GSArgs {
  const unsigned char* src[kGSMaxSrc];
  unsigned char* dst[kGSMaxSrc];
  int64_t src_rows[kGSMaxSrc], dst_rows[kGSMaxSrc];
  int64_t row_bytes;
  const int64_t* idx;        // [n] source rows (null: j)
  const int64_t* n_dev;      // n on the device (null: n_max)
  int64_t n_max;
  const int64_t* dst_pos;    // [n] destination rows (null: dst_row0 + j)
  int64_t dst_row0;
  const unsigned char* tail; // rows appended to dst[0] behind the gathered ones (null: none)
  int64_t tail_rows;
  int* status;               // null, or a device word: |= 2 when a row index is out of range
};
__device__ __forceinline__ void copy_row(const unsigned char* __restrict__ s, unsigned char* __restrict__ d,
                                         int64_t row_bytes) {
  if ((row_bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
    const int64_t nv = row_bytes >> 4;
    for (int64_t t = threadIdx.x; t < nv; t += blockDim.x) {
#ifndef VC2_GATHER_NO_NT      // streaming stores: the kept rows are consumed much later (gather 16.1 -> 15.5 us)
      const uint4 v = reinterpret_cast<const uint4*>(s)[t];                     // (non-temporal loads too: no difference)
      __builtin_nontemporal_store(v.x, &reinterpret_cast<uint32_t*>(d)[4 * t]);
      __builtin_nontemporal_store(v.y, &reinterpret_cast<uint32_t*>(d)[4 * t + 1]);
      __builtin_nontemporal_store(v.z, &reinterpret_cast<uint32_t*>(d)[4 * t + 2]);
      __builtin_nontemporal_store(v.w, &reinterpret_cast<uint32_t*>(d)[4 * t + 3]);
#else
      reinterpret_cast<uint4*>(d)[t] = reinterpret_cast<const uint4*>(s)[t];
#endif
    }
  } else {
    for (int64_t t = threadIdx.x; t < row_bytes; t += blockDim.x) d[t] = s[t];
  }
}
__global__ __launch_bounds__(256) void k_gather_rows(GSArgs a) {   // (448 / 512 threads per row: 18.0-18.3 instead of 15.6 us)
  const int t = blockIdx.y;
  if (blockIdx.x == 0 && t == 0 && threadIdx.x == 0) VC2_STAMP(900);
  const int64_t n = a.n_dev ? min(a.n_dev[0], a.n_max) : a.n_max;
  const int64_t total = n + (t == 0 ? a.tail_rows : 0);
  for (int64_t j = blockIdx.x; j < total; j += gridDim.x) {
    if (j < n) {
      const int64_t r = a.idx ? a.idx[j] : j;
      const int64_t d = a.dst_pos ? a.dst_pos[j] : a.dst_row0 + j;
      if (r < 0 || r >= a.src_rows[t] || d < 0 || d >= a.dst_rows[t]) {
        if (a.status && threadIdx.x == 0) atomicOr(a.status, 2);
        continue;
      }
      copy_row(a.src[t] + r * a.row_bytes, a.dst[t] + d * a.row_bytes, a.row_bytes);
    } else {                                                   // (t == 0) the tail rows, behind the last gathered one
      const int64_t d = a.dst_row0 + j;
      if (d >= a.dst_rows[0]) { if (a.status && threadIdx.x == 0) atomicOr(a.status, 2); continue; }
      copy_row(a.tail + (j - n) * a.row_bytes, a.dst[0] + d * a.row_bytes, a.row_bytes);
    }
  }
  if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && t == 0 && threadIdx.x == 0) VC2_STAMP(909);
}

// Sequence positions a pruned prefill keeps (hooks, reference models/qwen2_5_vl.py:153-160): every position that is
// not a video token, plus the video tokens whose ordinal is in `kept` (ascending).  keep_out = those positions in
// order; vis_rows_out (optional) = for the positions flagged in visual_mask, the ordinals (among them) that are
// kept -- the rows Qwen3-VL's deep-stack tensors keep (qwen3_vl.py:141-149).  counts_out = {n_keep, n_vis_rows}.
// One workgroup of 1024 threads, thread-contiguous chunks of positions.  Round 5: the kept ordinals go into an LDS
// BITMAP first (one coalesced sweep of kept[], which also checks that it ascends), so that "is video token number o
// kept" is an LDS bit test -- rounds 2-4 walked kept[] with a binary search and a merge per thread, a chain of dependent
// global loads per position (54.8 us for 20 832 positions); the mask bytes of a chunk are fetched sixteen at a time.
// counts_out may be pinned host memory (vc2_keep_positions translates it): its words are stored system-scope, the error
// word LAST, so that a host spinning on it (vc2_wait_host_count on counts_out + 2, preset to -1) needs no copy.
constexpr int kKeepNT = 1024;
constexpr int64_t kKeepBitmapMaxS = int64_t(120) * 1024 * 8;      // positions whose bitmap fits the workgroup's LDS
// V > 0: a thread owns 16 V consecutive positions and keeps their mask bytes in REGISTERS (one round of 16-byte loads,
// S <= 1024 * 16 V); V = 0: thread-contiguous chunks of any length, re-read sixteen bytes at a time in every pass.
template <typename F>
__device__ __forceinline__ void keep_walk(const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int64_t b, int64_t e, F f) {
  for (int64_t p0 = b; p0 < e; p0 += 16) {                        // sixteen mask bytes (of each mask) in flight
    uint8_t a[16], v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int64_t p = p0 + u < e ? p0 + u : e - 1; a[u] = m0[p]; v[u] = m1 ? m1[p] : uint8_t(0); }
#pragma unroll
    for (int u = 0; u < 16; ++u) if (p0 + u < e) f(p0 + u, a[u] != 0, v[u] != 0);
  }
}
template <int V>
__global__ __launch_bounds__(kKeepNT) void k_keep_positions(const uint8_t* __restrict__ video_mask, int64_t S,
                                                            const int64_t* __restrict__ kept,
                                                            const int64_t* __restrict__ K_dev, int64_t K_max,
                                                            const uint8_t* __restrict__ visual_mask,
                                                            int64_t* __restrict__ keep_out, int64_t keep_cap,
                                                            int64_t* __restrict__ vis_rows_out, int64_t vis_cap,
                                                            int64_t* __restrict__ counts_out, int use_bitmap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t bitmap[];     // [ceil(S / 32)] when use_bitmap
  __shared__ uint32_t xch[16];
  __shared__ int err_s;
  constexpr int NW = kKeepNT / 64;
  constexpr int PB = V > 0 ? 16 * V : 1;                           // positions (mask bytes) a thread holds in registers
  const int tid = threadIdx.x;
  const int64_t K = K_dev ? min(K_dev[0], K_max) : K_max;
  if (tid == 0) err_s = 0;
  const int64_t E = V > 0 ? PB : (S + kKeepNT - 1) / kKeepNT;
  const int64_t b = min(S, tid * E), e = min(S, b + E);
  union Bytes { uint4 q[V > 0 ? V : 1]; uint8_t c[PB]; };
  Bytes vm, vs;
  if constexpr (V > 0) {
    // (the masks are torch tensors: 16-byte aligned; a vector that crosses the end is fetched byte by byte)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int64_t p0 = b + 16 * j;
      if (p0 + 16 <= S && (reinterpret_cast<uintptr_t>(video_mask + p0) & 15) == 0 &&
          (!visual_mask || (reinterpret_cast<uintptr_t>(visual_mask + p0) & 15) == 0)) {
        vm.q[j] = *reinterpret_cast<const uint4*>(video_mask + p0);
        vs.q[j] = visual_mask ? *reinterpret_cast<const uint4*>(visual_mask + p0) : make_uint4(0u, 0u, 0u, 0u);
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const bool in = p0 + u < S;
          vm.c[16 * j + u] = in ? video_mask[p0 + u] : uint8_t(0);
          vs.c[16 * j + u] = (in && visual_mask) ? visual_mask[p0 + u] : uint8_t(0);
        }
      }
    }
  }
  // f(position, is video, is visual) over my positions in order
  auto walk = [&](auto f) {
    if constexpr (V > 0) {                                         // (bytes picked out of the packed words: 8 V registers, not 32 V)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const uint32_t wm[4] = {vm.q[j].x, vm.q[j].y, vm.q[j].z, vm.q[j].w}, wv[4] = {vs.q[j].x, vs.q[j].y, vs.q[j].z, vs.q[j].w};
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (b + 16 * j + u < e) f(b + 16 * j + u, ((wm[u >> 2] >> (8 * (u & 3))) & 0xFFu) != 0u, ((wv[u >> 2] >> (8 * (u & 3))) & 0xFFu) != 0u);
      }
    } else {
      keep_walk(video_mask, visual_mask, b, e, f);
    }
  };
  uint32_t nv = 0;
  walk([&](int64_t, bool vid, bool) { nv += vid ? 1u : 0u; });
  if (use_bitmap) for (int64_t w = tid; w < (S + 31) / 32; w += kKeepNT) bitmap[w] = 0u;
  uint32_t tot;
  const int64_t ord0 = block_excl_scan<NW>(nv, xch, tot);        // video ordinal of my first video token
  __syncthreads();
  {   // kept[] must be strictly ascending ordinals of video positions: anything else would leave holes in keep_out
    int bad = 0;
    for (int64_t i = tid; i < K; i += kKeepNT) {
      const int64_t v = kept[i];
      if (v < 0 || v >= int64_t(tot) || (i + 1 < K && kept[i + 1] <= v)) bad = 1;
      else if (use_bitmap) atomicOr(&bitmap[v >> 5], 1u << (v & 31));
    }
    if (bad) atomicOr(&err_s, 4);
  }
  __syncthreads();
  int64_t q = 0;                                                 // (no bitmap) first entry of kept[] that is >= ord0
  if (!use_bitmap) { int64_t lo = 0, hi = K; while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (kept[m] < ord0) lo = m + 1; else hi = m; } q = lo; }
  // is video token number o kept?  (qq: the thread's cursor into kept[] when there is no bitmap)
  auto kept_has = [&](int64_t o, int64_t& qq) -> bool {
    if (use_bitmap) return (bitmap[o >> 5] >> (o & 31)) & 1u;
    while (qq < K && kept[qq] < o) ++qq;
    return qq < K && kept[qq] == o;
  };
  // pass 1: how many positions / visual rows do I keep, how many visual positions precede mine; the decisions stay in a
  // register mask when the positions do
  uint32_t nk = 0, nvis = 0, nvk = 0;
  unsigned long long keepbits[(PB + 63) / 64] = {};
  {
    int64_t o = ord0, qq = q;
    int u = 0;
    walk([&](int64_t, bool vid, bool vis) {
      bool keep = true;
      if (vid) { keep = kept_has(o, qq); ++o; }
      if (V > 0) { if (keep) keepbits[u >> 6] |= 1ull << (u & 63); ++u; }
      nk += keep ? 1u : 0u;
      if (vis) { ++nvis; nvk += keep ? 1u : 0u; }
    });
  }
  const int64_t k0 = block_excl_scan<NW>(nk, xch, tot);
  const uint32_t tot_k = tot;
  __syncthreads();
  int64_t v0 = 0, vk0 = 0;
  uint32_t tot_vk = 0;
  if (visual_mask) {                                             // (uniform)
    v0 = block_excl_scan<NW>(nvis, xch, tot);
    __syncthreads();
    vk0 = block_excl_scan<NW>(nvk, xch, tot);
    tot_vk = tot;
    __syncthreads();
  }
  if (keep_out) for (int64_t i = int64_t(tot_k) + tid; i < keep_cap; i += kKeepNT) keep_out[i] = -1;
  {
    int64_t o = ord0, qq = q, kk = k0, vv = v0, vk = vk0;
    int u = 0;
    walk([&](int64_t p, bool vid, bool vis) {
      bool keep = true;
      if (V > 0) { keep = (keepbits[u >> 6] >> (u & 63)) & 1ull; ++u; }
      else if (vid) { keep = kept_has(o, qq); ++o; }
      if (keep && keep_out) { if (kk < keep_cap) keep_out[kk] = p; ++kk; }
      if (vis) { if (keep && vis_rows_out) { if (vk < vis_cap) vis_rows_out[vk] = vv; ++vk; } ++vv; }
    });
  }
  if (tid == 0 && counts_out) {
    // counts_out[2]: 1 = more kept positions than keep_cap, 2 = more visual rows than vis_cap (neither is written
    // past its capacity), 4 = kept[] not strictly ascending inside [0, video positions), 8 = fewer positions than
    // keep_cap (the caller's count of video positions was wrong: the rest of keep_out is filled with -1)
    const long long err = err_s | (int64_t(tot_k) > keep_cap ? 1 : 0) | ((vis_rows_out && int64_t(tot_vk) > vis_cap) ? 2 : 0) |
                          ((keep_out && int64_t(tot_k) < keep_cap) ? 8 : 0);
    long long* co = reinterpret_cast<long long*>(counts_out);
    __hip_atomic_store(co, (long long)tot_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(co + 1, (long long)tot_vk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(co + 2, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // (last: what a spinning host waits for)
  }
}

// ---- known-answer-test kernels -------------------------------------------------------
template <int DT>
__global__ void k_kat_exp(const void* __restrict__ in, int64_t n, void* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) stT<DT>(out, i, float(exp(double(ldT<DT>(in, i)))));
}
template <int DT>
__global__ void k_kat_round(const float* __restrict__ in, int64_t n, void* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) stT<DT>(out, i, in[i]);
}

// ======================================================================================
// host side: workspace plan + launchers
// ======================================================================================
#ifndef VC2_DEV_ONLY      // (host side: plans and launches)
constexpr int64_t kMaxFramesTotal = 65536;   // frames of the WHOLE video in the frame-sharded path

struct Plan {
  int64_t F, N, D, R;
  int dt, ES, VEC, CV, TPB;     // VEC actually used (1 = scalar fallback), column vectors, threads
  int G, rows_per_group;        // sweep-1 row groups (G = F * stat_splits), stat blocks
  int stat_splits, NB, BF;      // groups per frame; stat blocks of BF (<= kStatBlockFrames) frames
  int64_t F_total;              // frames of the WHOLE video (frame-sharded pass: canonical blockings depend on it)
  int ord_S, ord_m;             // sweep 2, ORD form (k_norm_colsum2<.., ORD>): frames [0, ord_nA) are swept by ord_S - 1 workgroups each, the
  int ord_nA;                   //   others by ord_S (ord_nA = 0: all by ord_S) -- OrdGeo; ord_m != 0: the plan has an ORD geometry
  int S, S_q, S_W;              // sweep 2: S_W chunks of S_q consecutive rows (frame boundaries inside a chunk cut it in
                                //   segments); a frame's segments fill its first slots of S in `part`
  int S2, rows_per_split2;      // sweep-3 splits per frame (<= kDistMaxRows rows each)
  int skew2_q10;                // how much the first split of a frame exceeds the mean, in 1/1024 (k_dist)
  // workspace offsets (bytes)
  size_t o_part_stats, o_stats, o_bstats, o_var_f32, o_var_T, o_mask, o_cols, o_order, o_opos, o_spos, o_perm, o_den, o_part_col, o_fc, o_csum, o_csum_part, o_vc,
      o_rflag, o_vpart, o_total, o_s, o_zbuf, o_scales_f32, o_scales_T, o_offs, o_ticket, o_nfixlist, o_corr, o_vscratch, o_vticket, o_dmin, o_fmark, o_tmp_f32, o_rlist, o_bsum, total_bytes;
  int vstride;
};

// non-zero: replay torch's fp32 accumulation order for boundary-fragile tokens (half precision) so that the result is
// bit-exact to the CPU reference -- 4 (DEFAULT since round 4: the frame-mean margin grows under cancellation), 1 (the
// faster empirical margins: opt-in), 3 (proven centre margins), 2 (debug); 0: plain correctly-rounded-op semantics.
// See vc2_set_mode.
// A process-wide setting (vc2_set_mode) that a thread may override for itself (vc2_set_thread_mode): a pass issued from
// a worker thread (HF generate's streaming thread) follows what the application set, and two threads that need
// different modes cannot disturb each other.
std::atomic<int> g_mode_default{4};
thread_local int g_mode_thread = -1;             // -1: follow the process-wide setting
inline int cur_mode() { return g_mode_thread >= 0 ? g_mode_thread : g_mode_default.load(std::memory_order_relaxed); }

int make_plan(int64_t F, int64_t N, int64_t D, int dt, Plan* p, int64_t F_total = 0, int block_frames = 0) {
  if (F <= 0 || N <= 0 || D <= 0) return fail(VC2_ERR_ARG, "F, N, D must be positive (got %lld, %lld, %lld)",
                                              (long long)F, (long long)N, (long long)D);
  if (dt < 0 || dt > 2) return fail(VC2_ERR_ARG, "unknown dtype code %d", dt);
  p->F = F; p->N = N; p->D = D; p->R = F * N; p->dt = dt;
  p->ES = dt == VC2_F32 ? 4 : 2;
  const int full = dt == VC2_F32 ? 4 : 8;
  p->VEC = (D % full == 0) ? full : 1;
  p->CV = int(D / p->VEC);
  if (p->CV > 1024)
    return fail(VC2_ERR_UNSUPPORTED, "D=%lld needs %d column vectors per row; at most 1024 are supported "
                "(D <= 8192 for 16-bit, <= 4096 for fp32, D %% %d == 0)", (long long)D, p->CV, full);
  p->TPB = int(cdiv(p->CV, 64) * 64);
  p->F_total = F_total > 0 ? F_total : F;
  // sweep-1 groups: `stat_splits` pieces per frame -- a function of the WHOLE video's frame count, so that every
  // rank of a frame-sharded pass cuts its frames exactly like the unsharded pass does
  static const int env_groups = [] { const char* e = getenv("VC2_STAT_GROUPS"); return e ? atoi(e) : 128; }();   // (experiments: row groups aimed at)
  p->stat_splits = int(std::max<int64_t>(1, std::min<int64_t>(cdiv(env_groups, p->F_total), std::max<int64_t>(1, N / 32))));
  p->rows_per_group = int(cdiv(N, p->stat_splits));
  p->stat_splits = int(cdiv(N, p->rows_per_group));
  p->G = int(F * p->stat_splits);
  p->BF = block_frames > 0 ? std::min(block_frames, kStatBlockFrames) : kStatBlockFrames;
  p->NB = int(cdiv(F, p->BF));
  // sweep 2 runs as ONE wave of workgroups: 2 per CU fit (LDS row buffers, VGPRs) = 512 slots on 256 CUs, and every
  // workgroup lives for the whole sweep.  The ORDER riders ("torch order" mode, 16-bit inputs) are workgroups of the
  // same launch, so the splits are chosen to leave them slots -- with 512 + 16 workgroups the last 16 start when the
  // riders end and the sweep takes 51 instead of 36 us; with 384 + 16 it takes 41.
  const int64_t riders = (cur_mode() && dt != VC2_F32) ? rider_parts_max(p->R, D * p->ES) : 0;
  // S is chosen from THIS rank's frames (occupancy), not from the whole video's: the frame sums are fp64 sums of
  // T-rounded x^ in [-1, 1] -- exact (so independent of how the rows are cut) for fp16 by range (2^-24 .. 1, <= 8192
  // rows: 48 bits), and for bf16 unless a nonzero |x^| < 2^-39 meets a frame sum > 2^6
  {
    // Equal CHUNKS of consecutive rows, not equal pieces of frames: with 448 slots and 128 frames of 196 rows the
    // frame-aligned cut was 4 x 49 rows for half of the frames and 3 x 66 for the others -- the sweep took as long as
    // the 66-row workgroups (36 us against 25 for the others: scripts/dbg_wg.py).  A chunk that contains a frame
    // boundary is swept segment by segment (one flush of the column sums per segment).
    // (the streamlined sweep built for three workgroups per CU -- VC2_S2_WAVES = 3 -- has 768 slots: 512 streaming
    // workgroups AND the riders are resident together; decided from what the plan knows, shape and mode)
    static const int env_budget = [] { const char* e = getenv("VC2_S2_BUDGET"); return e ? atoi(e) : 0; }();
    const bool three = VC2_S2_WAVES >= 3 && cur_mode() != 0 && dt != VC2_F32 && (D == 1024 || D == 3584 || D == 4096);
    const int64_t budget = env_budget > 0 ? env_budget : (three ? 512 : 512 - riders);
    int64_t W = std::max<int64_t>(1, std::min<int64_t>(budget, p->R / (8 * kRowWaves)));   // >= 32 rows a chunk
    int64_t q = std::max<int64_t>(cdiv(p->R, W), cdiv(N, 6));          // (a frame meets at most (N - 1) / q + 2 <= 8 chunks)
    p->S_q = int(q);
    p->S_W = int(cdiv(p->R, q));
    p->S = int(std::min<int64_t>(8, (N - 1) / q + 2));
    // the ORD form (torch-ordered frame sums, see k_norm_colsum2 / OrdGeo): frames of <= 512 tokens, the streamlined sweep's
    // shapes, every frame workgroup resident next to the riders.  S - 1 workgroups for the first nA frames, S for the
    // others, as many as the slots allow (a workgroup needs at least one block; `part` holds 8 pieces per frame)
    p->ord_S = p->ord_m = p->ord_nA = 0;
    if (cur_mode() != 0 && dt != VC2_F32 && (D == 1024 || D == 3584 || D == 4096) && N <= 512) {
      const int64_t nblk = std::max<int64_t>(1, N / 16);
      const int64_t smax = std::min<int64_t>(8, nblk), slots = 512 - riders;
      const int64_t s_lo = std::min<int64_t>(smax, slots / F);
      if (s_lo >= 1) {
        const int64_t more = s_lo < smax ? std::min<int64_t>(F, slots - F * s_lo) : 0;     // frames that take s_lo + 1
        if (more > 0) { p->ord_S = int(s_lo + 1); p->ord_nA = int(F - more); }
        else { p->ord_S = int(s_lo); p->ord_nA = 0; }
        p->ord_m = 1;
      }
    }
  }
  {
    // sweep 3: one workgroup per (frame, split); ~1024 workgroups when the video allows, and at most 25 rows each so
    // that the 10 exp per token of the fused epilogue are ONE round over the workgroup's 256 threads
#ifndef VC2_DIST_WGS
#define VC2_DIST_WGS 1024
#endif
#ifndef VC2_DIST_RPS_MIN
#define VC2_DIST_RPS_MIN 16
#endif
    int64_t rps = std::max<int64_t>(VC2_DIST_RPS_MIN, std::min<int64_t>(25, cdiv(p->R, VC2_DIST_WGS)));
    static const int env_rps = [] { const char* e = getenv("VC2_DIST_RPS"); return e ? atoi(e) : 0; }();   // (experiments)
    if (env_rps > 0) rps = std::min<int64_t>(env_rps, kDistMaxRows);
    rps = std::min<int64_t>(rps, N);
    p->S2 = int(cdiv(N, rps));
#ifndef VC2_DIST_SKEW
#define VC2_DIST_SKEW 138
#endif
    // the skew pays when the launch is ONE wave of workgroups with several of them per CU (4 at 1024 on 256 CUs)
    const int64_t wgs = F * p->S2;
    static const int env_skew = [] { const char* e = getenv("VC2_DIST_SKEW_Q10"); return e ? atoi(e) : -1; }();
    p->skew2_q10 = env_skew >= 0 ? env_skew : (p->S2 >= 4 && wgs > 512 && wgs <= 1024) ? int(VC2_DIST_SKEW * (wgs - 256) / 768) : 0;
    p->rows_per_split2 = dist_split_cut(1, p->S2, int(N), p->skew2_q10);          // the longest split
  }
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
  p->o_part_stats = take(size_t(p->G) * 2 * D * 8);
  p->o_stats = take(size_t(2) * D * 8);
  p->o_bstats = take(size_t(F) * 2 * D * 8);                  // (room for blocks of one frame)
  p->o_var_f32 = take(size_t(D) * 4);
  p->o_var_T = take(size_t(D) * 4);
  p->o_mask = take(size_t(D));
  p->o_cols = take(size_t(D) * 4);
  p->o_order = take(size_t(D) * 4);
  p->o_opos = take(size_t(D) * 4);
  p->o_spos = take(size_t(D) * 4);
  p->o_perm = take(size_t(D) * 4);
  p->o_den = take(size_t(p->R) * 4);
  p->o_part_col = take(size_t(F) * 8 * D * 8);                // (S <= 8 whatever the mode: one workspace size per shape)
  p->o_fc = take(size_t(F) * D * 4);
  p->o_csum = take(size_t(D) * 8);
  p->o_csum_part = take(size_t(2) * size_t(cdiv(F, kCentreFL)) * D * 8);   // [sums | bounds of sum |x^|][groups][C]
  p->o_vc = take(size_t(D) * 4);
  p->o_rflag = take(size_t(p->R));
  p->o_vpart = take(size_t(F) * p->S2 * 8);
  p->o_total = take(size_t(p->R) * 4);
  p->o_s = take(size_t(F) * 4);
  p->o_zbuf = take(size_t(std::max<int64_t>(F, kMaxFramesTotal)) * 4);       // whole-video frame count
  p->o_scales_f32 = take(size_t(std::max<int64_t>(F, kMaxFramesTotal)) * 4);  // (frame-sharded case)
  p->o_scales_T = take(size_t(F) * 4);
  p->o_offs = take(size_t(F + 1) * 8);
  p->o_ticket = take(256);                                    // 64 ints (kTk*)
  p->o_nfixlist = take(size_t(p->R) * 8 + size_t(cdiv(F, 2)) * 8);   // 8-byte queue granules (fixq_pack); behind them, zeroed with them:
  p->o_fmark = p->o_nfixlist + size_t(p->R) * 8;              //   per frame: "all its means are on the replay list" (FixPush)
  p->o_corr = take(size_t(p->R) * sizeof(NormCorr));
  {
    const int lpv = cascade_lp(p->R);           // level-1 groups of a video-centre column (k_video_centre's scratch)
    p->vstride = cascade_modelled(p->R) ? int(cdiv(p->R >> lpv, int64_t(1) << lpv) + 1) : 1;
  }
  p->o_vscratch = take(size_t(D) * p->vstride * 4);
  p->o_vticket = take(size_t(D) * 4);
  p->o_dmin = take(size_t(F) * 4);                 // per frame: the smallest denominator (centre-mean margins)
  p->o_tmp_f32 = take(size_t(std::max<int64_t>(p->R, D)) * 4);
  p->o_bsum = take(p->ord_m > 0 ? size_t(F) * size_t(N / 16 + 1) * size_t(D / 2) * 4 : 0);   // (ORD) frame block sums, fp32
  p->o_rlist = take(size_t(2) * F * D * 4);        // frame means to replay: frame * C + column (k_frame_centres / fix riders -> frame_replay_wave)
  p->total_bytes = o;
  return VC2_OK;
}

template <typename T> T* wsp(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VC2_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return VC2_OK;
}

// ---- optional per-kernel timing (bench.py's roofline leg): hipEvents around every launch --------
enum KernelId { KID_STATS = 0, KID_STATS_REDUCE, KID_CHAN_SELECT, KID_NORM_COLSUM, KID_CENTRES, KID_DIST,
                KID_EPILOGUE, KID_SCALES, KID_KS, KID_SELECT, KID_GATHER_ROWS, KID_OTHER, KID_CHAN_ORDER, KID_DIST_FIX,
                KID_CENTRE_FIX, KID_COUNT };
const char* const kKernelNames[KID_COUNT] = {"k_chan_stats", "k_stats_reduce", "k_chan_select", "k_norm_colsum",
                                             "k_centres", "k_dist", "k_token_epilogue", "k_scales", "k_ks(unused)",
                                             "k_select", "k_gather_rows", "k_norm_fix", "k_chan_order",
                                             "k_dist_fix", "k_centre_fix"};
struct ProfRec { int id; hipEvent_t a, b; };
bool g_prof = false;                // (bench-only; the record list is mutex-protected)
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
double g_prof_ms[KID_COUNT];
int64_t g_prof_n[KID_COUNT];

struct ProfScope {
  int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_) {
    if (g_prof) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
  }
  ~ProfScope() {
    if (g_prof && a) { (void)hipEventRecord(b, st); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_recs.push_back({id, a, b}); }
  }
};

#define VC2_DISPATCH_DT(dt, ...)                                    \
  switch (dt) {                                                     \
    case VC2_F32: { constexpr int DT = VC2_F32; __VA_ARGS__; } break;   \
    case VC2_BF16: { constexpr int DT = VC2_BF16; __VA_ARGS__; } break; \
    default: { constexpr int DT = VC2_F16; __VA_ARGS__; } break;        \
  }
// VEC is either the dtype's 16-byte vector width or 1
#define VC2_DISPATCH_VEC(p, ...)                                            \
  VC2_DISPATCH_DT((p).dt, if ((p).VEC == 1) { constexpr int VEC = 1; __VA_ARGS__; } \
                  else { constexpr int VEC = Tr<DT>::VEC; __VA_ARGS__; })

int need_ws(const Plan& p, void* ws, size_t ws_bytes) {
  if (!ws) return fail(VC2_ERR_ARG, "workspace pointer is null");
  if (ws_bytes < p.total_bytes)
    return fail(VC2_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, p.total_bytes);
  return VC2_OK;
}

// sweep 1 -> (mean, M2) stats and/or var
// sweep 1 -> per stat block (mean, M2) in bstats[NB][2][D] (ws when bstats == nullptr), and -- var_f32 / var_T --
// the variance of THESE rows reduced from them (single-rank case)
// sweep 1 alone: the per-group partials of x into ws (pool.xin != null: x is produced here, see k_chan_stats)
int launch_stats_sweep(const Plan& p, const void* x, void* ws, const PoolSrc& pool, hipStream_t st, uint32_t* var_reset = nullptr) {
  double* part = wsp<double>(ws, p.o_part_stats);
  if (p.G > 65535) return fail(VC2_ERR_UNSUPPORTED, "too many sweep-1 row groups (%d)", p.G);
  dim3 grid(unsigned(cdiv(p.CV, 64)), unsigned(p.G));
  ProfScope ps_(KID_STATS, st);
  if (pool.xin) {
    VC2_DISPATCH_VEC(p, hipLaunchKernelGGL((k_chan_stats<DT, VEC, 8, 1>), grid, dim3(kStatsWaves * 64), 0, st, x,
                                            p.R, int(p.D), p.CV, int(p.N), p.stat_splits, p.rows_per_group, p.BF, part,
                                            pool));
  } else {
    VC2_DISPATCH_VEC(p, hipLaunchKernelGGL((k_chan_stats<DT, VEC, 8, 0>), grid, dim3(kStatsWaves * 64), 0, st, x,
                                            p.R, int(p.D), p.CV, int(p.N), p.stat_splits, p.rows_per_group, p.BF, part,
                                            pool, var_reset));
  }
  return check_launch("chan_stats");
}

int launch_chan_stats(const Plan& p, const void* x, void* ws, double* bstats, void* var_T, float* var_f32,
                      hipStream_t st, bool zero_queue_counters = false, bool have_partials = false,
                      int64_t* kstatus = nullptr) {
  double* part = wsp<double>(ws, p.o_part_stats);
  if (!have_partials) { int rcs = launch_stats_sweep(p, x, ws, PoolSrc{}, st); if (rcs) return rcs; }
  { ProfScope ps_(KID_STATS_REDUCE, st);
  const PartSrc src{part, x, p.R, p.BF * p.stat_splits, p.G, int(p.N), p.BF};
  if (bstats)         // the frame-sharded pass needs the block statistics themselves (exchange 1)
    VC2_DISPATCH_DT(p.dt, hipLaunchKernelGGL((k_block_stats<DT>), dim3(unsigned(cdiv(p.D, 64)), unsigned(p.NB)), dim3(64),
                                             0, st, src, int(p.D), bstats));
  if (var_f32 || var_T) {
    const int64_t n_each = int64_t(p.BF) * p.N, n_last = p.R - int64_t(p.NB - 1) * n_each;
    // narrow column slabs (16 columns x 16 lanes: 224 workgroups at D = 3584) spread the loads of sweep 1's partials
    // (7 MB) over the whole chip instead of 56 CUs; VC2_VAR_CW = 64 is the wide form (experiments)
    static const int cw_env = [] { const char* e = getenv("VC2_VAR_CW"); return e ? atoi(e) : 16; }();
    auto go = [&](auto dt_tag, auto cw_tag) {
      constexpr int DT = decltype(dt_tag)::value, CW = decltype(cw_tag)::value;
      hipLaunchKernelGGL((k_var_from_stats<DT, CW>), dim3(unsigned(cdiv(p.D, CW))), dim3(CW * kRedGL), 0, st,
                         (const double*)nullptr, p.NB, n_each, n_last, int(p.D), var_T, var_f32,
                         zero_queue_counters ? wsp<int>(ws, p.o_ticket) : (int*)nullptr, src,
                         make_fold_tab(p.NB, n_each, n_last),
                         zero_queue_counters ? wsp<unsigned long long>(ws, p.o_nfixlist) : nullptr,
                         int(p.R + cdiv(p.F, 2)), reinterpret_cast<unsigned long long*>(kstatus));
    };
    VC2_DISPATCH_DT(p.dt, (cw_env == 64 ? go(std::integral_constant<int, DT>{}, std::integral_constant<int, 64>{})
                           : cw_env == 32 ? go(std::integral_constant<int, DT>{}, std::integral_constant<int, 32>{})
                                          : go(std::integral_constant<int, DT>{}, std::integral_constant<int, 16>{})));
  } }
  return check_launch("chan_stats");
}

// Opt a kernel into > 48 KiB of dynamic LDS -- once per (kernel, device), not per launch.
std::mutex g_attr_mu;
std::set<std::pair<const void*, int>> g_attr_done;
// (static_lds: the kernel's own __shared__ arrays -- the dynamic part can only grow to what they leave)
template <typename K> int allow_big_lds(K kernel, size_t smem, const char* what, size_t static_lds = 0) {
  if (smem <= 48 * 1024 - static_lds) return VC2_OK;
  if (smem > 160 * 1024 - 256 - static_lds) return fail(VC2_ERR_UNSUPPORTED, "%s needs %zu bytes of LDS (D too large)", what, smem);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::pair<const void*, int> key(reinterpret_cast<const void*>(kernel), dev);
  std::lock_guard<std::mutex> lk(g_attr_mu);
  if (g_attr_done.count(key)) return VC2_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     int(160 * 1024 - 256 - static_lds));
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(VC2_ERR_LAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e)); }
  g_attr_done.insert(key);
  return VC2_OK;
}

int launch_chan_select(const float* var_f32, int64_t D, int64_t k, uint8_t* mask, int* cols, int* perm,
                       hipStream_t st, uint32_t* wperm = nullptr, uint32_t* wcpos = nullptr, int* status = nullptr,
                       bool words64_expected = false) {
  if (k <= 0 || k > D) return fail(VC2_ERR_UNSUPPORTED, "channel count k=%lld out of range for D=%lld",
                                   (long long)k, (long long)D);
  const size_t smem = chan_select_lds(int(D));
  // VC2_SEL3=1 (experiment, round 6; OFF by default): D <= 4096 on four waves with the array in registers (k_chan_select3).
  // Bit-identical (409 selection / full-pass tests green with it) and SLOWER: 26.1 against 20.9 us at D = 3584 -- its
  // rounds are ~70 instructions per 64-element row with VALU -> SGPR -> VALU dependencies that one wave per SIMD cannot hide
  // (4.2 / 3.0 / 2.3 / 2.4 us for the rounds the 16-wave LDS form does in 2.0 each: profiles/r06_b_sel3_register_rounds.csv)
  const int sel3_env = [] { const char* e = getenv("VC2_SEL3"); return e ? atoi(e) : 0; }();     // (read per call: the test-suite
  //                                                                                                   compares the forms in one process)
  // VC2_SEL4 (default 1): D <= 4096, sixteen waves, thread-contiguous registers (k_chan_select4); 0: the LDS rounds
  const int sel4_env = [] { const char* e = getenv("VC2_SEL4"); return e ? atoi(e) : 1; }();
  if (sel4_env != 0 && sel3_env == 0 && D <= 4096 && !words64_expected) {
    { int rca = allow_big_lds(&k_chan_select4, smem, "k_chan_select4"); if (rca) return rca; }
    ProfScope ps_(KID_CHAN_SELECT, st);
    hipLaunchKernelGGL(k_chan_select4, dim3(1), dim3(kSelNT), smem, st, var_f32, int(D), int(k), mask, cols, perm, wperm, wcpos, status);
    return check_launch("chan_select4");
  }
  if (sel3_env != 0 && D <= 4096 && !words64_expected) {
    ProfScope ps_(KID_CHAN_SELECT, st);
    { int rca = allow_big_lds(&k_chan_select3<16>, smem, "k_chan_select3"); if (rca) return rca; }
    hipLaunchKernelGGL(k_chan_select3<16>, dim3(1), dim3(kSel3NT), smem, st, var_f32, int(D), int(k), mask, cols, perm, wperm, wcpos, status);
    return check_launch("chan_select3");
  }
  { int rca = allow_big_lds(&k_chan_select, smem, "k_chan_select"); if (rca) return rca; }
  { ProfScope ps_(KID_CHAN_SELECT, st);
  hipLaunchKernelGGL(k_chan_select, dim3(1), dim3(kSelNT), smem, st, var_f32, int(D), int(k), mask, cols, perm, wperm,
                     wcpos, status); }
  return check_launch("chan_select");
}
// torch.topk's ORDER of the kept channels as its own kernel (the scoring entry points attach it to sweep 2 instead)
int launch_chan_order(const OrderArgs& oa, hipStream_t st) {
  const size_t smem = chan_order_lds(oa.D, oa.k, 8);
  { int rca = allow_big_lds(&k_chan_order, smem, "k_chan_order"); if (rca) return rca; }
  { ProfScope ps_(KID_CHAN_ORDER, st);
  hipLaunchKernelGGL(k_chan_order, dim3(unsigned(order_parts(oa.k, VC2_ORDER_PARTS))), dim3(kOrdNT), smem, st, oa); }
  return check_launch("chan_order");
}


// the scored channels: ascending list (nullptr = all D), the same channels in torch.topk's order and their
// positions in `cols` (both nullptr = identity when cols is nullptr, else strict mode is unavailable)
struct ChanSet {
  const int* cols;
  const int* spos;     // position of cols[i] in torch.topk's order (nullptr with cols == nullptr: identity)
  int C;
  int strict;
};
inline ChanSet make_chanset(const Plan& p, const int* cols, const int* spos, int64_t C) {
  ChanSet cs{cols, spos, int(C), 0};
  cs.strict = (cur_mode() && p.ES == 2 && (cols == nullptr || spos)) ? cur_mode() : 0;   // 2 = replay always
  return cs;
}

// ACC policy of the two scoring sweeps: fp32 accumulation + wider replay margins in "torch order" mode (half
// precision only), fp64 otherwise
inline bool fast_acc(const Plan& p, const ChanSet& cs) { return cs.strict != 0 && p.dt != VC2_F32; }

template <int DT, int VEC, int NPLB, int ACC>
int launch_norm_acc(const Plan& p, const void* x, const ChanSet& cs, void* ws, const OrderArgs& rider, hipStream_t st) {
  const int* cols = cs.cols; const int C = cs.C;
#ifdef VC2_NO_RIDER
  constexpr int RIDER = 0;
#else
  constexpr int RIDER = (ACC == 1 && VEC > 1) ? 1 : 0;       // the kernels that can carry the ORDER rider
#endif
  if (rider.perm && !RIDER) return fail(VC2_ERR_UNSUPPORTED, "internal: ORDER rider on a sweep without one");
  size_t smem = std::max<size_t>(2 * kRowWaves * row_lds_bytes(int(p.D), Tr<DT>::ES),
                                 size_t(kRowWaves) * NPLB * 64 * 8);               // row buffers, then the combine
  if (rider.perm) smem = std::max(smem, chan_order_lds(rider.D, rider.k, 4));      // the rider workgroup's arrays
  int rc = allow_big_lds(&k_norm_colsum<DT, VEC, NPLB, ACC, RIDER>, smem, "k_norm_colsum");
  if (rc) return rc;
  hipLaunchKernelGGL((k_norm_colsum<DT, VEC, NPLB, ACC, RIDER>), dim3(unsigned(p.S_W + (rider.perm ? (rider.parts & 0xFF) : 0))),
                     dim3(kRowWaves * 64), smem, st, x,
                     int(p.N), int(p.D), p.CV, C, cols, cs.strict, p.S, p.S_q, p.R,
                     wsp<float>(ws, p.o_den), wsp<double>(ws, p.o_part_col), wsp<int>(ws, p.o_ticket),
                     wsp<unsigned long long>(ws, p.o_nfixlist), int(p.R), wsp<uint8_t>(ws, p.o_rflag), rider);
  return VC2_OK;
}
// the streamlined sweep 2 (k_norm_colsum2): 16-bit rows of NCH full 1 KiB chunks, C = D / 2, "torch order" mode
#ifndef VC2_S2_V2
#define VC2_S2_V2 1
#endif
std::atomic<int> g_s2v2{-1};            // -1: not read yet (environment VC2_S2_V2, default: on)
inline bool s2v2_on() {
  int v = g_s2v2.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("VC2_S2_V2"); v = e ? (atoi(e) != 0) : VC2_S2_V2; g_s2v2.store(v, std::memory_order_relaxed); }
  return v != 0;
}
std::atomic<int> g_s3v2{-1};            // the streamlined sweep 3 (k_dist<.., V2 = 1>): environment VC2_S3_V2; default OFF -- measured
                                        // no faster than the general form (35.4 vs 34.6 us bf16, 47 vs 45 fp16: NOTES_r05.md)
inline bool s3v2_on() {
  int v = g_s3v2.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("VC2_S3_V2"); v = e ? (atoi(e) != 0) : 0; g_s3v2.store(v, std::memory_order_relaxed); }
  return v != 0;
}
inline int s2v2_nch(const Plan& p, const ChanSet& cs) {
  if (!s2v2_on() || p.dt == VC2_F32 || p.VEC == 1 || !cs.cols || !fast_acc(p, cs)) return 0;
  if (p.D % 512 != 0 || int64_t(cs.C) * 2 != p.D || p.R >= (int64_t(1) << 31) || p.F * int64_t(p.S) >= (int64_t(1) << 31) / 8) return 0;
  const int nch = int(p.D / 512);
  return (nch == 2 || nch == 7 || nch == 8) ? nch : 0;
}
// the ORD form of sweep 2 + centre kernels (torch-ordered frame sums): when the plan has a geometry for it and the
// streamlined sweep applies.  OPT-IN (environment VC2_S2_ORD=1): bit-identical results (the whole parity suite passes
// with it, no frame mean is ever replayed), but a wave must sweep whole 16-row blocks, which leaves 384 workgroups of
// 64 / 68 rows where the row-interleaved form has 448 of 56 -- the busiest CU streams 132 rows instead of 112: sweep 2
// 30.6 -> 34.7 us at the target shape, more than the centre kernels win there (fp16: -14 us in k_video_centre, +9 in the
// sweep).  NOTES_r05.md.
#ifndef VC2_S2_ORD_DEFAULT
#define VC2_S2_ORD_DEFAULT 1
#endif
inline bool ord_on(const Plan& p, const ChanSet& cs) {
  const char* e = getenv("VC2_S2_ORD");                          // (read per pass: the test-suite runs both forms in one process)
  return (e ? atoi(e) : VC2_S2_ORD_DEFAULT) != 0 && p.ord_m > 0 && s2v2_nch(p, cs) != 0;
}
template <int DT, int NCH>
int launch_norm_v2(const Plan& p, const void* x, const ChanSet& cs, void* ws, const OrderArgs& rider, hipStream_t st) {
  size_t smem = s2v2_lds(NCH);                                                     // row buffers + the combine's scratch
  if (rider.perm) smem = std::max(smem, chan_order_lds(rider.D, rider.k, 4));      // the rider workgroup's arrays
  const unsigned nr = rider.perm ? unsigned(rider.parts & 0xFF) : 0u;
  if (ord_on(p, cs)) {
    int rc = allow_big_lds(&k_norm_colsum2<DT, NCH, 1, 1>, smem, "k_norm_colsum2");
    if (rc) return rc;
    hipLaunchKernelGGL((k_norm_colsum2<DT, NCH, 1, 1>), dim3(unsigned(ord_total_wgs(OrdGeo{p.ord_S, p.ord_nA}, int(p.F))) + nr),
                       dim3(kRowWaves * 64), smem, st, x, int(p.N), cs.cols, cs.strict, p.ord_S, p.ord_nA, p.R,
                       wsp<float>(ws, p.o_den), wsp<double>(ws, p.o_part_col), wsp<int>(ws, p.o_ticket),
                       wsp<unsigned long long>(ws, p.o_nfixlist), int(p.R), wsp<uint8_t>(ws, p.o_rflag), rider,
                       wsp<float>(ws, p.o_bsum), p.ord_m);
    return VC2_OK;
  }
  int rc = allow_big_lds(&k_norm_colsum2<DT, NCH, 1, 0>, smem, "k_norm_colsum2");
  if (rc) return rc;
  hipLaunchKernelGGL((k_norm_colsum2<DT, NCH, 1, 0>), dim3(unsigned(p.S_W) + nr),
                     dim3(kRowWaves * 64), smem, st, x, int(p.N), cs.cols, cs.strict, p.S, p.S_q, p.R,
                     wsp<float>(ws, p.o_den), wsp<double>(ws, p.o_part_col), wsp<int>(ws, p.o_ticket),
                     wsp<unsigned long long>(ws, p.o_nfixlist), int(p.R), wsp<uint8_t>(ws, p.o_rflag), rider,
                     (float*)nullptr, 0);
  return VC2_OK;
}
template <int DT, int VEC, int NPLB>
int launch_norm_t(const Plan& p, const void* x, const ChanSet& cs, void* ws, const OrderArgs& rider, hipStream_t st) {
  if constexpr (DT != VC2_F32 && VEC > 1) {
    switch (s2v2_nch(p, cs)) {
      case 2: if constexpr (NPLB == 8) return launch_norm_v2<DT, 2>(p, x, cs, ws, rider, st); break;
      case 7: if constexpr (NPLB == 28) return launch_norm_v2<DT, 7>(p, x, cs, ws, rider, st); break;
      case 8: if constexpr (NPLB == 32) return launch_norm_v2<DT, 8>(p, x, cs, ws, rider, st); break;
      default: break;
    }
  }
  if constexpr (DT != VC2_F32) {
    if (fast_acc(p, cs)) return launch_norm_acc<DT, VEC, NPLB, 1>(p, x, cs, ws, rider, st);
  }
  return launch_norm_acc<DT, VEC, NPLB, 0>(p, x, cs, ws, rider, st);
}
template <int DT, int VEC, int NPLB>
int launch_norm_fix_t(const Plan& p, const void* x, const ChanSet& cs, void* ws, hipStream_t st) {
  const size_t smem1 = std::max(row_lds_bytes(int(p.D), Tr<DT>::ES), size_t(cs.C) * 4 + 16);   // row, then C fp32
  int rc1 = allow_big_lds(&k_norm_fix<DT, VEC, NPLB>, smem1, "k_norm_fix");
  if (rc1) return rc1;
  // fp16 queues ~3 % of the rows (its T ulp is 2^13 fp32 ulps) and replays ONE sequential chain per row: more waves
  const int nfix = cs.strict == 2 ? int(std::min<int64_t>(p.R, 4096)) : (p.dt == VC2_F16 ? 2048 : 512);
  hipLaunchKernelGGL((k_norm_fix<DT, VEC, NPLB>), dim3(unsigned(nfix)), dim3(64), smem1, st, x, int(p.D), p.CV, cs.C,
                     cs.cols, cs.spos, wsp<float>(ws, p.o_den), wsp<int>(ws, p.o_ticket) + kTkFixCount,
                     wsp<unsigned long long>(ws, p.o_nfixlist), int(p.R), wsp<int>(ws, p.o_ticket) + kTkCorrCount,
                     wsp<NormCorr>(ws, p.o_corr), int(p.N), FixPush{});
  return VC2_OK;
}
struct DistOut { void* v_T; void* f_T; float* total; };

template <int DT, int VEC, int NPLB, int ACC>
int launch_dist_acc(const Plan& p, const void* x, const ChanSet& cs, void* ws, const DistOut& o, hipStream_t st) {
  const int* cols = cs.cols; const int C = cs.C;
  const size_t area = (std::max(size_t(kRowWaves) * row_lds_bytes(int(p.D), Tr<DT>::ES), size_t(C) * 4 + 16) + 15) / 16 * 16;
  const size_t smem = area + size_t(kDistMaxRows) * (4 + 8 + 40 + 8 + 20 + 1) + 64;
  auto launch = [&](auto kernel) -> int {
    int rc = allow_big_lds(kernel, smem, "k_dist");
    if (rc) return rc;
    ProfScope ps_(KID_DIST, st);
    hipLaunchKernelGGL(kernel, dim3(unsigned(p.F * p.S2)), dim3(kRowWaves * 64), smem, st, x,
                       int(p.N), int(p.D), p.CV, C, cols, cs.spos, cs.strict, p.S2, p.rows_per_split2, p.skew2_q10,
                       wsp<float>(ws, p.o_den),
                       (DT == VC2_F16 && s2v2_nch(p, cs) == 0) ? (uint8_t*)nullptr : wsp<uint8_t>(ws, p.o_rflag),   // (fp16: only k_norm_colsum2 writes it)
                       wsp<float>(ws, p.o_vc),
                       wsp<float>(ws, p.o_fc), o.v_T, o.f_T, o.total, wsp<double>(ws, p.o_vpart));
    return VC2_OK;
  };
  if constexpr (ACC == 1 && NPLB % 4 == 0 && DT != VC2_F32 && VEC > 1 && (NPLB == 8 || NPLB == 28 || NPLB == 32)) {
    if (s3v2_on() && s2v2_nch(p, cs) * 4 == NPLB) return launch(k_dist<DT, VEC, NPLB, ACC, 1, 1>);   // streamlined (sweep 2 wrote rflag)
  }
  if constexpr (ACC == 1 && NPLB % 4 == 0) {
    if (cols && (C & 3) == 0) return launch(k_dist<DT, VEC, NPLB, ACC, 1>);      // tables in 16-byte loads
  }
  return launch(k_dist<DT, VEC, NPLB, ACC, 0>);
}
template <int DT, int VEC, int NPLB>
int launch_dist_t(const Plan& p, const void* x, const ChanSet& cs, void* ws, const DistOut& o, hipStream_t st) {
  if constexpr (DT != VC2_F32) {
    if (fast_acc(p, cs)) return launch_dist_acc<DT, VEC, NPLB, 1>(p, x, cs, ws, o, st);
  }
  return launch_dist_acc<DT, VEC, NPLB, 0>(p, x, cs, ws, o, st);
}
// compact positions per lane -> compile-time bucket (28 = 3584-d, 32 = 4096-d models)
#define VC2_DISPATCH_NPL(npl, FN, ...)                                   \
  ((npl) <= 8 ? FN<DT, VEC, 8>(__VA_ARGS__) : (npl) <= 16 ? FN<DT, VEC, 16>(__VA_ARGS__) \
   : (npl) <= 28 ? FN<DT, VEC, 28>(__VA_ARGS__) : (npl) <= 32 ? FN<DT, VEC, 32>(__VA_ARGS__) \
   : (npl) <= 64 ? FN<DT, VEC, 64>(__VA_ARGS__)                          \
   : fail(VC2_ERR_UNSUPPORTED, "more than 4096 scored channels"))

// sweep 2 + centres.  single_rank: also the video centre; else only the rank's csum (for the all-gather).
// The two strict-mode queue counters (ticket[2], ticket[3]) must be zero on entry (zero_counters).
// rider: an ORDER job (chan_order_body) attached to sweep 2 -- it produces cs.spos for the fix-up kernels.
#ifndef VC2_FRAME_A
#define VC2_FRAME_A 4
#endif
int launch_phase1(const Plan& p, const void* x, const ChanSet& cs, void* ws, bool single_rank, hipStream_t st,
                  const OrderArgs& rider = OrderArgs{}, bool own_stats = false) {
  // (ws holds sweep 1's partials of x -- every caller ran the statistics sweep with this workspace: they bound
  // sum |x^| for the centre-mean replay margins)
  const FrameStatSrc fs{wsp<double>(ws, p.o_part_stats), p.stat_splits, p.BF, int(p.N),
                        wsp<int>(ws, p.o_ticket) + kTkFrameReplays};
  const int C = cs.C;
  double* part = wsp<double>(ws, p.o_part_col);
  double* cpart = wsp<double>(ws, p.o_csum_part);
  const int npl = int(cdiv(C, 64));
  const bool ord = ord_on(p, cs);                                // torch-ordered frame sums (k_norm_colsum2<.., ORD>)
  OrderArgs ride = rider;
  if (ride.perm && (p.VEC == 1 || !fast_acc(p, cs) || g_prof || ride.k > 4096)) {   // no rider on this sweep variant (or per-kernel
    int rc = launch_chan_order(ride, st);                          // timing wanted): the ORDER job as its own kernel
    if (rc) return rc;
    ride = OrderArgs{};
  }
  if (ride.perm) ride.parts = order_parts(ride.k, rider_parts_max(p.R, p.D * p.ES));
#ifdef VC2_RIDER_PROBE
  if (ride.perm) { const char* e = getenv("VC2_RIDER_PROBE_MODE"); if (e) ride.parts |= atoi(e) << 8; }
#endif
  { ProfScope ps_(KID_NORM_COLSUM, st);
  int rc = VC2_OK;
  VC2_DISPATCH_VEC(p, rc = VC2_DISPATCH_NPL(npl, launch_norm_t, p, x, cs, ws, ride, st));
  if (rc) return rc; }
  // Fused centre launch (see k_frame_centres): the norm fix-ups as rider workgroups -- single rank (the corrections of the
  // video-centre sums are added by k_video_centre), 16-bit inputs on the vector path, <= 32 compact positions per lane
  // (the rider's registers must fit a 1024-thread workgroup), frames short enough for the replays; else, and for
  // per-kernel timing, k_norm_fix runs first and k_frame_centres applies its corrections itself.
  const bool replays = cs.strict != 0 && p.dt != VC2_F32;
  const int rcap = int(std::min<int64_t>(p.F * int64_t(C), INT32_MAX / 2));      // regular entries: every mean at most once
  const int rcap2 = int(p.F * cdiv(C, 64));                                        // correction entries: every (frame, column block) once
#ifdef VC2_NO_FUSED_FIX
  const bool fused = false;
#else
  const bool fused = replays && single_rank && !g_prof && p.VEC > 1 && npl <= 32 && (((p.N >> 4) + 15) >> 4) <= kCFixSolo &&
                     p.F * int64_t(C) <= INT32_MAX / 2;
#endif
  if (cs.strict && !fused) {
    ProfScope ps_(KID_OTHER, st);
    int rc = VC2_OK;
    VC2_DISPATCH_VEC(p, rc = VC2_DISPATCH_NPL(npl, launch_norm_fix_t, p, x, cs, ws, st));
    if (rc) return rc;
  }
  { ProfScope ps_(KID_CENTRES, st);
  const int FG = int(cdiv(p.F, kCentreFL));
  // rider rows: fp16 queues ~3 % of the rows (its T ulp is 2^13 fp32 ulps), bf16 ~0.4 %; debug mode 2 all of them
  const int bxn = int(cdiv(C, 64));
  // (a rider workgroup is as large as a frame workgroup -- 16 waves -- and with ~120 VGPRs only ONE workgroup fits a CU:
  // every rider workgroup takes a whole CU for the length of its rows' chains.  So few of them with many busy waves.)
  static const int env_fw = [] { const char* e = getenv("VC2_FIX_WAVES"); return e ? atoi(e) : 0; }();
  const size_t fix_row = (std::max(row_lds_bytes(int(p.D), p.ES), size_t(C) * 4 + 16) + 15) / 16 * 16;
  int fwaves = env_fw > 0 ? env_fw : VC2_FIX_WAVES_DEFAULT;
  fwaves = int(std::max<int64_t>(1, std::min<int64_t>(std::min(fwaves, kCentreFL), int64_t(160 * 1024 - 256 - 24 * 1024) / int64_t(fix_row))));
  // rider slots ~ the rows sweep 2 queues: bf16 ~0.4 % of them, fp16 ~3.7 % (its T ulp is 2^13 fp32 ulps); a rider wave
  // that finds more entries than slots takes several in turn
  const int64_t want_slots = cs.strict == 2 ? 8192 : std::max<int64_t>(64, p.dt == VC2_F16 ? p.R / 24 : p.R / 128);
  const int fy = !fused ? 0 : int(std::min<int64_t>(cdiv(want_slots, int64_t(bxn) * fwaves), cdiv(cs.strict == 2 ? 8192 : 1024, int64_t(bxn) * fwaves)));
  const FixRiders fr{fy, fwaves, p.CV, int(p.R), wsp<int>(ws, p.o_ticket) + kTkFixCount, wsp<unsigned long long>(ws, p.o_nfixlist),
                     wsp<int>(ws, p.o_ticket) + kTkCorrCount, wsp<NormCorr>(ws, p.o_corr), wsp<float>(ws, p.o_den),
                     FixPush{wsp<uint32_t>(ws, p.o_rlist) + rcap, wsp<int>(ws, p.o_ticket) + kTkFixEntries, rcap2, wsp<int>(ws, p.o_fmark)}};
  const size_t fix_lds = !fused ? 0 : size_t(fwaves) * fix_row;
  int rcl = VC2_OK;
  auto launch_fc = [&](auto kernel) {
    if ((rcl = allow_big_lds(kernel, fix_lds, "k_frame_centres", 24 * 1024))) return;
    hipLaunchKernelGGL(kernel, dim3(unsigned(bxn), unsigned(FG + fy)), dim3(64 * kCentreFL), fix_lds, st,
                       part, int(p.F), ord ? p.ord_S : p.S, p.S_q, int(p.N), C, wsp<float>(ws, p.o_fc), cpart, x, int(p.D), cs.cols, cs.spos,
                       wsp<float>(ws, p.o_den),
                       (cs.strict && !fused) ? wsp<int>(ws, p.o_ticket) + kTkCorrCount : (int*)nullptr,
                       wsp<NormCorr>(ws, p.o_corr), cs.strict, wsp<int>(ws, p.o_vticket), fs,
                       margin_depth(p.N, cs.strict), (!single_rank && cs.strict == 3) ? 1 : 0, wsp<float>(ws, p.o_dmin),
                       (own_stats && cs.strict == 4 && !ord) ? double(VC2_FRAME_A) : 0.0, wsp<uint32_t>(ws, p.o_rlist), rcap, fr,
                       OrdSrc{ord ? wsp<float>(ws, p.o_bsum) : (const float*)nullptr, ord ? 1 : 0, ord ? p.ord_nA : 0});
  };
  if (!fused) {
    VC2_DISPATCH_DT(p.dt, launch_fc(k_frame_centres<DT, 1, 0>));
  } else if (p.dt == VC2_BF16) {
    constexpr int DT = VC2_BF16, VEC = Tr<VC2_BF16>::VEC;
    if (npl <= 8) launch_fc(k_frame_centres<DT, VEC, 8>); else if (npl <= 16) launch_fc(k_frame_centres<DT, VEC, 16>);
    else if (npl <= 28) launch_fc(k_frame_centres<DT, VEC, 28>); else launch_fc(k_frame_centres<DT, VEC, 32>);
  } else {
    constexpr int DT = VC2_F16, VEC = Tr<VC2_F16>::VEC;
    if (npl <= 8) launch_fc(k_frame_centres<DT, VEC, 8>); else if (npl <= 16) launch_fc(k_frame_centres<DT, VEC, 16>);
    else if (npl <= 28) launch_fc(k_frame_centres<DT, VEC, 28>); else launch_fc(k_frame_centres<DT, VEC, 32>);
  }
  if (rcl) return rcl;
  // the frame means k_frame_centres listed are replayed by rider waves of the video-centre launch (single rank), or
  // by a launch of their own (frame-sharded pass: its video centre comes later, from the all-gathered sums)
  const FrameReplay frp{wsp<uint32_t>(ws, p.o_rlist), wsp<int>(ws, p.o_ticket) + kTkFrameReplays, rcap,
                        wsp<float>(ws, p.o_fc), int(p.N),
                        FrameFix{part, ord ? p.ord_S : p.S, p.S_q, cs.strict, margin_depth(p.N, cs.strict),
                                 (own_stats && cs.strict == 4) ? double(VC2_FRAME_A) : 0.0, fs,
                                 wsp<int>(ws, p.o_ticket) + kTkCorrCount, wsp<NormCorr>(ws, p.o_corr),
                                 OrdSrc{ord ? wsp<float>(ws, p.o_bsum) : (const float*)nullptr, ord ? 1 : 0, ord ? p.ord_nA : 0}},
                        fused ? wsp<uint32_t>(ws, p.o_rlist) + rcap : (const uint32_t*)nullptr,
                        wsp<int>(ws, p.o_ticket) + kTkFixEntries, rcap2,
                        fused ? wsp<int>(ws, p.o_fmark) : (const int*)nullptr};
  static const int env_rw = [] { const char* e = getenv("VC2_REPLAY_WAVES"); return e ? atoi(e) : 0; }();
  // rider waves (debug mode 2 replays every mean; fp16 lists ~8x bf16's).  ORD: no mean is replayed -- the riders only redo the
  // frames that hold a row with a corrected norm (one entry per block of 64 columns: bf16 ~2 frames per pass, fp16 ~16)
  const int rwaves = env_rw > 0 ? env_rw : ord ? (p.dt == VC2_F16 ? 1024 : 256) : cs.strict == 2 ? 8192 : (p.dt == VC2_F16 ? 4096 : 1024);
  if (single_rank) {
    const int lpv = cascade_lp(p.R);
    const int G1v = int(cdiv(p.R >> lpv, int64_t(1) << lpv));
    const int Y = int(std::max<int64_t>(1, std::min<int64_t>(32, cdiv(G1v, 64 >> std::min(lpv, 6)))));
    const int RY = replays ? int(cdiv(rwaves, cdiv(C, 64))) : 0;
    VC2_DISPATCH_DT(p.dt, hipLaunchKernelGGL((k_video_centre<DT>), dim3(unsigned(cdiv(C, 64)), unsigned(Y + RY)), dim3(64),
                                             vc_lds_bytes(p.R, p.N), st, cpart, FG, int64_t(C), C, p.R, wsp<float>(ws, p.o_vc), x, int(p.D),
                                             cs.cols, cs.spos, wsp<float>(ws, p.o_den), cs.strict, 1,
                                             wsp<int>(ws, p.o_ticket) + kTkVcFragile,
                                             wsp<float>(ws, p.o_vscratch), p.vstride, wsp<int>(ws, p.o_vticket),
                                             (uint8_t*)nullptr, 0, margin_depth(p.R, cs.strict), fs,
                                             wsp<float>(ws, p.o_dmin), int(p.F), frp, Y,
                                             fused ? wsp<int>(ws, p.o_ticket) + kTkCorrCount : (const int*)nullptr,
                                             wsp<NormCorr>(ws, p.o_corr)));
  } else if (replays && !ord) {                 // (ORD: nothing is listed)
    VC2_DISPATCH_DT(p.dt, hipLaunchKernelGGL((k_frame_replay<DT>), dim3(unsigned(rwaves)), dim3(64), 0, st, frp, x, int(p.D),
                                             C, cs.cols, cs.spos, wsp<float>(ws, p.o_den)));
  }
  }
  return check_launch("scores phase 1");
}

int zero_counters(const Plan& p, void* ws, hipStream_t st) {
  hipError_t e = hipMemsetAsync(wsp<int>(ws, p.o_ticket), 0, 64, st);
  if (e == hipSuccess) e = hipMemsetAsync(wsp<char>(ws, p.o_nfixlist), 0, size_t(p.R) * 8 + size_t(cdiv(p.F, 2)) * 8, st);
  if (e != hipSuccess) return fail(VC2_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  return VC2_OK;
}

// sweep 3 incl. the per-token epilogue; s != nullptr: also the per-frame uniqueness scores (stage APIs / sharded
// path -- the fused pass hands the workgroup partials straight to k_select)
int launch_phase2(const Plan& p, const void* x, const ChanSet& cs, void* ws, void* v_T, void* f_T,
                  float* total, float* s, hipStream_t st) {
  const int C = cs.C;
  if (p.rows_per_split2 > kDistMaxRows) return fail(VC2_ERR_UNSUPPORTED, "internal: sweep-3 split too long");
  { int rc = VC2_OK;
  const int npl = int(cdiv(C, 64));
  const DistOut o{v_T, f_T, total};
  VC2_DISPATCH_VEC(p, rc = VC2_DISPATCH_NPL(npl, launch_dist_t, p, x, cs, ws, o, st));
  if (rc) return rc; }
  if (s) {
    ProfScope ps_(KID_EPILOGUE, st);
    VC2_DISPATCH_DT(p.dt, hipLaunchKernelGGL((k_frame_scores<DT>), dim3(unsigned(cdiv(p.F, 128))), dim3(128), 0, st,
                                             wsp<double>(ws, p.o_vpart), int(p.F), p.S2, int(p.N), s));
  }
  return check_launch("scores phase 2");
}

int launch_scales(int dt, const float* s, int64_t F, double base, double temp, float* zbuf, float* scales_f32,
                  void* scales_T, hipStream_t st) {
  ProfScope ps_(KID_SCALES, st);
  VC2_DISPATCH_DT(dt, hipLaunchKernelGGL((k_scales<DT>), dim3(1), dim3(kBudNT), 0, st, s, int(F), float(base),
                                         float(temp), zbuf, scales_f32, scales_T));
  return check_launch("compute_scales");
}

// F budget frames, of which this launch selects [f0, f0 + F_sel) (total / ks / offs are indexed by the local frame).
// Budgets: vpart (sweep-3 partials, S2 per frame) or frame_scores -> compute_scales in the kernel; else scales_f32[F].
struct BudgetSrc { const float* scales_f32; const double* vpart; int S2; const float* frame_scores; double base; double temp;
                   float* scales_out; int64_t tpf = 0;         // tpf: the multiplier of vidcom2.py:72 (0: N)
                   int* status = nullptr;                      // the pass's kTkStatus word (reported in K_out[1])
                   long long* khost = nullptr;                 // pinned host mirror: K, early status, final status (vc2_compress_ex2)
                   int* arrive = nullptr; };                   // ... and the arrival ticket behind the final word (zero on entry)
std::atomic<int> g_force_guard{-1};       // test hook (vc2_selftest_force_guard): this local frame's selection reports a guard hit
int launch_select(int dt, const float* total, int64_t F, int64_t f0, int64_t F_sel, int64_t N, int map_mode,
                  int64_t grid_h, int64_t* ks, int64_t* offs, int64_t* idx_out, int64_t cap, int64_t* K_out,
                  const BudgetSrc& b, hipStream_t st) {
  const size_t smem = sel2_bytes(int(N), dt == VC2_F32 ? 8 : 4);
  // K_out[1] collects the launch's status bits by atomic OR: it starts at zero -- the one-launch pass's first kernel
  // zeroed it (b.status != nullptr); a stage call zeroes it here
  if (!b.status && hipMemsetAsync(K_out + 1, 0, 8, st) != hipSuccess) return fail(VC2_ERR_LAUNCH, "status memset failed");
  ProfScope ps_(KID_SELECT, st);
  VC2_DISPATCH_DT(dt, {
    int rca = allow_big_lds(&k_select<DT>, smem, "k_select");
    if (rca) return rca;
    hipLaunchKernelGGL((k_select<DT>), dim3(unsigned(F_sel)), dim3(kFrameNT), smem, st, total, b.scales_f32, int(F),
                       int(f0), int(N), int(b.tpf > 0 ? b.tpf : N), map_mode, int(grid_h), N, cap, ks, offs, idx_out, K_out,
                       b.vpart, b.S2,
                       b.frame_scores, float(b.base), float(b.temp), b.scales_out, b.status, b.khost, b.arrive,
                       g_force_guard.load(std::memory_order_relaxed));
  });
  return check_launch("select");
}

int launch_gather(const GSArgs& a, int n_src, hipStream_t st) {
  const int64_t rows = a.n_max + a.tail_rows;
  if (rows <= 0 || n_src <= 0) return VC2_OK;
  const unsigned grid = unsigned(std::min<int64_t>(rows, 16384));
  ProfScope ps_(KID_GATHER_ROWS, st);
  hipLaunchKernelGGL(k_gather_rows, dim3(grid, unsigned(n_src)), dim3(256), 0, st, a);
  return check_launch("gather_rows");
}
int launch_gather_rows(const void* src, int64_t src_rows, int64_t D, int ES, const int64_t* idx,
                       const int64_t* K_dev, int64_t cap, void* dst, hipStream_t st, const void* tail = nullptr,
                       int64_t tail_rows = 0) {
  GSArgs a{};
  a.src[0] = static_cast<const unsigned char*>(src); a.dst[0] = static_cast<unsigned char*>(dst);
  a.src_rows[0] = src_rows; a.dst_rows[0] = cap + tail_rows;
  a.row_bytes = D * ES; a.idx = idx; a.n_dev = K_dev; a.n_max = cap;
  a.tail = static_cast<const unsigned char*>(tail); a.tail_rows = tail ? tail_rows : 0;
  return launch_gather(a, 1, st);
}

#endif   // VC2_DEV_ONLY
}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
#ifndef VC2_DEV_ONLY      // (VC2_DEV_ONLY: device-only builds of single kernels for ISA inspection, scripts/dev_isa.sh)
extern "C" {

const char* vc2_last_error(void) { return g_err; }
const char* vc2_version(void) { return "vidcom2_amd 0.1 (gfx950)"; }

int vc2_set_mode(int mode) {
  if (mode < 0 || mode > 4) return fail(VC2_ERR_ARG, "mode must be 0 (exact), 4 (torch order: the default), 1 (torch order, empirical margins only) or 3 (torch order, proven centre margins)");   // 2: debug
  g_mode_default.store(mode, std::memory_order_relaxed);
  g_mode_thread = -1;                               // (the caller sees what it just set)
  return VC2_OK;
}
int vc2_set_thread_mode(int mode) {
  if (mode < -1 || mode > 4) return fail(VC2_ERR_ARG, "mode must be -1 (follow the process), 0 (exact), 1, 3 or 4 (torch order variants)");
  g_mode_thread = mode;
  return VC2_OK;
}
int vc2_get_mode(void) { return cur_mode(); }

int vc2_workspace_bytes(int64_t F, int64_t N, int64_t D, int dtype, size_t* out_bytes) {
  if (!out_bytes) return fail(VC2_ERR_ARG, "out_bytes is null");
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  *out_bytes = p.total_bytes;
  return VC2_OK;
}

int64_t vc2_kept_capacity(int64_t F, int64_t N, double base_scale) {
  // sum_f scale_f = base*F up to the T rounding of four ops per frame (<= 2^-6 relative in bf16);
  // each k_f = max(1, round(scale_f*N)) adds at most 1.5 -> a safe, tight bound, capped at F*N.
  if (F <= 0 || N <= 0) return 0;
  double b = base_scale > 0 ? base_scale : 0.0;
  double cap = b * double(F) * double(N) * 1.04 + 2.0 * double(F) + 64.0;
  double full = double(F) * double(N);
  return int64_t(cap < full ? cap : full);
}

int vc2_stat_block_frames(void) { return kStatBlockFrames; }

int vc2_chan_stats(const void* x, int64_t F, int64_t N, int64_t D, int dtype, int64_t F_total, int block_frames,
                   void* ws, size_t ws_bytes, double* bstats, void* stream) {
  if (!x || !bstats) return fail(VC2_ERR_ARG, "null pointer");
  if (block_frames < 1 || block_frames > kStatBlockFrames) return fail(VC2_ERR_ARG, "block_frames must be 1..%d", kStatBlockFrames);
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, F_total, block_frames);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  return launch_chan_stats(p, x, ws, bstats, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int vc2_chan_var_from_stats(const double* bstats, int64_t NB, int64_t rows_per_block, int64_t R_total, int64_t D,
                            int dtype, void* var_T, float* var_f32, void* stream) {
  if (!bstats || NB <= 0 || D <= 0 || R_total <= 0 || rows_per_block <= 0 || (NB - 1) * rows_per_block >= R_total ||
      NB * rows_per_block < R_total)
    return fail(VC2_ERR_ARG, "bad stats arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_var_from_stats<DT>), dim3(unsigned(cdiv(D, 64))), dim3(64 * kRedGL), 0, st,
                                            bstats, int(NB), rows_per_block, R_total - (NB - 1) * rows_per_block, int(D),
                                            var_T, var_f32, (int*)nullptr, PartSrc{},
                                            make_fold_tab(int(NB), rows_per_block, R_total - (NB - 1) * rows_per_block)));
  return check_launch("var_from_stats");
}

int vc2_chan_var(const void* x, int64_t R, int64_t D, int dtype, void* ws, size_t ws_bytes, void* var_T,
                 float* var_f32, void* stream) {
  if (!x) return fail(VC2_ERR_ARG, "x is null");
  Plan p;
  int rc = make_plan(1, R, D, dtype, &p);   // the variance does not depend on the frame structure
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  float* vf = var_f32 ? var_f32 : wsp<float>(ws, p.o_var_f32);
  return launch_chan_stats(p, x, ws, nullptr, var_T, vf, static_cast<hipStream_t>(stream));
}

int vc2_chan_select(const float* var_f32, int64_t D, int64_t k, uint8_t* mask, int32_t* cols, int32_t* perm,
                    int32_t* order, int32_t* opos, int32_t* spos, void* stream) {
  if (!var_f32 || (!mask && !cols && !perm && !order)) return fail(VC2_ERR_ARG, "null pointer");
  if ((order || opos || spos) && !perm) return fail(VC2_ERR_ARG, "order / opos / spos need the perm scratch array");
  if ((opos || spos) && !(cols && opos)) return fail(VC2_ERR_ARG, "opos / spos need cols and opos");
  if (D > 8192) return fail(VC2_ERR_UNSUPPORTED, "D=%lld > 8192 channels", (long long)D);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = launch_chan_select(var_f32, D, k, mask, cols, perm, st);
  if (rc || !(order || opos || spos)) return rc;
  return launch_chan_order(OrderArgs{var_f32, perm, cols, order, opos, spos, int(D), int(k), 1, nullptr, nullptr, nullptr}, st);
}

int vc2_gather_cols(const void* x, int64_t R, int64_t D, int dtype, const int64_t* idx, int64_t C, void* out,
                    void* stream) {
  if (!x || !idx || !out) return fail(VC2_ERR_ARG, "null pointer");
  if (R <= 0 || C <= 0) return VC2_OK;
  if (R > 65535LL * 65535LL) return fail(VC2_ERR_UNSUPPORTED, "too many rows");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // grid.y is limited to 65535: loop in slices of rows
  for (int64_t r0 = 0; r0 < R; r0 += 65535) {
    const int64_t nr = std::min<int64_t>(65535, R - r0);
    const int ES = dtype == VC2_F32 ? 4 : 2;
    const void* xs = static_cast<const char*>(x) + r0 * D * ES;
    void* os = static_cast<char*>(out) + r0 * C * ES;
    VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_gather_cols<DT>), dim3(unsigned(cdiv(C, 256)), unsigned(nr)),
                                              dim3(256), 0, st, xs, nr, int(D), idx, int(C), os));
  }
  return check_launch("gather_cols");
}

static int check_cols(const int32_t* cols, int64_t C, int64_t D) {
  if (C <= 0 || C > D || (!cols && C != D)) return fail(VC2_ERR_ARG, "bad channel list (C=%lld, D=%lld)", (long long)C, (long long)D);
  if (C > 4096) return fail(VC2_ERR_UNSUPPORTED, "more than 4096 scored channels (C=%lld)", (long long)C);
  return VC2_OK;
}

int vc2_scores_phase1(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols, int64_t C,
                      int32_t* spos, const int32_t* perm, const float* var_f32, int64_t F_total, void* ws,
                      size_t ws_bytes, double* csum_parts, void* stream) {
  if (!x) return fail(VC2_ERR_ARG, "x is null");
  if (perm && !(var_f32 && cols && spos)) return fail(VC2_ERR_ARG, "the ORDER rider needs perm, var_f32, cols and spos");
  { int rcc = check_cols(cols, C, D); if (rcc) return rcc; }
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, F_total);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if ((rc = zero_counters(p, ws, st))) return rc;
  const ChanSet cs = make_chanset(p, cols, spos, C);
  OrderArgs rider{};
  if (perm && cs.strict)      // torch.topk's ORDER of the channels (-> spos), replayed by a rider workgroup of sweep 2
    rider = OrderArgs{var_f32, perm, cols, wsp<int>(ws, p.o_order), wsp<int>(ws, p.o_opos), spos, int(D), int(C), 1, nullptr, nullptr,
                      wsp<int>(ws, p.o_ticket) + kTkStatus};
  rc = launch_phase1(p, x, cs, ws, /*single_rank=*/false, st, rider, /*own_stats=*/true);   // (vc2_chan_stats ran with this workspace)
  if (rc) return rc;
  if (csum_parts) {       // per group of kCentreFL frames, in frame order: the fp64 sums of x^ [ceil(F/16)][C], then the
    //                        groups' bounds of sum |x^| [ceil(F/16)][C] (the replay margin of the video centre)
    hipError_t e = hipMemcpyAsync(csum_parts, wsp<double>(ws, p.o_csum_part), size_t(2) * size_t(cdiv(F, kCentreFL)) * size_t(C) * 8,
                                  hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(VC2_ERR_LAUNCH, "csum copy: %s", hipGetErrorString(e));
  }
  return VC2_OK;
}

// can the frame-sharded pass replay its video-centre means (see k_vc_blocks)?
static bool vc_blocks_ok(const Plan& p, int64_t R_total, int strict) {
  const int64_t B = int64_t(1) << cascade_lp(R_total);      // (a level-0 block meets at most two ranks)
  (void)B;                                                    // (a block may meet any number of ranks: k_vc_finish)
  return strict != 0 && p.dt != VC2_F32 && R_total % p.R == 0 && cascade_modelled(R_total);
}

int vc2_video_centre_blocks(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols, int64_t C,
                            const int32_t* spos, const double* csum_all, int64_t P, int64_t csum_stride,
                            int64_t rows_per_rank, int64_t R_total, int64_t row0, void* ws, size_t ws_bytes,
                            float* blocks_out, int cap, void* stream) {
  if (!x || !csum_all || !blocks_out || P <= 0 || csum_stride < C || cap <= 0 || rows_per_rank < 0 || row0 < 0 ||
      row0 + F * N > R_total ||
      (rows_per_rank > 0 && (rows_per_rank % 2 || P % rows_per_rank)))
    return fail(VC2_ERR_ARG, "bad video_centre_blocks arguments");
  { int rcc = check_cols(cols, C, D); if (rcc) return rcc; }
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, R_total / N);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ChanSet cs0 = make_chanset(p, cols, spos, C);
  if (!vc_blocks_ok(p, R_total, cs0.strict)) return VC2_OK;      // nothing to exchange: phase 2 keeps the exact means
  uint8_t* vflag = wsp<uint8_t>(ws, p.o_mask);
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_video_centre<DT>), dim3(unsigned(cdiv(C, 64))), dim3(64), 512 /* (replay_rows = 0: no LDS use) */, st,
                                            csum_all, int(P), csum_stride, int(C), R_total, wsp<float>(ws, p.o_vc), x,
                                            int(D), cols, spos, wsp<float>(ws, p.o_den), cs0.strict, 0,
                                            wsp<int>(ws, p.o_ticket) + 5, (float*)nullptr, 0, (int*)nullptr, vflag,
                                            int(rows_per_rank), margin_depth(R_total, cs0.strict), FrameStatSrc{},
                                            (const float*)nullptr, 0));
  const int lpv = cascade_lp(R_total);
  const int64_t nb = p.R >> lpv;                               // (at most; one more grid column for the raw edges)
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_vc_blocks<DT>), dim3(unsigned(cdiv(nb, 64) + 1), unsigned(cap)), dim3(64), 0,
                                            st, vflag, int(C), x, int(D), cols, spos, wsp<float>(ws, p.o_den), p.R, row0,
                                            lpv, blocks_out, 0));
  return check_launch("video_centre_blocks");
}

// Rounds of exchange 2b beyond the first `cap` flagged columns (vc2.h): the records of flagged columns
// [col_offset, col_offset + cap), the flags being those vc2_video_centre_blocks left in the workspace
int vc2_video_centre_blocks_round(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols,
                                  int64_t C, const int32_t* spos, int64_t R_total, int64_t row0, void* ws, size_t ws_bytes,
                                  float* blocks_out, int cap, int col_offset, void* stream) {
  if (!x || !blocks_out || cap <= 0 || col_offset < 0 || row0 < 0 || row0 + F * N > R_total)
    return fail(VC2_ERR_ARG, "bad video_centre_blocks_round arguments");
  { int rcc = check_cols(cols, C, D); if (rcc) return rcc; }
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, R_total / N);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  if (!vc_blocks_ok(p, R_total, make_chanset(p, cols, spos, C).strict)) return VC2_OK;
  const int lpv = cascade_lp(R_total);
  const int64_t nb = p.R >> lpv;
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_vc_blocks<DT>), dim3(unsigned(cdiv(nb, 64) + 1), unsigned(cap)), dim3(64), 0,
                                            static_cast<hipStream_t>(stream), wsp<uint8_t>(ws, p.o_mask), int(C), x, int(D),
                                            cols, spos, wsp<float>(ws, p.o_den), p.R, row0, lpv, blocks_out, col_offset));
  return check_launch("video_centre_blocks_round");
}

// how many columns vc2_video_centre_blocks flagged (SYNCHRONISES the stream): the rounds the caller has to go
int vc2_video_centre_flagged(int64_t F, int64_t N, int64_t D, int dtype, int64_t R_total, const void* ws, size_t ws_bytes,
                             int32_t* count_host, void* stream) {
  if (!count_host) return fail(VC2_ERR_ARG, "null pointer");
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, R_total / N);
  if (rc) return rc;
  if ((rc = need_ws(p, const_cast<void*>(ws), ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(count_host, static_cast<const char*>(ws) + p.o_ticket + 5 * sizeof(int), sizeof(int32_t),
                     hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fail(VC2_ERR_LAUNCH, "flagged-column read-back failed");
  return VC2_OK;
}

// ... and the finish of one round: torch's cascade over the whole video for flagged columns [col_offset, col_offset + cap)
// from the all-gathered records; the video centre in the workspace is corrected in place (vc2_scores_phase2_blocks with
// world = -1 then scores with it as it stands)
int vc2_video_centre_finish_round(int64_t F, int64_t N, int64_t D, int dtype, int64_t C, const int32_t* spos,
                                  int64_t R_total, void* ws, size_t ws_bytes, const float* blocks_all, int world, int cap,
                                  int col_offset, void* stream) {
  if (!blocks_all || world <= 0 || cap <= 0 || col_offset < 0) return fail(VC2_ERR_ARG, "bad video_centre_finish_round arguments");
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, R_total / N);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  if (R_total != int64_t(world) * p.R) return fail(VC2_ERR_ARG, "finish_round: R_total != world * F * N");
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_vc_finish<DT>), dim3(unsigned(cap)), dim3(256), 0,
                                            static_cast<hipStream_t>(stream), wsp<uint8_t>(ws, p.o_mask), int(C), spos,
                                            blocks_all, world, cap, p.R, R_total, wsp<float>(ws, p.o_vc),
                                            wsp<int>(ws, p.o_ticket) + 5, col_offset));
  return check_launch("video_centre_finish_round");
}

int vc2_scores_phase2(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols, int64_t C,
                      const int32_t* spos, const double* csum_all, int64_t P, int64_t csum_stride, int64_t rows_per_rank,
                      int64_t R_total, void* ws,
                      size_t ws_bytes, void* v_T, void* f_T, float* total_f32, float* s_f32, void* stream) {
  return vc2_scores_phase2_blocks(x, F, N, D, dtype, cols, C, spos, csum_all, P, csum_stride, rows_per_rank, R_total, ws,
                                  ws_bytes, v_T, f_T, total_f32, s_f32, nullptr, 0, 0, stream);
}

int vc2_scores_phase2_blocks(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols, int64_t C,
                             const int32_t* spos, const double* csum_all, int64_t P, int64_t csum_stride,
                             int64_t rows_per_rank, int64_t R_total, void* ws, size_t ws_bytes, void* v_T, void* f_T,
                             float* total_f32, float* s_f32, const float* blocks_all, int world, int cap, void* stream) {
  if (!x || !csum_all || P <= 0 || csum_stride < C || rows_per_rank < 0 ||
      (rows_per_rank > 0 && (rows_per_rank % 2 || P % rows_per_rank)))
    return fail(VC2_ERR_ARG, "bad phase-2 arguments");
  { int rcc = check_cols(cols, C, D); if (rcc) return rcc; }
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p, R_total / N);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ChanSet cs0 = make_chanset(p, cols, spos, C);
  const bool have_blocks = blocks_all && world > 0 && cap > 0 && vc_blocks_ok(p, R_total, cs0.strict) &&
                           R_total == int64_t(world) * p.R;
  const bool vc_final = world < 0;   // the caller ran vc2_video_centre_blocks + the finish rounds: the centre is final
  if (!have_blocks && !vc_final)     // (with blocks: vc2_video_centre_blocks already computed the means, the flags and the count)
    VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_video_centre<DT>), dim3(unsigned(cdiv(C, 64))), dim3(64), 512 /* (replay_rows = 0: no LDS use) */, st,
                                              csum_all, int(P), csum_stride, int(C), R_total, wsp<float>(ws, p.o_vc), x,
                                              int(D), cols, spos, wsp<float>(ws, p.o_den), cs0.strict, 0,
                                              wsp<int>(ws, p.o_ticket) + 5, (float*)nullptr, 0, (int*)nullptr,
                                              wsp<uint8_t>(ws, p.o_mask), int(rows_per_rank),
                                              margin_depth(R_total, cs0.strict), FrameStatSrc{}, (const float*)nullptr, 0));
  // ticket[5] = video-centre columns whose mean lies within the replay margin of a T rounding boundary.  With the
  // all-gathered level-0 block sums (vc2_video_centre_blocks) torch's cascade is finished here for the first `cap`
  // of them; the others keep the exactly rounded mean and stay counted (vc2_select_sharded reports K_out[2]).
  if (have_blocks)
    VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_vc_finish<DT>), dim3(unsigned(cap)), dim3(256), 0, st,
                                              wsp<uint8_t>(ws, p.o_mask), int(C), spos, blocks_all, world, cap,
                                              p.R, R_total, wsp<float>(ws, p.o_vc), wsp<int>(ws, p.o_ticket) + 5, 0));
  float* total = total_f32 ? total_f32 : wsp<float>(ws, p.o_total);
  float* s = s_f32 ? s_f32 : wsp<float>(ws, p.o_s);
  return launch_phase2(p, x, make_chanset(p, cols, spos, C), ws, v_T, f_T, total, s, st);
}

int vc2_scores(const void* x, int64_t F, int64_t N, int64_t D, int dtype, const int32_t* cols, int64_t C,
               const int32_t* spos, void* ws, size_t ws_bytes, void* v_T, void* f_T, float* total_f32,
               float* s_f32, void* stream) {
  if (!x) return fail(VC2_ERR_ARG, "x is null");
  { int rcc = check_cols(cols, C, D); if (rcc) return rcc; }
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ChanSet cs = make_chanset(p, cols, spos, C);
  if ((rc = zero_counters(p, ws, st))) return rc;
  // modes 3 / 4 bound sum |x^| from sweep-1 statistics of THESE columns (mean_delta): the stage call runs that sweep itself
  const bool stats = cs.strict == 3 || cs.strict == 4;
  if (stats && (rc = launch_stats_sweep(p, x, ws, PoolSrc{}, st))) return rc;
  if ((rc = launch_phase1(p, x, cs, ws, /*single_rank=*/true, st, OrderArgs{}, /*own_stats=*/stats))) return rc;
  float* total = total_f32 ? total_f32 : wsp<float>(ws, p.o_total);
  float* s = s_f32 ? s_f32 : wsp<float>(ws, p.o_s);
  return launch_phase2(p, x, cs, ws, v_T, f_T, total, s, st);
}

int vc2_compute_scales(const void* s_T, int64_t F, double base, double temp, int dtype, void* ws, size_t ws_bytes,
                       void* scales_T, void* stream) {
  if (!s_T || !scales_T || F <= 0) return fail(VC2_ERR_ARG, "bad compute_scales arguments");
  // only needs 3*F floats of scratch
  const size_t need = align_up(size_t(F) * 4) * 3;
  if (!ws || ws_bytes < need) return fail(VC2_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* s32 = wsp<float>(ws, 0);
  float* zbuf = wsp<float>(ws, align_up(size_t(F) * 4));
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_widen<DT>), dim3(unsigned(cdiv(F, 256))), dim3(256), 0, st, s_T, F,
                                            s32));
  return launch_scales(dtype, s32, F, base, temp, zbuf, nullptr, scales_T, st);
}

int vc2_select(const void* scores_T, const void* scales_T, int64_t F, int64_t N, int64_t tpf, int dtype, int map_mode,
               int64_t grid_h, void* ws, size_t ws_bytes, int64_t* ks, int64_t* offs, int64_t* idx_out,
               int64_t cap, int64_t* K_out, void* stream) {
  if (!scores_T || !scales_T || !ks || !offs || !idx_out || !K_out || F <= 0 || N <= 0)
    return fail(VC2_ERR_ARG, "bad select arguments");
  if (N > 8192) return fail(VC2_ERR_UNSUPPORTED, "N=%lld > 8192 tokens per frame", (long long)N);
  if (map_mode == VC2_MAP_GRID_VID && (grid_h <= 0 || grid_h * grid_h != N))
    return fail(VC2_ERR_ARG, "grid_vid mapping needs N == grid_h^2");
  const size_t o_sc = align_up(size_t(F) * N * 4);
  const size_t need = o_sc + align_up(size_t(F) * 4);
  if (!ws || ws_bytes < need) return fail(VC2_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* tot = wsp<float>(ws, 0);
  float* sc = wsp<float>(ws, o_sc);
  VC2_DISPATCH_DT(dtype, {
    hipLaunchKernelGGL((k_widen<DT>), dim3(unsigned(cdiv(F * N, 256))), dim3(256), 0, st, scores_T, F * N, tot);
    hipLaunchKernelGGL((k_widen<DT>), dim3(unsigned(cdiv(F, 256))), dim3(256), 0, st, scales_T, F, sc);
  });
  return launch_select(dtype, tot, F, 0, F, N, map_mode, grid_h, ks, offs, idx_out, cap, K_out,
                       BudgetSrc{sc, nullptr, 0, nullptr, 0.0, 0.01, nullptr, tpf}, st);
}

int vc2_map_indices(const int64_t* local_idx, const int64_t* ks, const int64_t* offs, int64_t F, int map_mode,
                    int64_t stride_or_h, int64_t* out, void* stream) {
  if (!local_idx || !ks || !offs || !out || F <= 0) return fail(VC2_ERR_ARG, "bad map arguments");
  hipLaunchKernelGGL(k_map_indices, dim3(unsigned(F)), dim3(128), 0, static_cast<hipStream_t>(stream), local_idx,
                     ks, offs, map_mode, stride_or_h, out);
  return check_launch("map_indices");
}

int vc2_pool_out_tokens(int64_t H, int64_t W, int mode, int64_t* h_out, int64_t* w_out) {
  if (H < 2 || W < 2 || mode < 1 || mode > 3 || !h_out || !w_out) return fail(VC2_ERR_ARG, "bad pool arguments");
  *h_out = mode == VC2_POOL_BILINEAR ? (H + 1) / 2 : H / 2;
  *w_out = mode == VC2_POOL_BILINEAR ? (W + 1) / 2 : W / 2;
  return VC2_OK;
}

int vc2_pool_stats(const void* xin, int64_t F, int64_t H, int64_t W, int64_t D, int dtype, int mode, void* ws,
                   size_t ws_bytes, void* x_out, void* stream) {
  if (!xin || !x_out) return fail(VC2_ERR_ARG, "null pointer");
  int64_t h, w;
  int rc = vc2_pool_out_tokens(H, W, mode, &h, &w);
  if (rc) return rc;
  if (H > 32768 || W > 32768) return fail(VC2_ERR_UNSUPPORTED, "H, W up to 32768");
  if (mode == VC2_POOL_BILINEAR && D % (dtype == VC2_F32 ? 8 : 16) != 0)
    return fail(VC2_ERR_UNSUPPORTED, "bilinear pooling: D=%lld is not a multiple of torch's vector width (%d)",
                (long long)D, dtype == VC2_F32 ? 8 : 16);
  // ATen hands an NCHW-contiguous input (what get_2dPool builds) to the vectorised channels-last kernel -- the one whose
  // association is reproduced here -- only while out_h + out_w <= 128 (UpSampleKernel.cpp
  // _use_vectorized_kernel_cond_2d); larger grids take upsample_generic_Nd, which adds in another order
  if (mode == VC2_POOL_BILINEAR && h + w > 128)
    return fail(VC2_ERR_UNSUPPORTED, "bilinear pooling to %lld x %lld: torch switches kernels (and summation order) "
                "beyond out_h + out_w = 128", (long long)h, (long long)w);
  Plan p;
  if ((rc = make_plan(F, h * w, D, dtype, &p))) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  const PoolSrc pool{xin, int(H), int(W), int(h), int(w), mode};
  return launch_stats_sweep(p, x_out, ws, pool, static_cast<hipStream_t>(stream));
}

int vc2_gather_scatter(const void* const* srcs, const int64_t* src_rows, void* const* dsts, const int64_t* dst_rows,
                       int n_src, int64_t D, int dtype, const int64_t* idx, const int64_t* n_dev, int64_t n_max,
                       const int64_t* dst_pos, int64_t dst_row0, const void* tail, int64_t tail_rows, int32_t* status,
                       void* stream) {
  if (!srcs || !dsts || !src_rows || !dst_rows) return fail(VC2_ERR_ARG, "null pointer");
  if (n_src < 1 || n_src > kGSMaxSrc) return fail(VC2_ERR_ARG, "n_src=%d outside [1, %d]", n_src, kGSMaxSrc);
  if (dtype < 0 || dtype > 2 || D <= 0 || n_max < 0 || tail_rows < 0 || (tail_rows > 0 && !tail))
    return fail(VC2_ERR_ARG, "bad gather_scatter arguments");
  GSArgs a{};
  for (int t = 0; t < n_src; ++t) {
    if (!srcs[t] || !dsts[t]) return fail(VC2_ERR_ARG, "null tensor %d", t);
    a.src[t] = static_cast<const unsigned char*>(srcs[t]); a.dst[t] = static_cast<unsigned char*>(dsts[t]);
    a.src_rows[t] = src_rows[t]; a.dst_rows[t] = dst_rows[t];
  }
  a.row_bytes = D * (dtype == VC2_F32 ? 4 : 2);
  a.idx = idx; a.n_dev = n_dev; a.n_max = n_max; a.dst_pos = dst_pos; a.dst_row0 = dst_row0;
  a.tail = static_cast<const unsigned char*>(tail); a.tail_rows = tail ? tail_rows : 0; a.status = status;
  return launch_gather(a, n_src, static_cast<hipStream_t>(stream));
}

int vc2_keep_positions(const uint8_t* video_mask, int64_t S, const int64_t* kept, const int64_t* K_dev, int64_t K_max,
                       const uint8_t* visual_mask, int64_t* keep_out, int64_t keep_cap, int64_t* vis_rows_out,
                       int64_t vis_cap, int64_t* counts_out, void* stream) {
  if (!video_mask || S <= 0 || K_max < 0 || (K_max > 0 && !kept) || keep_cap < 0 || vis_cap < 0)
    return fail(VC2_ERR_ARG, "bad keep_positions arguments");          // (keep_out may be null when nothing is kept)
  if (S > (int64_t(1) << 31) - 1) return fail(VC2_ERR_UNSUPPORTED, "S=%lld positions", (long long)S);
  if (counts_out) {       // pinned host memory (a caller that spins on counts_out[2] instead of copying)? then its device alias
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, counts_out) == hipSuccess) {
      if (at.type == hipMemoryTypeHost && at.devicePointer) counts_out = static_cast<int64_t*>(at.devicePointer);
    } else {
      (void)hipGetLastError();
    }
  }
  int use_bitmap = S <= kKeepBitmapMaxS ? 1 : 0;
  size_t smem = use_bitmap ? size_t((S + 31) / 32) * 4 + 16 : 16;
  auto go = [&](auto kernel) -> int {
    int rca = allow_big_lds(kernel, smem, "k_keep_positions", 128);
    if (rca && use_bitmap) {      // (a device that does not grant the bitmap's LDS: the binary-search form needs 16 bytes)
      use_bitmap = 0; smem = 16; g_err[0] = 0;
      rca = allow_big_lds(kernel, smem, "k_keep_positions", 128);
    }
    if (rca) return rca;
    hipLaunchKernelGGL(kernel, dim3(1), dim3(kKeepNT), smem, static_cast<hipStream_t>(stream), video_mask, S, kept,
                       K_dev, K_max, visual_mask, keep_out, keep_out ? keep_cap : 0, vis_rows_out,
                       vis_rows_out ? vis_cap : 0, counts_out, use_bitmap);
    return VC2_OK;
  };
  // positions per thread held in registers: 16 / 32 / 64 (prompts up to 65 536 positions); longer ones re-read their chunks
  const int64_t per = cdiv(S, kKeepNT);
  int rcg = per <= 16 ? go(k_keep_positions<1>) : per <= 32 ? go(k_keep_positions<2>) : per <= 64 ? go(k_keep_positions<4>)
                                                                                                 : go(k_keep_positions<0>);
  if (rcg) return rcg;
  return check_launch("keep_positions");
}

int vc2_gather_rows(const void* src, int64_t src_rows, int64_t D, int dtype, const int64_t* idx,
                    const int64_t* K_dev, int64_t cap, void* dst, void* stream) {
  if (!src || !idx || !K_dev || !dst) return fail(VC2_ERR_ARG, "null pointer");
  return launch_gather_rows(src, src_rows, D, dtype == VC2_F32 ? 4 : 2, idx, K_dev, cap, dst,
                            static_cast<hipStream_t>(stream));
}

int vc2_compress(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale, int map_mode,
                 int64_t grid_h, const void* gather_src, int64_t gather_rows, void* ws, size_t ws_bytes,
                 void* out_rows, int64_t* idx_out, int64_t cap, int64_t* ks, int64_t* K_out, void* v_T, void* f_T,
                 void* stream) {
  return vc2_compress_ex(x, F, N, D, dtype, base_scale, map_mode, grid_h, gather_src, gather_rows, ws, ws_bytes,
                           out_rows, idx_out, cap, ks, K_out, v_T, f_T, nullptr, 0, 0, stream);
}

int vc2_compress_ex(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale, int map_mode,
                      int64_t grid_h, const void* gather_src, int64_t gather_rows, void* ws, size_t ws_bytes,
                      void* out_rows, int64_t* idx_out, int64_t cap, int64_t* ks, int64_t* K_out, void* v_T, void* f_T,
                      const void* tail, int64_t tail_rows, int flags, void* stream) {
  return vc2_compress_ex2(x, F, N, D, dtype, base_scale, map_mode, grid_h, gather_src, gather_rows, ws, ws_bytes, out_rows,
                          idx_out, cap, ks, K_out, v_T, f_T, tail, tail_rows, flags, nullptr, stream);
}

int64_t vc2_wait_host_count(const int64_t* K_host, double timeout_s) {
  // spins on the pinned mirror of K_out[0] (vc2_compress_ex2) until the selection launch has written it; < 0: timed out
  const volatile int64_t* p = K_host;
  if (!p) return -1;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t it = 0;; ++it) {
    const int64_t v = *p;
    if (v >= 0) { std::atomic_thread_fence(std::memory_order_acquire); return v; }
    if ((it & 1023) == 1023 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return -1;
  }
}

int vc2_compress_ex2(const void* x, int64_t F, int64_t N, int64_t D, int dtype, double base_scale, int map_mode,
                     int64_t grid_h, const void* gather_src, int64_t gather_rows, void* ws, size_t ws_bytes,
                     void* out_rows, int64_t* idx_out, int64_t cap, int64_t* ks, int64_t* K_out, void* v_T, void* f_T,
                     const void* tail, int64_t tail_rows, int flags, int64_t* K_host, void* stream) {
  if (!x || !idx_out || !ks || !K_out) return fail(VC2_ERR_ARG, "null pointer");
  if (tail_rows < 0 || (tail_rows > 0 && !tail)) return fail(VC2_ERR_ARG, "tail_rows without tail");
  if (N > 8192) return fail(VC2_ERR_UNSUPPORTED, "N=%lld > 8192 tokens per frame", (long long)N);
  if (map_mode == VC2_MAP_GRID_VID && (grid_h <= 0 || grid_h * grid_h != N))
    return fail(VC2_ERR_ARG, "grid_vid mapping needs N == grid_h^2");
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  if (D > 8192) return fail(VC2_ERR_UNSUPPORTED, "D=%lld > 8192 channels", (long long)D);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K_host) {       // the kernels store through this pointer: it must be pinned, device-mapped host memory -- translated
    //                    (not assumed equal: unified addressing is the common case, not a guarantee), once per pointer
    thread_local int64_t* seen_host = nullptr;
    thread_local int64_t* seen_dev = nullptr;
    if (seen_host != K_host) {
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, K_host, 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        return fail(VC2_ERR_ARG, "K_host is not pinned, device-mapped host memory (hipHostMalloc / pin_memory)");
      }
      seen_host = K_host; seen_dev = static_cast<int64_t*>(dp);
    }
    K_host = seen_dev;
  }
  float* var_f32 = wsp<float>(ws, p.o_var_f32);
  int* cols = wsp<int>(ws, p.o_cols);
  const int64_t kc = int64_t(double(D) * 0.5);            // int(x.shape[-1] * ratio), vidcom2.py:41
  const bool strict = cur_mode() && p.ES == 2;
  int* spos = strict ? wsp<int>(ws, p.o_spos) : nullptr;
  int* perm = strict ? wsp<int>(ws, p.o_perm) : nullptr;
  // (wperm / wcpos live in the o_tmp_f32 scratch, free until the selection stage: 2 * kc words <= R or D floats)
  uint32_t* wperm = strict && 2 * kc <= std::max<int64_t>(p.R, D) ? wsp<uint32_t>(ws, p.o_tmp_f32) : nullptr;
  uint32_t* wcpos = wperm ? wperm + kc : nullptr;
  int* const status = wsp<int>(ws, p.o_ticket) + kTkStatus;       // the pass's status word (-> K_out[1])
  // the variance reduction and the channel selection in ONE launch (k_var_select): 16-bit inputs, D <= 4096, this pass ran its
  // own sweep 1 (which left the variance array at "not written yet"), no per-kernel timing.  VC2_VARSEL=0: two launches
  const int varsel_env = [] { const char* e = getenv("VC2_VARSEL"); return e ? atoi(e) : 1; }();          // (read per pass, like VC2_S2_ORD)
  const int sel4_env2 = [] { const char* e = getenv("VC2_SEL4"); const char* e3 = getenv("VC2_SEL3"); return (e ? atoi(e) : 1) != 0 && !(e3 && atoi(e3) != 0); }();
  // (the selection workgroup WAITS inside the launch for the variance workgroups: they must be able to run beside it -- a
  //  device with a handful of CUs keeps the two launches; a stream whose CU mask leaves one CU must set VC2_VARSEL=0: the bounded
  //  wait would expire, and the pass would report status bit 2 instead of hanging)
  static const int n_cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev);
                                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
                                return n; }();
  if (varsel_env != 0 && n_cus >= 16 && sel4_env2 && p.ES == 2 && D <= 4096 && !(flags & VC2_FLAG_HAVE_STATS) && !g_prof && kc > 0 && kc < D) {
    if ((rc = launch_stats_sweep(p, x, ws, PoolSrc{}, st, reinterpret_cast<uint32_t*>(var_f32)))) return rc;
    const int64_t n_each = int64_t(p.BF) * p.N, n_last = p.R - int64_t(p.NB - 1) * n_each;
    VarSelArgs va{p.NB, n_each, n_last, int(D), var_f32, wsp<int>(ws, p.o_ticket),
                  PartSrc{wsp<double>(ws, p.o_part_stats), x, p.R, p.BF * p.stat_splits, p.G, int(p.N), p.BF},
                  make_fold_tab(p.NB, n_each, n_last), wsp<unsigned long long>(ws, p.o_nfixlist), int(p.R + cdiv(p.F, 2)),
                  reinterpret_cast<unsigned long long*>(K_out + 1), int(kc), cols, perm, wperm, wcpos, status,
                  [] { const char* e = getenv("VC2_CODE_WARM"); const int v = e ? atoi(e) : 88; return v < 0 ? 0 : (v > 88 ? 88 : v) * 1024; }()};
    const size_t smem = chan_select_lds(int(D));
    const unsigned nvb = unsigned(cdiv(D, 16));                    // workgroups of the variance reduction
    auto go = [&](auto kernel) -> int {
      int rca = allow_big_lds(kernel, smem, "k_var_select", 3 * kRedGL * 16 * 8);
      if (rca) return rca;
      hipLaunchKernelGGL(kernel, dim3(1 + nvb), dim3(kSelNT), smem, st, va);
      return check_launch("var_select");
    };
    if (p.dt == VC2_BF16) rc = go(k_var_select<VC2_BF16>); else rc = go(k_var_select<VC2_F16>);
    if (rc) return rc;
  } else {
    if ((rc = launch_chan_stats(p, x, ws, nullptr, nullptr, var_f32, st, /*zero_queue_counters=*/true,
                                /*have_partials=*/(flags & VC2_FLAG_HAVE_STATS) != 0, /*kstatus=*/K_out + 1)))
      return rc;
    if ((rc = launch_chan_select(var_f32, D, kc, nullptr, cols, perm, st, wperm, wcpos, status, /*words64_expected=*/p.ES == 4))) return rc;
  }
  const ChanSet cs = make_chanset(p, cols, spos, kc);
  // torch.topk's ORDER of the selected channels (needed only by the "torch order" fix-ups, which run after sweep 2)
  // is replayed by a rider workgroup of sweep 2 itself
  OrderArgs rider{};
  if (strict) rider = OrderArgs{var_f32, perm, cols, wsp<int>(ws, p.o_order), wsp<int>(ws, p.o_opos), spos, int(D), int(kc), 1,
                                wperm, wcpos, status};
  if ((rc = launch_phase1(p, x, cs, ws, true, st, rider, /*own_stats=*/true))) return rc;
  float* total = wsp<float>(ws, p.o_total);
  float* scales = wsp<float>(ws, p.o_scales_f32);
  const double bs = base_scale < 0 ? 0.0 : base_scale;
  const bool fused_budget = F <= kFusedScalesMaxF;
  // budgets: every k_select workgroup derives them itself from sweep 3's workgroup partials (F <= 1024; saves two
  // kernel boundaries).  Fusing them into sweep 3's last workgroup through an agent-scope ticket measured SLOWER than
  // a separate kernel -- the release fences write back L2.
  float* s = fused_budget ? nullptr : wsp<float>(ws, p.o_s);
  if ((rc = launch_phase2(p, x, cs, ws, v_T, f_T, total, s, st))) return rc;
  if (fused_budget) {
    rc = launch_select(dtype, total, F, 0, F, N, map_mode, grid_h, ks, wsp<int64_t>(ws, p.o_offs), idx_out, cap, K_out,
                       BudgetSrc{nullptr, wsp<double>(ws, p.o_vpart), p.S2, nullptr, bs, 0.01, scales, 0,
                                 wsp<int>(ws, p.o_ticket) + kTkStatus, reinterpret_cast<long long*>(K_host),
                                 wsp<int>(ws, p.o_ticket) + kTkSelArrive}, st);
  } else {
    if ((rc = launch_scales(dtype, s, F, bs, 0.01, wsp<float>(ws, p.o_zbuf), scales, nullptr, st))) return rc;
    rc = launch_select(dtype, total, F, 0, F, N, map_mode, grid_h, ks, wsp<int64_t>(ws, p.o_offs), idx_out, cap, K_out,
                       BudgetSrc{scales, nullptr, 0, nullptr, bs, 0.01, nullptr, 0, wsp<int>(ws, p.o_ticket) + kTkStatus,
                                 reinterpret_cast<long long*>(K_host), wsp<int>(ws, p.o_ticket) + kTkSelArrive}, st);
  }
  if (rc) return rc;
  if (out_rows && gather_src)
    rc = launch_gather_rows(gather_src, gather_rows, D, p.ES, idx_out, K_out, cap, out_rows, st, tail, tail_rows);
  return rc;
}

int vc2_select_sharded(const float* total_f32, const float* s_all_f32, int64_t F_total, int64_t f0,
                       int64_t F_local, int64_t N, int64_t D, double base_scale, int dtype, void* ws,
                       size_t ws_bytes, int64_t* ks, int64_t* idx_out, int64_t cap, int64_t* K_out,
                       const void* gather_src, void* out_rows, void* stream) {
  if (!total_f32 || !s_all_f32 || !ks || !idx_out || !K_out) return fail(VC2_ERR_ARG, "null pointer");
  if (F_total <= 0 || F_local <= 0 || f0 < 0 || f0 + F_local > F_total || F_total > kMaxFramesTotal)
    return fail(VC2_ERR_ARG, "bad frame range [%lld, %lld) of %lld", (long long)f0, (long long)(f0 + F_local),
                (long long)F_total);
  if (N > 8192) return fail(VC2_ERR_UNSUPPORTED, "N=%lld > 8192 tokens per frame", (long long)N);
  Plan p;
  int rc = make_plan(F_local, N, D, dtype, &p, F_total);
  if (rc) return rc;
  if ((rc = need_ws(p, ws, ws_bytes))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* scales = wsp<float>(ws, p.o_scales_f32);
  // budgets over ALL frames of the video (softmax + mean are global, vidcom2.py:66-67), selection only for this
  // rank's frames [f0, f0 + F_local)
  if (F_total <= kFusedScalesMaxF) {
    rc = launch_select(dtype, total_f32, F_total, f0, F_local, N, VC2_MAP_LINEAR, 0, ks, wsp<int64_t>(ws, p.o_offs),
                       idx_out, cap, K_out, BudgetSrc{nullptr, nullptr, 0, s_all_f32, base_scale, 0.01, nullptr}, st);
  } else {
    if ((rc = launch_scales(dtype, s_all_f32, F_total, base_scale, 0.01, wsp<float>(ws, p.o_zbuf), scales, nullptr, st)))
      return rc;
    rc = launch_select(dtype, total_f32, F_total, f0, F_local, N, VC2_MAP_LINEAR, 0, ks, wsp<int64_t>(ws, p.o_offs),
                       idx_out, cap, K_out, BudgetSrc{scales, nullptr, 0, nullptr, base_scale, 0.01, nullptr}, st);
  }
  if (rc) return rc;
  {   // K_out[2] = number of video-centre columns left at the exactly rounded mean (see vc2_scores_phase2)
    if (hipMemsetAsync(K_out + 2, 0, 8, st) != hipSuccess ||
        hipMemcpyAsync(K_out + 2, wsp<int>(ws, p.o_ticket) + 5, 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(VC2_ERR_LAUNCH, "status copy failed");
  }
  if (out_rows && gather_src)
    rc = launch_gather_rows(gather_src, F_local * N, D, p.ES, idx_out, K_out, cap, out_rows, st);
  return rc;
}

int vc2_multi_scale_gaussian(const void* x, int64_t F, int64_t N, int64_t C, int dtype, const void* centre,
                             int64_t n_centres, const double* alphas, int n_alphas, void* out_T, void* stream) {
  if (!x || !centre || !alphas || !out_T) return fail(VC2_ERR_ARG, "null pointer");
  if (dtype < 0 || dtype > 2) return fail(VC2_ERR_ARG, "unknown dtype code %d", dtype);
  if (F < 0 || N <= 0 || C <= 0) return fail(VC2_ERR_SHAPE, "bad shape F=%lld N=%lld C=%lld", (long long)F,
                                              (long long)N, (long long)C);
  if (n_centres != 1 && n_centres != F) return fail(VC2_ERR_SHAPE, "centre must have 1 or F rows");
  if (n_alphas < 1 || n_alphas > kMaxAlphas) return fail(VC2_ERR_UNSUPPORTED, "1..%d scales", kMaxAlphas);
  if (C > 8192) return fail(VC2_ERR_UNSUPPORTED, "C <= 8192");
  if (F == 0) return VC2_OK;
  MsgAlphas al;
  al.n = n_alphas;
  for (int i = 0; i < n_alphas; ++i) al.two_a[i] = float(2.0 * alphas[i]);   // python: -d / (2 * a), scalar -> fp32
  const int64_t R = F * N;
  const unsigned grid = unsigned(std::min<int64_t>(cdiv(R, kMsgWaves), 65535 * 16));
  const size_t smem = cur_mode() ? size_t(kMsgWaves) * size_t(C) * 4 : 0;
  VC2_DISPATCH_DT(dtype, {
    if (smem > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_multi_scale_gaussian<DT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
    hipLaunchKernelGGL((k_multi_scale_gaussian<DT>), dim3(grid), dim3(kMsgWaves * 64), smem,
                       static_cast<hipStream_t>(stream), x, R, int(C), centre, n_centres == 1 ? 0 : 1, int(N), al,
                       cur_mode() ? 1 : 0, out_T);
  });
  return check_launch("multi_scale_gaussian");
}

int vc2_kat_exp(const void* in_T, int64_t n, int dtype, void* out_T, void* stream) {
  if (!in_T || !out_T) return fail(VC2_ERR_ARG, "null pointer");
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_kat_exp<DT>), dim3(unsigned(cdiv(n, 256))), dim3(256), 0,
                                            static_cast<hipStream_t>(stream), in_T, n, out_T));
  return check_launch("kat_exp");
}
int vc2_kat_round(const float* in, int64_t n, int dtype, void* out_T, void* stream) {
  if (!in || !out_T) return fail(VC2_ERR_ARG, "null pointer");
  VC2_DISPATCH_DT(dtype, hipLaunchKernelGGL((k_kat_round<DT>), dim3(unsigned(cdiv(n, 256))), dim3(256), 0,
                                            static_cast<hipStream_t>(stream), in, n, out_T));
  return check_launch("kat_round");
}

int vc2_pass_counters(int64_t F, int64_t N, int64_t D, int dtype, const void* ws, int32_t* out8) {
  // diagnostic: the strict-mode queue counters of the last pass that used `ws` (synchronises the device)
  if (!ws || !out8) return fail(VC2_ERR_ARG, "null pointer");
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(out8, static_cast<const char*>(ws) + p.o_ticket, 32, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(VC2_ERR_LAUNCH, "counter read-back failed");
  return VC2_OK;
}

int vc2_selftest_force_guard(int frame) {
  g_force_guard.store(frame, std::memory_order_relaxed);
  return VC2_OK;
}
int vc2_selftest_counters(int32_t* out8_host, int reset) {
  // diagnostic (SYNCHRONISES): how often a loop bound of the selection engine expired (must be all zero)
  if (!out8_host) return fail(VC2_ERR_ARG, "null pointer");
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(vc2::g_sel2_guard_hits), 32) != hipSuccess)
    return fail(VC2_ERR_LAUNCH, "counter read-back failed");
  if (reset) { int z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(vc2::g_sel2_guard_hits), z, 32); }
  return VC2_OK;
}

int vc2_selftest_ord_pieces(int64_t F, int64_t N, int64_t D, int dtype, int32_t* out5_host, int64_t cap) {
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  if (p.ord_m == 0) return 0;
  const OrdGeo g{p.ord_S, p.ord_nA};
  const int n = ord_total_wgs(g, int(F));
  if (out5_host) {
    for (int b = 0; b < n && b < cap; ++b) {
      const OrdPiece o = ord_piece(g, b, int(N >> 4));
      int32_t* r = out5_host + int64_t(b) * 5;
      r[0] = o.f; r[1] = o.j; r[2] = o.Sf; r[3] = o.b0; r[4] = o.nb;
    }
  }
  return n;
}

int vc2_profile_enable(int on) {
  if (on && !g_prof) {
    for (int i = 0; i < KID_COUNT; ++i) { g_prof_ms[i] = 0.0; g_prof_n[i] = 0; }
  }
  g_prof = on != 0;
  return VC2_OK;
}

int vc2_profile_collect(int max_kernels, const char** names, double* total_ms, int64_t* launches) {
  // synchronises on the recorded events, accumulates and frees them
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_prof_ms[r.id] += double(ms);
      g_prof_n[r.id] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof_recs.clear();
  const int n = max_kernels < KID_COUNT ? max_kernels : int(KID_COUNT);
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = kKernelNames[i];
    if (total_ms) total_ms[i] = g_prof_ms[i];
    if (launches) launches[i] = g_prof_n[i];
  }
  return n;
}

#ifdef VC2_DEBUG_EXPORTS     // (scripts/dev/*: where a plan keeps its arrays, run-time switches of kernel variants)
int vc2_debug_plan(int64_t F, int64_t N, int64_t D, int dtype, int64_t* out /*[16]*/) {
  Plan p;
  int rc = make_plan(F, N, D, dtype, &p);
  if (rc) return rc;
  const int64_t v[16] = {int64_t(p.o_den), int64_t(p.o_part_col), int64_t(p.o_rflag), int64_t(p.o_ticket), int64_t(p.o_nfixlist),
                         p.S, p.S_q, p.S_W, int64_t(p.o_fc), int64_t(p.o_vc), int64_t(p.o_total), int64_t(p.o_cols), p.S2, int64_t(p.o_vpart), 0, 0};
  for (int i = 0; i < 16; ++i) out[i] = v[i];
  return VC2_OK;
}
int vc2_debug_set(int key, int value) {
  if (key == 0) g_s2v2.store(value, std::memory_order_relaxed);
  if (key == 1) g_s3v2.store(value, std::memory_order_relaxed);
  return VC2_OK;
}
#endif
#ifdef VC2_DEBUG_TIMING
int vc2_debug_wg(unsigned long long* out /*[8][2][4096]*/) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(vc2::g_dbg_wg), sizeof(unsigned long long) * 8 * 2 * 4096);
  return 0;
}
int vc2_debug_vc(unsigned long long* out) {      // [6][4096]: k_video_centre's per-wave stamps (scripts/dev/vc_waves.py)
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vc2::g_dbg_vc), sizeof(unsigned long long) * 6 * 4096) == hipSuccess ? 0 : -1;
}
int vc2_debug_read_rounds(unsigned long long* t, int* v) {      // [2][128] each: the per-round stamps (VC2_ROUND)
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(vc2::g_dbg_rt), sizeof(unsigned long long) * 256);
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(vc2::g_dbg_rv), sizeof(int) * 256);
  return 0;
}
int vc2_debug_read(unsigned long long* t, int* v, int* n, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(vc2::g_dbg_n), sizeof(int));
  (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(vc2::g_dbg_t), sizeof(unsigned long long) * 512);
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(vc2::g_dbg_v), sizeof(int) * 512);
  if (reset) { int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(vc2::g_dbg_n), &z, sizeof(int)); }
  return 0;
}
#endif

}  // extern "C"
#else
namespace {
VC2_DEV_ONLY
}
#endif
