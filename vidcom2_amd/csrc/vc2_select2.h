// vc2_select2.h -- low-latency replay of the selection torch.topk performs on CPU (second generation).
//
// The reference decides which tokens / channels are kept with torch.topk(largest=False) (vidcom2.py:42 and
// :76), i.e. libstdc++ std::nth_element / std::partial_sort / std::sort on (value, index) pairs
// (ATen/native/TopKImpl.h).  In bf16 / fp16 the kept SET is decided by how those
// algorithms permute ties, so they are replayed step by step.  What is parallel in them is one Hoare partition
// (__unguarded_partition): its i-th swap pairs the i-th element from the left that is !(a < pivot) with the i-th
// from the right that is !(pivot < a) while the left position is below the right one -- two prefix counts, a
// rank -> position scatter and independent swaps reproduce the serial loop's permutation exactly.
//
// A partition ROUND here is built for latency (the rounds are a serial chain of 13 + ~12 at D = 3584):
//   * an element is ONE word: (order-preserving key | index).  Half-precision values widened to fp32 have 13 zero
//     low bits, so key and index share 32 bits (W = uint32_t); arbitrary fp32 values take 64 (W = uint64_t);
//   * a thread owns CONSECUTIVE positions of the current range (re-blocked every round: 4, 8, .. 32 elements
//     per thread), fetched with 16-byte LDS reads; its ranks are a running count on top of one DPP wave scan;
//   * long ranges are partitioned by 4 cooperating waves (3 workgroup barriers per round), ranges of at most
//     2048 elements by ONE wave with no barrier at all (a wave's LDS accesses execute in order);
//   * the cut needs no search: it is the minimum of "partner position" over the swapped left elements and
//     "own position" over the unswapped ones -- a DPP wave minimum (+ one LDS atomic-min across waves).
// The rarely taken pieces (heap-select fallback at the depth limit, partial_sort for k*64 <= n, the final
// insertion sort of <= 3 elements) run serially on one lane, replaying libstdc++ statement by statement.
#pragma once

#include "vc2_device.h"

// debug builds (-DVC2_DEBUG_TIMING): wall-clock stamps per selection round of workgroup 0 (scripts/dbg_timing.py)
#if defined(VC2_DEBUG_TIMING) && defined(VC2_STAMP)
#define VC2_SEL_STAMP(tag) do { if (blockIdx.x == 0) VC2_STAMP(tag); } while (0)
#else
#define VC2_SEL_STAMP(tag) ((void)0)
#endif
// ... and cheap per-ROUND stamps of the two selection kernels (no atomic: a counter in the caller's Sel2, one store per
// stamp): slot 0 = k_chan_select, slot 1 = k_select's frame 0; value = tag | range length << 12
// (scripts/dev/chain_rounds.py -> profiles/r06_chan_select_rounds.csv)
#if defined(VC2_DEBUG_TIMING)
namespace vc2 {
__device__ unsigned long long g_dbg_rt[2][128];
__device__ int g_dbg_rv[2][128];
}
#define VC2_ROUND(S, tag, len) do { if ((S).dbg_slot >= 0 && (S).dbg_i < 127) { \
    g_dbg_rt[(S).dbg_slot][(S).dbg_i] = wall_clock64(); g_dbg_rv[(S).dbg_slot][(S).dbg_i] = int(tag) | (int(len) << 12); \
    ++(S).dbg_i; g_dbg_rv[(S).dbg_slot][127] = (S).dbg_i; } } while (0)
#define VC2_ROUND_RAW(slot, i, tag) do { g_dbg_rt[slot][i] = wall_clock64(); g_dbg_rv[slot][i] = int(tag); } while (0)
#else
#define VC2_ROUND(S, tag, len) ((void)0)
#define VC2_ROUND_RAW(slot, i, tag) ((void)0)
#endif

namespace vc2 {

// ---- element words -------------------------------------------------------------------------------
template <typename W> struct WordTr;
template <> struct WordTr<uint32_t> {
  static constexpr int kIdxBits = 13;                        // n <= 8192
  static __device__ __forceinline__ uint32_t key(uint32_t w) { return w >> kIdxBits; }
  static __device__ __forceinline__ int idx(uint32_t w) { return int(w & ((1u << kIdxBits) - 1u)); }
  // key32 = topk_key(value); the low 13 bits carry no order information for widened 16-bit values (all equal)
  static __device__ __forceinline__ uint32_t pack(uint32_t key32, int i) {
    return (key32 & ~((1u << kIdxBits) - 1u)) | uint32_t(i);
  }
};
template <> struct WordTr<uint64_t> {
  static __device__ __forceinline__ uint32_t key(uint64_t w) { return uint32_t(w >> 32); }
  static __device__ __forceinline__ int idx(uint64_t w) { return int(uint32_t(w)); }
  static __device__ __forceinline__ uint64_t pack(uint32_t key32, int i) { return (uint64_t(key32) << 32) | uint32_t(i); }
};
// can the values (fp32-widened) be packed into 32-bit words?  (a property of the whole array)
__device__ __forceinline__ bool key_fits_u32(float f) {
  return (f != f) || (__float_as_uint(f) & 0x1FFFu) == 0u;
}

// self-check: every loop of the engine is bounded; a bound that actually expires is counted here (read back by
// vc2_selftest_counters; the test-suite asserts zeros)
__device__ int g_sel2_guard_hits[8];
// A hit is also reported by the pass it happened in: bit kSel2StatusGuard of the pass's status word (Sel2::status, the
// workspace ticket the launching entry point hands down; nullptr for stand-alone stage calls) -> K_out[1] bit 2.  (Until
// round 4 this was ONE device-global flag, read and cleared by whichever pass's k_select came first: with two passes in
// flight on two streams a hit in clip A could be reported by clip B.)
constexpr int kSel2StatusGuard = 2;
__device__ __forceinline__ void guard_hit(int which, int* status) {
  atomicAdd(&g_sel2_guard_hits[which], 1);
  if (status) atomicOr(status, kSel2StatusGuard);
}

#ifdef VC2_SEL2_DEBUG
__device__ unsigned long long g_sel2_dbg[128];
#endif

constexpr int kXchCut = 16, kXchDummy = 20;
constexpr int kSel2Pad = 64;           // positions a round may read past the end of the range (never used)
constexpr int kSel2XchBytes = 512;     // 128 words: [0, 24) the LDS rounds' cells; [32, 128) the register rounds' (kX3*)

template <typename W> struct Sel2 {
  W* w;            // [n + kSel2Pad]
  uint16_t* la;    // [n + kSel2Pad] left-stop positions by rank
  uint16_t* lb;    // [n + kSel2Pad] right-stop positions by rank (numbered from the left)
  uint32_t* xch;   // [24] cross-wave exchange: [0,16) per-wave totals, [16] cut, [20..21] dummy cells (predicated stores)
  int dumw;        // index of a dummy word in w[] (predicated swaps of no-op elements land there)
  int* status;     // the pass's status word (guard_hit), or nullptr
#if defined(VC2_DEBUG_TIMING)
  int dbg_slot;    // per-round stamps (VC2_ROUND): -1 none
  mutable int dbg_i;
#endif
};
__host__ __device__ inline size_t sel2_bytes(int n, int wbytes) {
  return (size_t(n + kSel2Pad) * size_t(wbytes) + 15) / 16 * 16 + size_t(n + kSel2Pad) * 2 * 2 + kSel2XchBytes + 32;
}
template <typename W> __device__ __forceinline__ Sel2<W> sel2_carve(unsigned char* smem, int n, int* status = nullptr) {
  Sel2<W> S;
  S.status = status;
#if defined(VC2_DEBUG_TIMING)
  S.dbg_slot = -1; S.dbg_i = 0;
#endif
  const size_t wb = (size_t(n + kSel2Pad) * sizeof(W) + 15) / 16 * 16;
  S.w = reinterpret_cast<W*>(smem);
  unsigned char* p = smem + wb;
  S.xch = reinterpret_cast<uint32_t*>(p);
  S.la = reinterpret_cast<uint16_t*>(p + kSel2XchBytes);
  S.lb = S.la + (n + kSel2Pad);
  S.dumw = n + kSel2Pad - 1;
  return S;
}

// ---- wave primitives (DPP: VALU latency, no LDS crossbar round trip) -------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, false));   // row_shr:1
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, false));   // row_shr:2
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, false));   // row_shr:4
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, false));   // row_shr:8 -> scan inside each row of 16
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1, 3
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2, 3
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
  const uint32_t t = uint32_t(__builtin_amdgcn_update_dpp(-1, int(v), CTRL, ROW_MASK, 0xF, false));
  return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wave_min_bcast_u32(uint32_t v) {
  v = dpp_min_u32<0xB1, 0xF>(v);
  v = dpp_min_u32<0x4E, 0xF>(v);
  v = dpp_min_u32<0x141, 0xF>(v);
  v = dpp_min_u32<0x140, 0xF>(v);
  v = dpp_min_u32<0x142, 0xA>(v);
  v = dpp_min_u32<0x143, 0xC>(v);
  return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}
__device__ __forceinline__ void wave_lds_order() {     // single wave: the DS queue is in order; pin the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int NW> __device__ __forceinline__ void sel2_sync() {
  if constexpr (NW > 1) __syncthreads(); else wave_lds_order();
}

// ---- serial libstdc++ pieces (one lane) ------------------------------------------------------------
template <typename W> __device__ __forceinline__ bool w_less(W a, W b) { return WordTr<W>::key(a) < WordTr<W>::key(b); }

// bits/stl_heap.h __adjust_heap (+ inlined __push_heap)
template <typename W>
__device__ __forceinline__ void s2_adjust_heap(W* w, int first, int hole, int len, W v) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (w_less(w[first + child], w[first + child - 1])) child--;
    w[first + hole] = w[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    w[first + hole] = w[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && w_less(w[first + parent], v)) {
    w[first + hole] = w[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  w[first + hole] = v;
}
// bits/stl_algo.h __heap_select(first, middle, last)
template <typename W>
__device__ __forceinline__ void s2_heap_select(W* w, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {   // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const W v = w[first + parent];
      s2_adjust_heap(w, first, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i) {
    if (w_less(w[i], w[first])) {   // __pop_heap(first, middle, i)
      const W v = w[i];
      w[i] = w[first];
      s2_adjust_heap(w, first, 0, len, v);
    }
  }
}
// bits/stl_heap.h __sort_heap(first, last) on a heap
template <typename W>
__device__ __forceinline__ void s2_sort_heap(W* w, int first, int last) {
  while (last - first > 1) {
    --last;
    const W v = w[last];
    w[last] = w[first];
    s2_adjust_heap(w, first, 0, last - first, v);
  }
}
// bits/stl_algo.h __insertion_sort
template <typename W>
__device__ __forceinline__ void s2_insertion_sort(W* w, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const W v = w[i];
    if (w_less(v, w[first])) {
      for (int j = i; j > first; --j) w[j] = w[j - 1];
      w[first] = v;
    } else {   // __unguarded_linear_insert
      int l = i, nx = i - 1;
      while (w_less(v, w[nx])) { w[l] = w[nx]; l = nx; --nx; }
      w[l] = v;
    }
  }
}

// ---- one __unguarded_partition_pivot(lo, hi) by NW cooperating waves ---------------------------------
// (hi - lo > 3.)  tid = 0 .. 64*NW-1 inside the group; every thread of the group calls this with identical
// arguments; for NW > 1 the group must be the whole workgroup (__syncthreads).  la / lb: rank scratch with room
// for hi - lo entries that no concurrent partition uses.  The thread's 4*EQ consecutive elements live in registers
// for the whole round (4*EQ*64*NW >= hi - lo + 3).  Returns the cut (the same value in every thread).
// BRANCH-FREE: every LDS access is issued unconditionally, "not mine" cases redirected to a dummy cell or a clamped
// index -- an exec-masked branch per element per step costs ~50 cycles, and a lone wave issues one instruction
// every ~4 cycles, so the instruction count IS the round time (measured: 350 instructions = 1500 cycles at EQ = 1).
template <typename W, int NW, int EQ>
__device__ __forceinline__ int sel2_partition_t(const Sel2<W>& S, int lo, int hi, uint16_t* la, uint16_t* lb, int tid) {
  using T = WordTr<W>;
  constexpr int E = EQ == 0 ? 1 : 4 * EQ, NT = 64 * NW;        // EQ = 0: ONE element per thread (ranges <= 64*NW)
  const int lane = tid & 63, wave = tid >> 6;
  const int first = lo + 1;
  const int base = EQ == 0 ? first : (first & ~3);           // 16-byte aligned blocking of [base, hi)
  const int er = EQ == 0 ? 1 : ((((hi - base) + NT - 1) / NT + 3) & ~3);   // elements per thread this round (<= E)
  const int p0 = base + tid * er;
  W el[E];
  if constexpr (EQ == 0) {
    el[0] = S.w[p0 < hi ? p0 : lo];
  } else {
    constexpr int WPV = 16 / int(sizeof(W));                 // words per 16-byte read
    const uint4* src = reinterpret_cast<const uint4*>(S.w + (p0 < hi ? p0 : (hi & ~3)));   // (idle threads: any valid quad)
    union { uint4 v; W e[WPV]; } u;
#pragma unroll
    for (int q = 0; q < E / WPV; ++q) {                       // (reads past the range stay inside the padded array)
      u.v = src[q];
#pragma unroll
      for (int t = 0; t < WPV; ++t) el[q * WPV + t] = u.e[t];
    }
  }
  const int pa = lo + 1, pb = lo + (hi - lo) / 2, pc = hi - 1;
  const W wlo = S.w[lo], wa = S.w[pa], wb = S.w[pb], wc = S.w[pc];
  int msrc;
  W wp;
  {
    const uint32_t ka = T::key(wa), kb = T::key(wb), kc = T::key(wc);
    const bool ab = ka < kb, bc = kb < kc, ac = ka < kc;
    const int sel = ab ? (bc ? 1 : (ac ? 2 : 0)) : (ac ? 0 : (bc ? 2 : 1));
    msrc = sel == 0 ? pa : (sel == 1 ? pb : pc);
    wp = sel == 0 ? wa : (sel == 1 ? wb : wc);
  }
  const uint32_t pk = T::key(wp);
  uint32_t mA = 0u, mB = 0u, cA = 0u, cB = 0u;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int p = p0 + j;
    el[j] = (p == msrc) ? wlo : el[j];                         // iter_swap(lo, msrc), register copy
    const uint32_t k = T::key(el[j]);
    const bool in = j < er && p >= first && p < hi;
    const uint32_t a = (in && k >= pk) ? 1u : 0u, b = (in && k <= pk) ? 1u : 0u;
    mA |= a << j; mB |= b << j;
    cA += a; cB += b;
  }
  const uint32_t packed = cA | (cB << 16);
  const uint32_t incl = wave_incl_scan_u32(packed);
  const uint32_t wtot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
  uint32_t bas = incl - packed, tot = wtot;
  if constexpr (NW > 1) {
    if (lane == 63) S.xch[wave] = incl;
    __syncthreads();                                         // also: every trip-1 read is done before the swaps
    uint32_t pre = 0u, all = 0u;
#pragma unroll
    for (int v = 0; v < NW; ++v) { const uint32_t t = S.xch[v]; pre += v < wave ? t : 0u; all += t; }
    bas += pre; tot = all;
  }
  if (tid == 0) {                                            // iter_swap(lo, msrc), LDS copy
    S.w[lo] = wp;
    S.w[msrc] = wlo;
    if constexpr (NW > 1) S.xch[kXchCut] = 0xFFFFFFFFu;
  }
  const int totA = int(tot & 0xFFFFu), totB = int(tot >> 16);
  uint16_t* const dum16 = reinterpret_cast<uint16_t*>(S.xch + kXchDummy);
  {
    int rA = int(bas & 0xFFFFu), rB = int(bas >> 16);
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const bool A = (mA >> j) & 1u, B = (mB >> j) & 1u;
      uint16_t* da = A ? la + rA : dum16;
      uint16_t* db = B ? lb + rB : dum16 + 1;
      *da = uint16_t(p0 + j);
      *db = uint16_t(p0 + j);
      rA += A ? 1 : 0; rB += B ? 1 : 0;
    }
  }
  sel2_sync<NW>();
  uint32_t cand = 0xFFFFFFFFu;
  {
    int rA = int(bas & 0xFFFFu), rB = int(bas >> 16);
    constexpr int CH = E < 8 ? E : 8;                        // lookups in flight together (bounded: registers)
#pragma unroll
    for (int j0 = 0; j0 < E; j0 += CH) {
      int qa[CH], qb[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j0 + c;
        const bool A = (mA >> j) & 1u, B = (mB >> j) & 1u;
        const int ia = totB - 1 - rA, ib = totB - 1 - rB;      // A: partner = lb_right[rA]; B: own right rank
        const bool okA = A && rA < totB, okB = B && ib < totA;
        const int ra = int(lb[okA ? ia : 0]), rb = int(la[okB ? ib : 0]);
        qa[c] = okA ? ra : -1;
        qb[c] = okB ? rb : 0x7FFFFFFF;
        rA += A ? 1 : 0; rB += B ? 1 : 0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j0 + c;
        const int p = p0 + j;
        const bool A = (mA >> j) & 1u;
        const bool vA = qa[c] > p;                            // swapped as a left element (qa = -1 otherwise)
        const bool vB = !vA && qb[c] < p;                     // swapped as a right element (qb = INT_MAX otherwise)
        const uint32_t cc = A ? (vA ? uint32_t(qa[c]) : uint32_t(p)) : 0xFFFFFFFFu;
        cand = cc < cand ? cc : cand;
        S.w[vA ? qa[c] : (vB ? qb[c] : S.dumw)] = el[j];
      }
    }
  }
  uint32_t cutv = wave_min_bcast_u32(cand);
  if constexpr (NW > 1) {
    if (lane == 0) atomicMin(&S.xch[kXchCut], cutv);
    __syncthreads();
    cutv = S.xch[kXchCut];
  } else {
    wave_lds_order();
  }
  return cutv < uint32_t(hi) ? int(cutv) : hi;
}

// dispatch on the elements per thread the range needs; the instantiated sizes are EQLO .. EQHI (powers of two)
template <typename W, int NW, int EQLO, int EQHI>
__device__ __forceinline__ int sel2_partition(const Sel2<W>& S, int lo, int hi, uint16_t* la, uint16_t* lb, int tid) {
  constexpr int NT = 64 * NW;
  const int q = (((hi - ((lo + 1) & ~3)) + NT - 1) / NT + 3) >> 2;      // quads per thread
  if constexpr (EQLO == 0) { if (hi - lo - 1 <= NT) return sel2_partition_t<W, NW, 0>(S, lo, hi, la, lb, tid); }
  if constexpr (EQLO <= 1 && EQHI > 1) { if (q <= 1) return sel2_partition_t<W, NW, 1>(S, lo, hi, la, lb, tid); }
  if constexpr (EQLO <= 2 && EQHI > 2) { if (q <= 2) return sel2_partition_t<W, NW, 2>(S, lo, hi, la, lb, tid); }
  if constexpr (EQLO <= 4 && EQHI > 4) { if (q <= 4) return sel2_partition_t<W, NW, 4>(S, lo, hi, la, lb, tid); }
  return sel2_partition_t<W, NW, EQHI>(S, lo, hi, la, lb, tid);
}
// the longest range a group of NW waves partitions with 4*EQ elements per thread
__host__ __device__ constexpr int sel2_capacity(int nw, int eq) { return 64 * nw * 4 * eq - 4; }

// ---- the last rounds of introselect in REGISTERS (ranges of at most 64 elements, one wave) -----------------------
// An LDS round costs a lone wave ~0.6 us whatever the range (three dependent LDS trips, two DPP scans, ~350
// instructions), and the last six or seven of the ~12 rounds of a 3584-element selection work on <= 64 elements.  Here
// the range lives one element per lane for all those rounds and a round needs no memory at all:
//   * the pivot's three candidates come by v_readlane, the median is scalar code;
//   * the two stop sets of __unguarded_partition are 64-bit ballots: A = "not less than the pivot", B = "not greater";
//     with rA = #A left of me (v_mbcnt) and Bge = #B at or right of me, the serial loop's i-th swap pairs the i-th A
//     from the left with the i-th B from the right while the A stands left of the B, i.e. an A element is swapped
//     iff more than rA B's stand strictly right of it and a B element iff at least Bge A's stand strictly left of it
//     -- decided per lane, no positions looked up;
//   * the swapped elements change lanes through rank space: A number r sends itself to slot r, B number s (from the
//     right) to slot 63 - s (ds_permute_b32: the LDS crossbar, no LDS memory), then A number r fetches slot 63 - r and
//     B number s slot s (ds_bpermute_b32).  At most 31 pairs: slot 31 stays free for everybody else;
//   * the cut is where std::__unguarded_partition's `first` halts: the first A that is not swapped, or the last
//     (= leftmost) swapped B if that comes first -- two s_ff1 on ballots.
// A round is ~40 instructions and two crossbar trips.  Depth limit, the final insertion sort of <= 3 elements and the
// heap-select fallback are libstdc++'s (stl_algo.h __introselect).  (A first version replayed the swaps one by one
// in scalar code -- v_readlane / select pairs, ~150 cycles per swap: no faster than the LDS rounds.)
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }
__device__ __forceinline__ uint64_t rdlane(uint64_t v, int l) {
  return uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), l))) |
         (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), l))) << 32);
}
template <typename W> __device__ __forceinline__ W wrlane(W old, W val, int l, int lane) {
  return lane == l ? val : old;
}
__device__ __forceinline__ uint32_t xbar_push(int slot, uint32_t v) { return uint32_t(__builtin_amdgcn_ds_permute(slot << 2, int(v))); }
__device__ __forceinline__ uint64_t xbar_push(int slot, uint64_t v) {
  return uint64_t(xbar_push(slot, uint32_t(v))) | (uint64_t(xbar_push(slot, uint32_t(v >> 32))) << 32);
}
__device__ __forceinline__ uint32_t xbar_pull(int slot, uint32_t v) { return uint32_t(__builtin_amdgcn_ds_bpermute(slot << 2, int(v))); }
__device__ __forceinline__ uint64_t xbar_pull(int slot, uint64_t v) {
  return uint64_t(xbar_pull(slot, uint32_t(v))) | (uint64_t(xbar_pull(slot, uint32_t(v >> 32))) << 32);
}
constexpr int kSel2TailMax = 64;

// finishes std::__introselect on S.w[lo, hi) (hi - lo <= 64) for position nth, `depth` partitions left before the
// heap-select fallback; whole wave (all 64 lanes active), identical arguments in every lane
template <typename W>
__device__ __forceinline__ void introselect_tail64(const Sel2<W>& S, int lo_, int hi_, int nth_, int depth_, int lane) {
  using T = WordTr<W>;
  const int lo = __builtin_amdgcn_readfirstlane(lo_), hi = __builtin_amdgcn_readfirstlane(hi_);
  const int nr = __builtin_amdgcn_readfirstlane(nth_) - lo;
  int depth = __builtin_amdgcn_readfirstlane(depth_);
  const int n = hi - lo;
  W el = S.w[lo + (lane < n ? lane : 0)];
  int l = 0, h = n;
  bool fallback = false;
  for (int guard = 0; h - l > 3 && guard < 256; ++guard) {
    if (depth == 0) { fallback = true; break; }
    --depth;
    if (lane == 0) VC2_ROUND(S, 250, h - l);
    const int pa = l + 1, pb = l + (h - l) / 2, pc = h - 1;
    const W wlo = rdlane(el, l), wa = rdlane(el, pa), wb = rdlane(el, pb), wc = rdlane(el, pc);
    const uint32_t ka = T::key(wa), kb = T::key(wb), kc = T::key(wc);
    const bool ab = ka < kb, bc = kb < kc, ac = ka < kc;          // __move_median_to_first
    const int sel = ab ? (bc ? 1 : (ac ? 2 : 0)) : (ac ? 0 : (bc ? 2 : 1));
    const int msrc = sel == 0 ? pa : (sel == 1 ? pb : pc);
    const W wp = sel == 0 ? wa : (sel == 1 ? wb : wc);
    el = lane == l ? wp : (lane == msrc ? wlo : el);              // iter_swap(first, median)
    const uint32_t pk = T::key(wp), k = T::key(el);
    const bool in = lane > l && lane < h;
    const bool isA = in && k >= pk, isB = in && k <= pk;
    const uint64_t mA = __ballot(isA), mB = __ballot(isB);
    const int rA = int(__builtin_amdgcn_mbcnt_hi(uint32_t(mA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mA), 0u)));
    const int Bge = int(__builtin_popcountll(mB)) -
                    int(__builtin_amdgcn_mbcnt_hi(uint32_t(mB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mB), 0u)));
    const bool swapA = isA && (Bge - (isB ? 1 : 0)) > rA;         // B's strictly right of me > A's left of me
    const bool swapB = isB && rA >= Bge;                          // A's strictly left of me > B's strictly right (= Bge - 1)
    const int slot = swapA ? rA : (swapB ? 64 - Bge : 31);        // A number r -> slot r, B number s = Bge - 1 -> slot 63 - s
    const W inbox = xbar_push(slot, el);
    const int from = swapA ? 63 - rA : Bge - 1;
    const W got = xbar_pull(from, inbox);
    el = (swapA || swapB) ? got : el;
    const uint64_t nsA = __ballot(isA && !swapA), sB = __ballot(swapB);
    const int am = nsA ? int(__builtin_ctzll(nsA)) : h;
    const int bl = sB ? int(__builtin_ctzll(sB)) : h;
    const int cut = am < bl ? am : bl;
    if (cut <= nr) l = cut; else h = cut;
    if (guard == 255 && lane == 0) guard_hit(1, S.status);
  }
  if (!fallback) {                                               // __insertion_sort on the <= 3 elements left
    const int m = h - l;
    if (m >= 2) {
      W w0 = rdlane(el, l), w1 = rdlane(el, l + 1);
      if (w_less(w1, w0)) { const W t = w0; w0 = w1; w1 = t; }
      if (m == 3) {
        const W v = rdlane(el, l + 2);
        if (w_less(v, w0)) { el = wrlane(el, v, l, lane); el = wrlane(el, w0, l + 1, lane); el = wrlane(el, w1, l + 2, lane); }
        else if (w_less(v, w1)) { el = wrlane(el, w0, l, lane); el = wrlane(el, v, l + 1, lane); el = wrlane(el, w1, l + 2, lane); }
        else { el = wrlane(el, w0, l, lane); el = wrlane(el, w1, l + 1, lane); }
      } else {
        el = wrlane(el, w0, l, lane); el = wrlane(el, w1, l + 1, lane);
      }
    }
  }
  if (lane < n) S.w[lo + lane] = el;
  wave_lds_order();
  if (fallback) {                                                // depth limit: __heap_select(first, nth + 1, last); iter_swap(first, nth)
    if (lane == 0) {
      s2_heap_select(S.w, lo + l, lo + nr + 1, lo + h);
      const W t = S.w[lo + l]; S.w[lo + l] = S.w[lo + nr]; S.w[lo + nr] = t;
    }
    wave_lds_order();
  }
}

// ---- third generation (round 6): the range lives in REGISTERS for ALL rounds ---------------------------------------
// Measured (profiles/r06_a_chan_select_rounds.csv): a cooperative LDS round of the 16-wave channel selection costs 2.0 us
// whatever its range (3584 or 1207 elements: four waves per SIMD, ~350 instructions each, issue-bound), a one-wave LDS
// round 0.8-1.6 us, a register round (introselect_tail64) 0.36 us.  Here every round is a register round:
//   * position p of the array is owned by (row r = (p - base) / 64, lane (p - base) % 64); row r belongs to wave r % NW and
//     is that wave's register slot r / NW (E slots per lane: NW * E rows <= 64).  The words never move between rounds --
//     a swap EXCHANGES two registers through a mailbox in rank space;
//   * a v_cmp IS a ballot: the two stop sets of a row are two SGPR pairs; ranks are v_mbcnt on top of per-row prefix
//     counts, which cross the waves as one packed word per row (one LDS store per wave, one barrier, one DPP scan);
//   * the i-th stop from the left that is "not less than the pivot" (A) is swapped with the i-th from the right that is
//     "not greater" (B) while it stands left of it -- decided per lane from the ranks alone (introselect_tail64's rule);
//     swapped A number r leaves its word in mbA[r] and takes mbB[r], B number s (from the right) leaves mbB[s] and takes mbA[s];
//   * S.w stays a SHADOW of the registers (one store per swapped element): the next round's pivot candidates, the serial
//     fallbacks, the tail and the caller's epilogue read it.
// Three workgroup barriers per multi-wave round (counts / mailbox / shadow + cut), none in the one-wave instantiation.
constexpr int kX3Cnt = 32;      // xch words [32, 96): per-row stop counts, #A | #B << 16
constexpr int kX3Cut = 96;      // [96, 112): per-wave cut candidates
constexpr int kX3Dum = 112;     // predicated stores of lanes that have nothing to store

__device__ __forceinline__ int mbcnt64(uint64_t m) {
  return int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)));
}

// Partition rounds of std::__introselect on S.w[lo, hi) while hi - lo > stop_len and depth > 0.  el[j] = the word at
// position base + (j * NW + wave) * 64 + lane (== S.w there; positions outside the array: anything).  All 64 * NW threads
// call with identical (base, lo, hi, nth, depth); lo / hi / depth come back updated (identical in every thread).
// mb: the mailbox, mbtop + 1 words (mbtop even, >= 2 (n / 2) + 16).
// Slots are processed in GROUPS of four (1024 consecutive positions at NW = 4): a group the range does not meet is skipped
// by one wave-uniform branch, inside a group the code is branch-free.  The two stop sets of a row are computed twice (before
// and after the count barrier) rather than kept: 4 E SGPRs across a barrier mean spills to VGPR lanes.
constexpr int kSel3Group = 4;
__device__ __forceinline__ int mbcnt64_from(uint64_t m, int base) {      // base + lanes of m below me
  return int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), uint32_t(base))));
}
template <int NW, int E>
__device__ __forceinline__ void sel3_rounds(const Sel2<uint32_t>& S, uint32_t (&el)[E], int base, int& lo, int& hi, int nth,
                                            int& depth, int stop_len, uint32_t* mb, int mbtop, int tid) {
  using T = WordTr<uint32_t>;
  constexpr int G = kSel3Group, NG = E / G;
  static_assert(NW * E <= 64, "one lane per row in the prefix scan");
  static_assert(E % G == 0, "whole groups");
  const int lane = tid & 63;
  const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int p0 = base + wave * 64 + lane;                            // my position in slot 0; slot j: + j * NW * 64
  // mailbox: swapped A number r leaves its word at index r, swapped B number s (from the right) at mbtop - s; each takes the
  // other's, i.e. index mbtop - (its own).  dumi and mbtop - dumi lie in the gap between the two halves.
  const int dumi = mbtop / 2 - 2;
  for (int guard = 0; hi - lo > stop_len && hi - lo > 3 && depth > 0 && guard < 256; ++guard) {
    --depth;
    if (tid == 0) VC2_ROUND(S, NW > 1 ? 310 : 330, hi - lo);
    const int first = lo + 1;
    const uint32_t len = uint32_t(hi - first);
    const int pb = lo + (hi - lo) / 2, pc = hi - 1;
    const uint32_t wlo = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[lo])));
    const uint32_t wa = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[first])));
    const uint32_t wb = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[pb])));
    const uint32_t wc = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[pc])));
    int msrc;
    uint32_t wp;
    {
      const uint32_t ka = T::key(wa), kb = T::key(wb), kc = T::key(wc);
      const bool ab = ka < kb, bc = kb < kc, ac = ka < kc;          // __move_median_to_first
      const int sel = ab ? (bc ? 1 : (ac ? 2 : 0)) : (ac ? 0 : (bc ? 2 : 1));
      msrc = sel == 0 ? first : (sel == 1 ? pb : pc);
      wp = sel == 0 ? wa : (sel == 1 ? wb : wc);
    }
    const uint32_t pk = T::key(wp);
    // which groups of mine meet [lo, hi)  (bit g; wave-uniform)
    uint32_t gact = 0u;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int gs = base + g * G * NW * 64, ge = gs + G * NW * 64;
      gact |= (ge > lo && gs < hi) ? (1u << g) : 0u;
    }
    uint32_t cntrow = 0u;                                          // lane j: the packed counts of my slot j
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (NG > 1 && !((gact >> g) & 1u)) continue;                  // (no per-lane state is written in a skipped group)
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = g * G + jj;
        const int p = p0 + j * NW * 64;
        el[j] = (p == msrc) ? wlo : el[j];                          // iter_swap(lo, median), register copy
        el[j] = (p == lo) ? wp : el[j];
        const uint32_t k = T::key(el[j]);
        const uint64_t m_in = __builtin_amdgcn_ballot_w64(uint32_t(p - first) < len);
        const uint64_t mA = __builtin_amdgcn_ballot_w64(k >= pk) & m_in, mB = __builtin_amdgcn_ballot_w64(k <= pk) & m_in;
        const uint32_t c = uint32_t(__builtin_popcountll(mA)) | (uint32_t(__builtin_popcountll(mB)) << 16);
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cntrow) : "s"(c), "n"(j));      // (c is wave-uniform: two s_bcnt1)
      }
    }
    if (lane < E) S.xch[kX3Cnt + lane * NW + wave] = cntrow;        // (rows of skipped groups: zero)
    sel2_sync<NW>();                                                // ---- 1: counts; every thread has read the candidates
    if (tid == 0) { S.w[lo] = wp; S.w[msrc] = wlo; }                // iter_swap(lo, median), shadow copy
    const uint32_t crow = lane < NW * E ? S.xch[kX3Cnt + lane] : 0u;
    const uint32_t incl = wave_incl_scan_u32(crow);
    const uint32_t excl = incl - crow;
    const int totB = int(uint32_t(__builtin_amdgcn_readlane(int(incl), 63)) >> 16);
    uint32_t cand = 0xFFFFFFFFu;
    int di[E];                                                      // where I left my word (dumi: nowhere)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (NG > 1 && !((gact >> g) & 1u)) continue;
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = g * G + jj;
        const int p = p0 + j * NW * 64;
        uint32_t w = el[j];
        asm volatile("" : "+v"(w));                                 // (recompute the stop sets: see above)
        const uint32_t k = T::key(w);
        const bool in = uint32_t(p - first) < len, ge = k >= pk, le = k <= pk;
        const uint64_t m_in = __builtin_amdgcn_ballot_w64(in);
        const uint64_t mA = __builtin_amdgcn_ballot_w64(ge) & m_in, mB = __builtin_amdgcn_ballot_w64(le) & m_in;
        const bool a = in && ge, b = in && le;
        const uint32_t ex = uint32_t(__builtin_amdgcn_readlane(int(excl), j * NW + wave));
        const int rA = mbcnt64_from(mA, int(ex & 0xFFFFu));                         // A's left of me
        const int Bge = totB - mbcnt64_from(mB, int(ex >> 16));                     // B's at or right of me
        const bool swapA = a && (Bge - (b ? 1 : 0)) > rA;                           // B's strictly right of me > A's left of me
        const bool swapB = b && rA >= Bge;                                          // A's strictly left of me > B's strictly right of me
        di[j] = swapA ? rA : (swapB ? mbtop + 1 - Bge : dumi);                      // (B number s = Bge - 1 -> mbtop - s)
        mb[di[j]] = el[j];
        const uint32_t cc = ((a && !swapA) || swapB) ? uint32_t(p) : 0xFFFFFFFFu;   // where `first` halts: the first A that
        cand = cc < cand ? cc : cand;                                              //   stays, or the leftmost swapped B
      }
    }
    const uint32_t wcut = wave_min_bcast_u32(cand);
    if constexpr (NW > 1) { if (lane == 0) S.xch[kX3Cut + wave] = wcut; }
    sel2_sync<NW>();                                                // ---- 2: mailbox, cut candidates
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (NG > 1 && !((gact >> g) & 1u)) continue;
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        const int j = g * G + jj;
        const uint32_t got = mb[mbtop - di[j]];
        const bool sw = di[j] != dumi;
        el[j] = sw ? got : el[j];
        S.w[sw ? p0 + j * NW * 64 : S.dumw] = el[j];
      }
    }
    uint32_t cutv = wcut;
    if constexpr (NW > 1) {
#pragma unroll
      for (int v = 0; v < NW; ++v) { const uint32_t t = S.xch[kX3Cut + v]; cutv = t < cutv ? t : cutv; }
      cutv = uint32_t(__builtin_amdgcn_readfirstlane(int(cutv)));
    }
    sel2_sync<NW>();                                                // ---- 3: the shadow is current, the cells are free
    const int cut = cutv < uint32_t(hi) ? int(cutv) : hi;
    if (cut <= nth) lo = cut; else hi = cut;
    if (guard == 255 && tid == 0) guard_hit(6, S.status);
  }
}

// the rest of std::__introselect on S.w[lo, hi) by ONE wave (tid < 64): register rounds over up to 64 * E2 elements, the
// last <= 64 by introselect_tail64, serial libstdc++ pieces at the depth limit / for <= 3 elements.  (A range longer than
// 64 * E2 -- only possible with other callers' arguments -- falls back to the LDS rounds.)
template <int E2>
__device__ __forceinline__ void introselect3_finish(const Sel2<uint32_t>& S, int lo, int hi, int nth, int depth, int nmax,
                                                    uint32_t* mb, int mbtop, int lane) {
  if constexpr (E2 >= kSel3Group) if (hi - lo > kSel2TailMax && hi - lo <= 64 * E2 && depth > 0) {
    uint32_t e2[E2];
    const int base = lo;
#pragma unroll
    for (int j = 0; j < E2; ++j) { const int p = base + 64 * j + lane; e2[j] = S.w[p < nmax ? p : nmax - 1]; }
    sel3_rounds<1, E2>(S, e2, base, lo, hi, nth, depth, kSel2TailMax, mb, mbtop, lane);
  }
  bool done = false;
  for (int guard = 0; hi - lo > 3 && guard < 256; ++guard) {
    if (hi - lo <= kSel2TailMax) { introselect_tail64<uint32_t>(S, lo, hi, nth, depth, lane); done = true; break; }
    if (depth == 0) {
      if (lane == 0) { s2_heap_select(S.w, lo, nth + 1, hi); const uint32_t t = S.w[lo]; S.w[lo] = S.w[nth]; S.w[nth] = t; }
      done = true;
      break;
    }
    --depth;
    if (lane == 0) VC2_ROUND(S, 230, hi - lo);
    const int cut = sel2_partition<uint32_t, 1, 0, 4>(S, lo, hi, S.la, S.lb, lane);
    if (cut <= nth) lo = cut; else hi = cut;
    if (guard == 255 && lane == 0) guard_hit(1, S.status);
  }
  if (!done && lane == 0) { VC2_ROUND(S, 240, hi - lo); s2_insertion_sort(S.w, lo, hi); }
}

// std::nth_element(first, first + nth, first + n) on S.w[0, n), n <= 64 * NW * E.  el[j]: the word at position
// (j * NW + wave) * 64 + lane, already stored to S.w as well (a barrier behind the stores).  All 64 * NW threads call.
template <int NW, int E, int E2>
__device__ __forceinline__ void introselect3(const Sel2<uint32_t>& S, uint32_t (&el)[E], int n, int nth, int tid) {
  if (n == 0 || nth >= n) return;
  int lo = 0, hi = n;
  int depth = 2 * (31 - __clz(n));
  uint32_t* const mb = reinterpret_cast<uint32_t*>(S.la);           // la | lb: n + kSel2Pad words: A words [0, n / 2), B words (mbtop - n / 2, mbtop]
  const int mbtop = (n / 2) * 2 + 16;
  if constexpr (NW > 1) sel3_rounds<NW, E>(S, el, 0, lo, hi, nth, depth, 64 * E2, mb, mbtop, tid);
  if (tid < 64) introselect3_finish<E2>(S, lo, hi, nth, depth, n + kSel2Pad - 1, mb, mbtop, tid);
  sel2_sync<NW>();
  if (tid == 0) VC2_ROUND(S, 290, 0);
}

// one-wave selections (k_select: N tokens of a frame): S.w[0, n) holds the words, n <= 64 * E2; lanes of ONE wave call
template <int E2>
__device__ __forceinline__ void topk_smallest3_solo(const Sel2<uint32_t>& S, int n, int k, int lane) {
  if (k <= 0 || k >= n) return;
  if (int64_t(k) * 64 <= int64_t(n)) {
    if (lane == 0) s2_heap_select(S.w, 0, k, n);
    wave_lds_order();
    return;
  }
  introselect3_finish<E2>(S, 0, n, k - 1, 2 * (31 - __clz(n)), n + kSel2Pad - 1, reinterpret_cast<uint32_t*>(S.la), (n / 2) * 2 + 16, lane);
  wave_lds_order();
  if (lane == 0) VC2_ROUND(S, 290, 0);
}

// ---- fourth form (round 6): registers + thread-contiguous positions + VALU ranks ------------------------------------------
// What the per-round stamps say (profiles/r06_a_chan_select_rounds.csv, r06_b_sel3_register_rounds.csv): a round of the
// 16-wave LDS form is ISSUE-bound -- 4 waves per SIMD x ~330 instructions -- and costs 2.0 us whatever its range; a round
// built on ballots / SGPR masks is LATENCY-bound on one wave per SIMD.  So: keep 16 waves, cut the instructions.
//   * thread t owns positions 4t .. 4t + 3 for the WHOLE selection, in registers (no re-blocking, no re-read per round);
//     S.w stays a shadow (one 16-byte store per thread and round);
//   * ranks are running counts inside the thread on top of ONE DPP scan of the packed per-thread counts and a second,
//     lane-parallel scan of the 16 wave totals (the LDS form adds them up in a 16-step loop in every thread);
//   * the swap partners meet in rank space (sel3_rounds' mailbox: A number r at r, B number s at mbtop - s), so there are no
//     rank -> position tables: one LDS write and one read per element where the LDS form has two and three;
//   * a wave whose 256 positions do not meet [lo, hi) only keeps the barriers company (one uniform branch).
// Rounds run while the range is longer than 64; the rest is introselect_tail64 / the serial libstdc++ pieces (wave 0).
constexpr int kX4State = 28;    // xch words [28, 31): lo, hi, depth after the round (for the waves that sleep through it)
// ONE partition round.  SOLO = false: all awake waves of the workgroup together (three workgroup barriers); SOLO = true: the
// range lies inside THIS wave's 256 positions -- no other wave is involved, no barrier (a wave's LDS accesses execute in order).
// (sel4_partition: the partition itself -- returns the cut, the same value in every thread; the shadow S.w is written but NOT
//  yet ordered against other waves: the caller's next barrier / wave_lds_order does that)
template <int NW, bool SOLO, int E = 4>
__device__ __forceinline__ int sel4_partition(const Sel2<uint32_t>& S, uint32_t (&el)[E], int n, int lo, int hi,
                                              uint32_t* mb, int mbtop, int p0, int lane, int wave, int off) {
  using T = WordTr<uint32_t>;
  const int wspan_lo = wave * 64 * E - off, wspan_hi = wspan_lo + 64 * E;
  const int dumi = mbtop / 2 - 2;
  const int first = lo + 1;
  const uint32_t len = uint32_t(hi - first);
  const int pb = lo + (hi - lo) / 2, pc = hi - 1;
  const uint32_t wlo = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[lo])));
  const uint32_t wa = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[first])));
  const uint32_t wb = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[pb])));
  const uint32_t wc = uint32_t(__builtin_amdgcn_readfirstlane(int(S.w[pc])));
  int msrc;
  uint32_t wp;
  {
    const uint32_t ka = T::key(wa), kb = T::key(wb), kc = T::key(wc);
    const bool ab = ka < kb, bc = kb < kc, ac = ka < kc;            // __move_median_to_first
    const int sel = ab ? (bc ? 1 : (ac ? 2 : 0)) : (ac ? 0 : (bc ? 2 : 1));
    msrc = sel == 0 ? first : (sel == 1 ? pb : pc);
    wp = sel == 0 ? wa : (sel == 1 ? wb : wc);
  }
  const uint32_t pk = T::key(wp);
  bool a[E], b[E];
  uint32_t packed = 0u;
  if (SOLO || (msrc >= wspan_lo && msrc < wspan_hi) || (lo >= wspan_lo && lo < wspan_hi)) {   // iter_swap(lo, median), register copy
#pragma unroll
    for (int e = 0; e < E; ++e) { el[e] = (p0 + e == msrc) ? wlo : el[e]; el[e] = (p0 + e == lo) ? wp : el[e]; }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t k = T::key(el[e]);
    const bool in = uint32_t(p0 + e - first) < len;
    a[e] = in && k >= pk;
    b[e] = in && k <= pk;
    packed += (a[e] ? 1u : 0u) + (b[e] ? 0x10000u : 0u);
  }
  const uint32_t incl = wave_incl_scan_u32(packed);
  uint32_t bas = incl - packed;
  int totB;
  if constexpr (SOLO) {
    wave_lds_order();                                               // (the candidates are read)
    totB = int(uint32_t(__builtin_amdgcn_readlane(int(incl), 63)) >> 16);
  } else {
    if (lane == 63) S.xch[wave] = incl;
    __syncthreads();                                                // ---- 1: wave totals; every thread has read the candidates
    const uint32_t wt = lane < NW ? S.xch[lane] : 0u;
    const uint32_t wincl = wave_incl_scan_u32(wt);
    bas += uint32_t(__builtin_amdgcn_readlane(int(wincl - wt), wave));
    totB = int(uint32_t(__builtin_amdgcn_readlane(int(wincl), 63)) >> 16);
  }
  if (uint32_t(lo - p0) < uint32_t(E)) { S.w[lo] = wp; S.w[msrc] = wlo; }    // iter_swap(lo, median), shadow copy: by the owner of `lo`
  int rA = int(bas & 0xFFFFu), nBl = int(bas >> 16);                 // A's left of my first element; B's left of it
  uint32_t cand = 0xFFFFFFFFu;
  int di[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int Bge = totB - nBl;                                      // B's at or right of this element
    const bool swapA = a[e] && (Bge - (b[e] ? 1 : 0)) > rA;
    const bool swapB = b[e] && rA >= Bge;
    di[e] = swapA ? rA : (swapB ? mbtop + 1 - Bge : dumi);
    mb[di[e]] = el[e];
    const uint32_t cc = ((a[e] && !swapA) || swapB) ? uint32_t(p0 + e) : 0xFFFFFFFFu;
    cand = cc < cand ? cc : cand;
    rA += a[e] ? 1 : 0; nBl += b[e] ? 1 : 0;
  }
  uint32_t cutv = wave_min_bcast_u32(cand);
  if constexpr (SOLO) {
    wave_lds_order();
  } else {
    if (lane == 0) S.xch[kX3Cut + wave] = cutv;
    __syncthreads();                                                // ---- 2: mailbox, cut candidates
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t got = mb[mbtop - di[e]];
    el[e] = di[e] != dumi ? got : el[e];
  }
  if (uint32_t(p0) < uint32_t(n)) {                                 // (the pad takes positions >= n)
#pragma unroll
    for (int q = 0; q < E / 4; ++q)
      reinterpret_cast<uint4*>(S.w + p0)[q] = make_uint4(el[4 * q], el[4 * q + 1], el[4 * q + 2], el[4 * q + 3]);
  }
  if constexpr (!SOLO) {
    const uint32_t ct = lane < NW ? S.xch[kX3Cut + lane] : 0xFFFFFFFFu;
    cutv = wave_min_bcast_u32(ct);
  }
  return cutv < uint32_t(hi) ? int(cutv) : hi;
}
// one introselect round: the partition, the side that holds nth, and (not SOLO) the new state for the waves that sleep
template <int NW, bool SOLO, int E = 4>
__device__ __forceinline__ void sel4_round(const Sel2<uint32_t>& S, uint32_t (&el)[E], int n, int& lo, int& hi, int nth,
                                           int depth, uint32_t* mb, int mbtop, int p0, int lane, int wave, int off) {
  const int cut = sel4_partition<NW, SOLO, E>(S, el, n, lo, hi, mb, mbtop, p0, lane, wave, off);
  if (cut <= nth) lo = cut; else hi = cut;
  if constexpr (SOLO) {
    wave_lds_order();
  } else {
    if (lane == 0) { S.xch[kX4State] = uint32_t(lo); S.xch[kX4State + 1] = uint32_t(hi); S.xch[kX4State + 2] = uint32_t(depth); }   // (every
    __syncthreads();                                                // ---- 3: the shadow is current, the cells are free   awake wave: the same values)
  }
}

// std::nth_element(first, first + nth, first + n) on S.w[0, n), n + off <= 256 NW.  el[e]: the word at position 4 tid - off + e,
// already stored to S.w as well (a barrier behind the stores; words at positions outside [0, n): anything).  All 64 * NW
// threads call.  off (a multiple of 4, sel4_offset): the positions are shifted against the threads so that `nth` -- which every
// range contains -- sits in the MIDDLE of a wave's 256 positions: with k = D / 2 = 7 x 256 it stood on a wave boundary and
// no range ever fitted one wave.
// Phase 1: rounds by all the waves whose positions meet the range (the others sleep through the barriers) while the range is
// longer than 64 and does not fit 256 positions from a 16-byte boundary.  Phase 2: wave 0 re-blocks the range from the shadow
// into its registers and goes on alone -- barrier-free rounds, then introselect_tail64 / the serial libstdc++ pieces -- while
// the others wait at the final barrier.
__host__ __device__ inline int sel4_offset(int n, int nth, int nw) {
  const int off = ((128 - (nth & 255) + 256) & 255) & ~3;
  return n + off <= 256 * nw ? off : 0;
}
template <int NW>
__device__ __forceinline__ void introselect4(const Sel2<uint32_t>& S, uint32_t (&el)[4], int n, int nth, int tid, int off) {
  if (n == 0 || nth >= n) return;
  static_assert(NW <= 16, "wave totals: one lane each, cells [0, 16) of xch");
  int lo = 0, hi = n;
  int depth = 2 * (31 - __clz(n));
  uint32_t* const mb = reinterpret_cast<uint32_t*>(S.la);           // la | lb: n + kSel2Pad words (see introselect3)
  const int mbtop = (n / 2) * 2 + 16;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = 4 * tid - off;
  const int wspan_lo = wave * 256 - off, wspan_hi = wspan_lo + 256;
  bool asleep = false;                                              // (wave-uniform) the range has left my wave's positions for good
  auto more = [&]() { return hi - lo > kSel2TailMax && depth > 0; };
  // the range fits ONE wave's registers once it is re-blocked: 256 positions from the 16-byte boundary at or below lo.  (Until
  // the last hours of round 6 the hand-over waited for the range to lie inside one wave's OWN 256 positions: the rounds at
  // 252 and 141 elements of the target input straddled two waves and cost 1.2 / 1.5 us instead of 0.76.)
  auto fits = [&]() { return hi - (lo & ~3) <= 256; };
  for (int guard = 0; more() && !fits() && guard < 256; ++guard) {
    if (tid == 0) VC2_ROUND(S, 410, hi - lo);
    if (!asleep && !(wspan_hi > lo && wspan_lo < hi)) {             // the range only shrinks: from now on this wave keeps the
      asleep = true;                                                //   barriers company and reads what the others decide
      if (lane == 63) S.xch[wave] = 0u;                             // (my cells are written by nobody else: once is enough)
      if (lane == 0) S.xch[kX3Cut + wave] = 0xFFFFFFFFu;
    }
    if (asleep) {
      __syncthreads(); __syncthreads(); __syncthreads();
      lo = __builtin_amdgcn_readfirstlane(int(S.xch[kX4State])); hi = __builtin_amdgcn_readfirstlane(int(S.xch[kX4State + 1]));
      depth = __builtin_amdgcn_readfirstlane(int(S.xch[kX4State + 2]));
      continue;
    }
    --depth;
    sel4_round<NW, false>(S, el, n, lo, hi, nth, depth, mb, mbtop, p0, lane, wave, off);
    if (guard == 255 && tid == 0) guard_hit(6, S.status);
  }
  // (every wave arrives here with the same lo / hi / depth; the shadow S.w is current: the last round -- or the caller's pack --
  //  ended with a barrier.)  Wave 0 goes on alone: it takes the 256 positions from `base` out of the shadow -- lane l: base + 4 l
  //  ... + 3 -- and runs barrier-free rounds, then the register tail; the others wait at the final barrier.
  if (wave == 0) {
    if (more()) {
      const int base = lo & ~3, q0 = base + 4 * lane;
      const uint4 v = *reinterpret_cast<const uint4*>(S.w + (q0 < n ? q0 : 0));     // (the pad takes the last lanes' excess)
      uint32_t r4[4] = {v.x, v.y, v.z, v.w};
      for (int guard = 0; more() && guard < 256; ++guard) {
        if (lane == 0) VC2_ROUND(S, 430, hi - lo);
        --depth;
        sel4_round<NW, true>(S, r4, n, lo, hi, nth, depth, mb, mbtop, q0, lane, wave, off);
        if (guard == 255 && lane == 0) guard_hit(6, S.status);
      }
    }
    introselect3_finish<1>(S, lo, hi, nth, depth, n + kSel2Pad - 1, mb, mbtop, lane);
  }
  __syncthreads();
  if (tid == 0) VC2_ROUND(S, 290, 0);
}

// one-wave selections (k_select: the N tokens of a frame, N <= 64 E): S.w[0, n) holds the words; lane l takes positions
// E l .. E l + E - 1 into registers and runs barrier-free sel4 rounds down to 64 elements, then the register tail.  (The pad
// behind S.w must hold E - 1 words more than the array: kSel2Pad = 64 does.)
template <int E>
__device__ __forceinline__ void topk_smallest4_solo(const Sel2<uint32_t>& S, int n, int k, int lane) {
  static_assert(E == 4 || E == 8, "whole 16-byte vectors");
  if (k <= 0 || k >= n) return;
  if (int64_t(k) * 64 <= int64_t(n)) {
    if (lane == 0) s2_heap_select(S.w, 0, k, n);
    wave_lds_order();
    return;
  }
  uint32_t el[E];
  const int p0 = E * lane;
#pragma unroll
  for (int q = 0; q < E / 4; ++q) {
    const uint4 v = reinterpret_cast<const uint4*>(S.w + (p0 < n ? p0 : 0))[q];
    el[4 * q] = v.x; el[4 * q + 1] = v.y; el[4 * q + 2] = v.z; el[4 * q + 3] = v.w;
  }
  int lo = 0, hi = n, depth = 2 * (31 - __clz(n));
  const int nth = k - 1;
  uint32_t* const mb = reinterpret_cast<uint32_t*>(S.la);
  const int mbtop = (n / 2) * 2 + 16;
  for (int guard = 0; hi - lo > kSel2TailMax && depth > 0 && guard < 256; ++guard) {
    if (lane == 0) VC2_ROUND(S, 430, hi - lo);
    --depth;
    sel4_round<1, true, E>(S, el, n, lo, hi, nth, depth, mb, mbtop, p0, lane, 0, 0);
    if (guard == 255 && lane == 0) guard_hit(6, S.status);
  }
  introselect3_finish<1>(S, lo, hi, nth, depth, n + kSel2Pad - 1, mb, mbtop, lane);
  wave_lds_order();
  if (lane == 0) VC2_ROUND(S, 290, 0);
}

// std::nth_element(first, first + nth, first + n) on S.w[0, n).  All 64*NW threads of the workgroup call this
// (tid = threadIdx.x).  Ranges longer than sel2_capacity(1, SOLO) are partitioned by all NW waves together
// (n <= sel2_capacity(NW, COOP)), shorter ones by wave 0 alone.
// NWA (<= NW): the waves that take part in a cooperative round; the others only keep the workgroup barriers company.
// A partition round is a serial chain of dependent steps: more than one wave per SIMD on it only contend for issue
// slots (measured: 16-wave rounds ~4 us, 4-wave rounds ~2.2 us at 3584 elements).
template <typename W, int NW, int SOLO, int COOP, int NWA = NW>
__device__ __forceinline__ void introselect2(const Sel2<W>& S, int n, int nth, int tid) {
  if (n == 0 || nth >= n) return;
  int lo = 0, hi = n;
  int depth = 2 * (31 - __clz(n));                            // std::__lg(n) * 2
  bool done = false;
  if constexpr (NW > 1) {
    for (int guard = 0; hi - lo > sel2_capacity(1, SOLO) && guard < 256; ++guard) {
      if (depth == 0) {
        if (tid == 0) { s2_heap_select(S.w, lo, nth + 1, hi); const W t = S.w[lo]; S.w[lo] = S.w[nth]; S.w[nth] = t; }
        done = true;
        break;
      }
      --depth;
      if (tid == 0) VC2_SEL_STAMP(210);                           // a cooperative round begins
      if (tid == 0) VC2_ROUND(S, 210, hi - lo);
      int cut;
      if constexpr (NWA == NW) {
        cut = sel2_partition<W, NW, 1, COOP>(S, lo, hi, S.la, S.lb, tid);
      } else {
        if (tid < 64 * NWA) {
          cut = sel2_partition<W, NWA, 1, COOP>(S, lo, hi, S.la, S.lb, tid);
        } else {                                              // the partition's three workgroup barriers, then its cut
          __syncthreads(); __syncthreads(); __syncthreads();
          const uint32_t cutv = S.xch[kXchCut];
          cut = cutv < uint32_t(hi) ? int(cutv) : hi;
        }
        // (the cut cell is reset by thread 0 after the NEXT round's first barrier: every reader is done by then)
      }
      if (cut <= nth) lo = cut; else hi = cut;
      if (guard == 255 && tid == 0) guard_hit(0, S.status);
    }
  }
  if (!done && tid < 64) {
    for (int guard = 0; hi - lo > 3 && guard < 256; ++guard) {
#ifndef VC2_NO_REG_TAIL
      // (one-wave selections only: in the 16-wave channel selection the same tail measured 0.7 us SLOWER than the LDS
      //  rounds it replaces -- 21.3 against 20.6 us over 1500 launches -- while k_select gained 2.2 us)
      if (NW == 1 && hi - lo <= kSel2TailMax) {                    // the rest in registers (incl. the final insertion sort)
        if (tid == 0) VC2_SEL_STAMP(250);
        introselect_tail64<W>(S, lo, hi, nth, depth, tid);
        done = true;
        break;
      }
#endif
      if (depth == 0) {
        if (tid == 0) { s2_heap_select(S.w, lo, nth + 1, hi); const W t = S.w[lo]; S.w[lo] = S.w[nth]; S.w[nth] = t; }
        done = true;
        break;
      }
      --depth;
      if (tid == 0) VC2_SEL_STAMP(230);                           // a one-wave LDS round begins
      if (tid == 0) VC2_ROUND(S, 230, hi - lo);
      const int cut = sel2_partition<W, 1, 0, SOLO>(S, lo, hi, S.la, S.lb, tid);
      if (cut <= nth) lo = cut; else hi = cut;
      if (guard == 255 && tid == 0) guard_hit(1, S.status);
    }
    if (!done && tid == 0) VC2_ROUND(S, 240, hi - lo);
    if (!done && tid == 0) s2_insertion_sort(S.w, lo, hi);
  }
  sel2_sync<NW>();
  if (tid == 0) VC2_ROUND(S, 290, 0);
}

// torch.topk(v, k, largest=False) SET: afterwards S.w[0, k) holds the kept elements (ATen/native/TopKImpl.h:
// partial_sort when k*64 <= n, else nth_element(k-1))
template <typename W, int NW, int SOLO, int COOP, int NWA = NW>
__device__ __forceinline__ void topk_smallest2(const Sel2<W>& S, int n, int k, int tid) {
  if (k <= 0 || k >= n) return;                               // k == n: everything kept, nothing moves
  if (int64_t(k) * 64 <= int64_t(n)) {
    if (tid == 0) s2_heap_select(S.w, 0, k, n);               // partial_sort = heap_select + sort_heap (only permutes [0,k))
    sel2_sync<NW>();
  } else {
    introselect2<W, NW, SOLO, COOP, NWA>(S, n, k - 1, tid);
  }
}

// ---- std::sort(first, first + n) replay (the `sorted=True` half of torch.topk, TopKImpl.h) ---------------
// __introsort_loop: every segment longer than 16 is partitioned (the same __unguarded_partition_pivot) and both
// halves recurse with depth_limit - 1; at depth 0 a segment is heap-sorted instead.  Segments are independent, so
// the order in which they are partitioned is free:
//   phase A  all waves together partition the longest pending segment while that pays (see there).  A cooperative
//            partition returns its cut to every thread, so the pending segments are THREAD-UNIFORM bookkeeping: four
//            register slots, no memory, no atomics, no barriers beyond the partition's own three;
//   phase B  the remaining segments are dealt round-robin to the waves; every wave sorts its own (a private stack,
//            one wave per partition, no synchronisation at all).
// __final_insertion_sort then equals a STABLE sort inside every leaf (<= 16 elements; elements never cross a cut
// and the insertion uses a strict compare); because everything left of a leaf is <= and everything right of it is
// >= its elements, the final position of element p is p + #{q in (p, p+16): key_q < key_p} - #{q in (p-16, p):
// key_q > key_p} -- no leaf bookkeeping at all.
// Slices: a partition's two sides are independent, so R workgroups replay the same sort side by side.  The workgroups
// [lo, hi) that stand on a segment all partition it (the same input as everybody else on that segment, so the same
// result) and then split in proportion to the two sides' lengths -- the first round(g * left / total) of them (at least
// one, at most g - 1) step left, the others right -- until a workgroup stands alone: that is the segment it sorts, ranks
// and stores.  The R arrival segments tile [0, n): no position is answered twice, none is left out.  (A segment that is
// already a leaf while several workgroups stand on it goes to the first of them.)
// out(p, i): original index i stands at sorted position p (called for the positions of the arrival segment).
constexpr int kSortOwnCap = 512;         // a wave's own segments (every segment > 16: at most n / 17 = 481 in all)
struct SortScratch2 {
  uint32_t* own;    // [NW][kSortOwnCap]  every wave's private stack
};
__host__ __device__ inline size_t sort2_bytes(int /*n*/, int nw = 4) { return size_t(nw) * kSortOwnCap * 4 + 16; }
__device__ __forceinline__ SortScratch2 sort2_carve(unsigned char* p, int /*nw*/) {
  SortScratch2 Q;
  Q.own = reinterpret_cast<uint32_t*>(p);
  return Q;
}
__device__ __forceinline__ uint32_t seg_pack(int first, int last, int depth) {
  return uint32_t(first) | (uint32_t(last) << 13) | (uint32_t(depth) << 26);   // last <= 8191, depth <= 26
}

template <typename W, int NW, int SOLO, int COOP, typename OUT>
__device__ __forceinline__ void introsort2(const Sel2<W>& S, const SortScratch2& Q, int n, OUT out, int tid,
                                           int part = 0, int nparts = 1) {
  using T = WordTr<W>;
  constexpr int NT = 64 * NW;
  constexpr int kCoopMin = NW > 1 ? 64 : 0x7FFFFFFF;            // longer segments: all waves together
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t* const mine = Q.own + wave * kSortOwnCap;
#ifdef VC2_SEL2_DEBUG
  if (tid == 0) g_sel2_dbg[0] = __builtin_readcyclecounter();
  int dbg_coop = 0, dbg_dealt = 0; unsigned long long dbg_tc = 0, dbg_td = 0;
#endif
  int nmine = 0, ndealt = 0;
  // pending segments for phase A: at most four, thread-uniform, kept in (scalar) registers sorted by length -- an LDS
  // round trip costs a lone wave ~200 cycles, so no list in memory
  uint32_t s0 = 0u, s1 = 0u, s2 = 0u, s3 = 0u;
  auto seglen = [](uint32_t sg) { return int((sg >> 13) & 0x1FFFu) - int(sg & 0x1FFFu); };
  auto push_mine = [&](uint32_t sg) {
    if (nmine < kSortOwnCap) { if (lane == 0) mine[nmine] = sg; ++nmine; } else if (lane == 0) guard_hit(4, S.status);
  };
  auto deal = [&](uint32_t sg) {                                  // round-robin over the waves (uniform decision)
    if ((ndealt % NW) == wave) push_mine(sg);
    ++ndealt;
  };
  auto push_uniform = [&](int f, int l, int depth) {             // identical arguments in every thread of the workgroup
    if (l - f <= 16) return;                                     // a leaf
    uint32_t sg = seg_pack(f, l, depth);
    if (NW > 1 && l - f > kCoopMin && depth != 0 && l - f <= sel2_capacity(NW, COOP)) {
      uint32_t t;
      if (seglen(sg) > seglen(s0)) { t = s0; s0 = sg; sg = t; }
      if (seglen(sg) > seglen(s1)) { t = s1; s1 = sg; sg = t; }
      if (seglen(sg) > seglen(s2)) { t = s2; s2 = sg; sg = t; }
      if (seglen(sg) > seglen(s3)) { t = s3; s3 = sg; sg = t; }
      if (sg != 0u) deal(sg);                                    // the shortest of five goes to a wave right away
    } else {
      deal(sg);
    }
  };
  // Walk down to my arrival segment [ta, tb).  Workgroups [lo, hi) are on the segment [first, last) together; after its
  // partition (repeated by all of them: same input, same cut) they split IN PROPORTION to the two sides -- round 5; with
  // one bit of `part` per level (rounds 3-4) the arrival segments were products of six random split ratios, 3 .. 200
  // positions where the mean is 28, and the slowest rider took twice the median's time.
  int ta = 0, tb = n;
  {
    int first = 0, last = n, depth = n > 1 ? 2 * (31 - __clz(n)) : 0;
    int lo = 0, hi = nparts > 1 ? nparts : 1;
    bool mine_ = true;
    // (round 6, measured and NOT kept: the walk's partitions on registers -- sel4_partition<4, false, 8>, thread t owning positions
    //  8t .. 8t + 7 -- are bit-identical and SLOWER here: cfg2's sweep-2 launch, which the riders set, 22.0 -> 24.5 us.  Eight
    //  elements per thread double the per-element code of a round; the LDS form re-blocks the range over the threads every round.)
    while (hi - lo > 1) {
      if (last - first <= 16 || depth == 0) { mine_ = part == lo; break; }   // a leaf (or a heapsort segment): one owner
      int cut;
      if constexpr (NW > 1) cut = sel2_partition<W, NW, 0, COOP>(S, first, last, S.la + first, S.lb + first, tid);
      else cut = sel2_partition<W, 1, 0, SOLO>(S, first, last, S.la + first, S.lb + first, lane);
      const int g = hi - lo, nl = cut - first, nt = last - first;
      int m = (2 * g * nl + nt) / (2 * nt);                       // round(g * nl / nt), at least one owner per side
      m = m < 1 ? 1 : (m > g - 1 ? g - 1 : m);
      if (part < lo + m) { hi = lo + m; last = cut; } else { lo += m; first = cut; }
      --depth;
    }
    if (mine_) { ta = first; tb = last; if (last - first > 16) push_uniform(first, last, depth); }
    else { ta = 0; tb = 0; }
  }
#ifdef VC2_SEL2_DEBUG
  if (tid == 0) g_sel2_dbg[1] = __builtin_readcyclecounter();
#endif
  if constexpr (NW > 1) {
    // Phase A.  All waves together partition the LONGEST pending segment while that pays: as long as it is longer
    // than 128, or there are fewer pending segments than waves (a cooperative round costs about as much as a short
    // solo one, but leaves the other waves idle only when there is nothing else to hand them).
    for (int guard = 0; s0 != 0u && guard < 4 * 8192; ++guard) {
      const int cnt = 1 + (s1 != 0u ? 1 : 0) + (s2 != 0u ? 1 : 0) + (s3 != 0u ? 1 : 0);
      const int blen = seglen(s0);
      if (!(blen > 128 || (cnt < NW && blen > 64))) break;
      const uint32_t bsg = s0;
      s0 = s1; s1 = s2; s2 = s3; s3 = 0u;
      const int first = int(bsg & 0x1FFFu), last = int((bsg >> 13) & 0x1FFFu), depth = int(bsg >> 26);
#ifdef VC2_SEL2_DEBUG
      const unsigned long long c0 = __builtin_readcyclecounter();
#endif
      const int cut = sel2_partition<W, NW, 0, COOP>(S, first, last, S.la + first, S.lb + first, tid);
#ifdef VC2_SEL2_DEBUG
      dbg_tc += __builtin_readcyclecounter() - c0; ++dbg_coop;
#endif
      push_uniform(first, cut, depth - 1);
      push_uniform(cut, last, depth - 1);
      if (guard == 4 * 8192 - 1 && tid == 0) guard_hit(2, S.status);
    }
    if (s0 != 0u) deal(s0);
    if (s1 != 0u) deal(s1);
    if (s2 != 0u) deal(s2);
    if (s3 != 0u) deal(s3);
  }
#ifdef VC2_SEL2_DEBUG
  if (tid == 0) g_sel2_dbg[2] = __builtin_readcyclecounter();
#endif
  // Phase B: my own segments, depth first; the segment to continue with stays in a register
  uint32_t cur = 0u;
  for (int guard = 0; guard < 4 * 8192; ++guard) {
    if (cur == 0u) {
      if (nmine == 0) break;
      --nmine;
      wave_lds_order();
      cur = uint32_t(__builtin_amdgcn_readfirstlane(int(mine[nmine])));
    }
    const int first = int(cur & 0x1FFFu), last = int((cur >> 13) & 0x1FFFu), depth = int(cur >> 26);
    cur = 0u;
    if (depth == 0 || last - first > sel2_capacity(1, SOLO)) {  // __partial_sort(first, last, last): heapsort
      if (lane == 0) { s2_heap_select(S.w, first, last, last); s2_sort_heap(S.w, first, last); }
      if (depth != 0 && lane == 0) guard_hit(5, S.status);               // (a long segment that phase A could not take)
      wave_lds_order();
      continue;
    }
#ifdef VC2_SEL2_DEBUG
    const unsigned long long d0 = __builtin_readcyclecounter();
#endif
    const int cut = sel2_partition<W, 1, 0, SOLO>(S, first, last, S.la + first, S.lb + first, lane);
#ifdef VC2_SEL2_DEBUG
    dbg_td += __builtin_readcyclecounter() - d0; ++dbg_dealt;
#endif
    const bool left = cut - first > 16, right = last - cut > 16;
    if (left) {
      cur = seg_pack(first, cut, depth - 1);
      if (right) push_mine(seg_pack(cut, last, depth - 1));
    } else if (right) {
      cur = seg_pack(cut, last, depth - 1);
    }
    if (guard == 4 * 8192 - 1 && lane == 0) guard_hit(3, S.status);
  }
  __syncthreads();
#ifdef VC2_SEL2_DEBUG
  if (tid == 0) { g_sel2_dbg[3] = __builtin_readcyclecounter(); g_sel2_dbg[7] = dbg_coop; g_sel2_dbg[8] = dbg_tc;
                  g_sel2_dbg[9] = dbg_dealt; g_sel2_dbg[10] = dbg_td; }
#endif
  // __final_insertion_sort == a stable rank inside the 31-wide window (see above)
  for (int p = ta + tid; p < tb; p += NT) {
    const W wp = S.w[p];
    const uint32_t kp = T::key(wp);
    int r = p;
#pragma unroll
    for (int d = 1; d < 16; ++d) {
      const int ql = p - d, qr = p + d;
      const uint32_t kl = T::key(S.w[ql >= 0 ? ql : 0]), kr = T::key(S.w[qr < n ? qr : n - 1]);
      r -= (ql >= 0 && kl > kp) ? 1 : 0;
      r += (qr < n && kr < kp) ? 1 : 0;
    }
    out(r, T::idx(wp));                                        // sorted position r holds original index idx
  }
  __syncthreads();
#ifdef VC2_SEL2_DEBUG
  if (tid == 0) g_sel2_dbg[5] = __builtin_readcyclecounter();
#endif
}

}  // namespace vc2
