// vc2_select.h -- workgroup-parallel emulation of the selection torch.topk performs on CPU.
//
// The reference decides which tokens / channels are kept with torch.topk(largest=False)
// (vidcom2.py:42 and :76).  In bf16/fp16 the scores carry only a handful of distinct values
// per frame, so the kept SET is decided by how libstdc++'s std::nth_element happens to permute
// ties (SURVEY.md finding 2-3, Appendix A).  To be index-exact this file replays that
// algorithm on the (key, index) array held in LDS:
//
//   __introselect:  while (last - first > 3) { depth check -> __heap_select fallback;
//                   median-of-3 to *first; Hoare __unguarded_partition; keep the side with nth }
//                   then __insertion_sort of the <= 3 remaining elements.
//
// A Hoare partition looks serial but is not: the i-th swap always pairs the i-th element from
// the left that is !(a < pivot) with the i-th element from the right that is !(pivot < a), for
// as long as the left position is below the right one.  So one partition round is two prefix
// counts, a rank->position scatter, a (monotone) crossing search and m independent swaps --
// all workgroup-parallel, and bit-for-bit the permutation the serial loop produces.
// The rarely taken pieces (median-of-3, heap-select fallback, final insertion sort,
// partial_sort for k*64 <= n) run serially on lane 0, replaying libstdc++ step by step.
#pragma once

#include "vc2_device.h"

namespace vc2 {

#ifdef VC2_DEBUG_TIMING
__device__ unsigned long long g_dbg_t[512];
__device__ int g_dbg_v[512];
__device__ int g_dbg_n;
struct DbgLds { unsigned long long t[96]; int v[96]; int n; int pad[3]; };
__device__ __forceinline__ DbgLds* dbg_lds() { __shared__ DbgLds d; return &d; }
__device__ __forceinline__ void dbg_stamp(int tag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    DbgLds* d = dbg_lds();
    if (tag == 1) d->n = 0;
    const int i = d->n;
    if (i < 96) { d->t[i] = __builtin_readcyclecounter(); d->v[i] = tag; d->n = i + 1; }
    if (tag == 9) { for (int j = 0; j < d->n; ++j) { g_dbg_t[j] = d->t[j]; g_dbg_v[j] = d->v[j]; } g_dbg_n = d->n; }
  }
}
#else
__device__ __forceinline__ void dbg_stamp(int) {}
#endif

struct SelShared {
  uint32_t* key;   // [n] total-order key (topk_key)
  uint16_t* idx;   // [n] original index
  uint16_t* la;    // [n] left-stop positions, ascending
  uint16_t* lb;    // [n] right-stop positions, descending
  uint32_t* wtot;  // [32] per-wave scan totals / valid-pair counts
};

__host__ __device__ inline size_t sel_shared_bytes(int n) {
  return size_t(n) * (4 + 2 + 2 + 2) + 32 * 4 + 64;
}
__device__ __forceinline__ SelShared sel_carve(unsigned char* smem, int n) {
  SelShared S;
  S.key = reinterpret_cast<uint32_t*>(smem);
  S.wtot = S.key + n;
  S.idx = reinterpret_cast<uint16_t*>(S.wtot + 32);
  S.la = S.idx + n;
  S.lb = S.la + n;
  return S;
}

// ---- serial libstdc++ pieces (lane 0 only) ---------------------------------------------
__device__ __forceinline__ void sel_swap(const SelShared& S, int a, int b) {
  const uint32_t k = S.key[a]; S.key[a] = S.key[b]; S.key[b] = k;
  const uint16_t i = S.idx[a]; S.idx[a] = S.idx[b]; S.idx[b] = i;
}
__device__ __forceinline__ void sel_move(const SelShared& S, int dst, int src) {
  S.key[dst] = S.key[src]; S.idx[dst] = S.idx[src];
}
// bits/stl_algo.h __move_median_to_first
__device__ inline void sel_median_to_first(const SelShared& S, int result, int a, int b, int c) {
  const uint32_t ka = S.key[a], kb = S.key[b], kc = S.key[c];
  if (ka < kb) {
    if (kb < kc) sel_swap(S, result, b);
    else if (ka < kc) sel_swap(S, result, c);
    else sel_swap(S, result, a);
  } else if (ka < kc) sel_swap(S, result, a);
  else if (kb < kc) sel_swap(S, result, c);
  else sel_swap(S, result, b);
}
// bits/stl_heap.h __adjust_heap (+ inlined __push_heap)
__device__ inline void sel_adjust_heap(const SelShared& S, int first, int hole, int len, uint32_t vk,
                                       uint16_t vi) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (S.key[first + child] < S.key[first + child - 1]) child--;
    sel_move(S, first + hole, first + child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    sel_move(S, first + hole, first + child - 1);
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && S.key[first + parent] < vk) {
    sel_move(S, first + hole, first + parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  S.key[first + hole] = vk;
  S.idx[first + hole] = vi;
}
// bits/stl_algo.h __heap_select(first, middle, last)
__device__ inline void sel_heap_select(const SelShared& S, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const uint32_t vk = S.key[first + parent];
      const uint16_t vi = S.idx[first + parent];
      sel_adjust_heap(S, first, parent, len, vk, vi);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i) {
    if (S.key[i] < S.key[first]) {  // __pop_heap(first, middle, i)
      const uint32_t vk = S.key[i];
      const uint16_t vi = S.idx[i];
      sel_move(S, i, first);
      sel_adjust_heap(S, first, 0, len, vk, vi);
    }
  }
}
// bits/stl_algo.h __insertion_sort
__device__ inline void sel_insertion_sort(const SelShared& S, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const uint32_t vk = S.key[i];
    const uint16_t vi = S.idx[i];
    if (vk < S.key[first]) {
      for (int j = i; j > first; --j) sel_move(S, j, j - 1);
      S.key[first] = vk; S.idx[first] = vi;
    } else {  // __unguarded_linear_insert
      int l = i, nx = i - 1;
      while (vk < S.key[nx]) { sel_move(S, l, nx); l = nx; --nx; }
      S.key[l] = vk; S.idx[l] = vi;
    }
  }
}

// Barrier between the phases of a round: a workgroup barrier for NT > 64; for a single wave the DS
// queue already executes in order, so only the compiler must be kept from reordering LDS accesses.
template <int NT> __device__ __forceinline__ void sel_sync() {
  if constexpr (NT > 64) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- workgroup scan of two packed 16-bit counters ----------------------------------------
template <int NT>
__device__ __forceinline__ void block_scan_pair(uint32_t packed, uint32_t& excl, uint32_t& total,
                                                uint32_t* wtot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t v = packed;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if constexpr (NT > 64) {
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      const uint32_t t = wtot[w];
      if (w < wave) base += t;
      tot += t;
    }
    excl = base + v - packed;
    total = tot;
  } else {
    total = __shfl(v, 63, 64);
    excl = v - packed;
  }
}

// Partition rounds of __introselect on [lo, hi) by NT cooperating threads (thread ids tid0 ..
// tid0+NT-1 of the workgroup, all of which must call this with identical arguments).  Runs until the
// range is <= max(3, stop_len) long or the heap-select fallback fired (returns true = finished).
template <int NT>
__device__ bool introselect_rounds(const SelShared& S, int& lo, int& hi, int& depth, int nth, int stop_len,
                                   int tid) {
  while (hi - lo > 3 && hi - lo > stop_len) {
    if (depth == 0) {
      if (tid == 0) { sel_heap_select(S, lo, nth + 1, hi); sel_swap(S, lo, nth); }
      sel_sync<NT>();
      return true;
    }
    --depth;
    if (tid == 0) sel_median_to_first(S, lo, lo + 1, lo + (hi - lo) / 2, hi - 1);
    sel_sync<NT>();
    const uint32_t pk = S.key[lo];
    const int first = lo + 1, len = hi - first;
    const int E = (len + NT - 1) / NT;
    const int b = first + tid * E;
    const int e = min(hi, b + E);
    uint32_t cnt = 0;
    for (int p = b; p < e; ++p) {
      const uint32_t k = S.key[p];
      cnt += (k >= pk ? 1u : 0u) + (k <= pk ? 0x10000u : 0u);
    }
    uint32_t excl, total;
    block_scan_pair<NT>(cnt, excl, total, S.wtot);
    const int totA = int(total & 0xFFFFu), totB = int(total >> 16);
    int ra = int(excl & 0xFFFFu), rb = int(excl >> 16);
    for (int p = b; p < e; ++p) {
      const uint32_t k = S.key[p];
      if (k >= pk) S.la[ra++] = uint16_t(p);
      if (k <= pk) S.lb[totB - 1 - (rb++)] = uint16_t(p);
    }
    sel_sync<NT>();
    // number of swapped pairs m = #{i : la[i] < lb[i]} (a prefix: la ascends, lb descends)
    int l = 0, r = min(totA, totB);
    while (l < r) {
      const int mm = (l + r) >> 1;
      if (S.la[mm] < S.lb[mm]) l = mm + 1; else r = mm;
    }
    const int m = l;
    int cut;
    if (m == 0) {
      cut = totA > 0 ? int(S.la[0]) : hi;
    } else {
      const int a = m < totA ? int(S.la[m]) : hi;
      cut = min(a, int(S.lb[m - 1]));
    }
    for (int i = tid; i < m; i += NT) sel_swap(S, S.la[i], S.lb[i]);
    sel_sync<NT>();
    if (cut <= nth) lo = cut; else hi = cut;
  }
  return false;
}

// ---- register-resident partition rounds for ONE wave --------------------------------------
// The generic rounds above pay ~10 dependent LDS round trips per round.  For ranges of at most
// 64*EM elements a single wave can hold the whole range in registers (slot e of lane l is position
// first + e*64 + l), derive every rank from wave ballots (no scan through LDS), and needs only three
// LDS round trips per round: (1) load keys + pivot candidates, (2) look up the swap partner by rank,
// (3) read the cut.  The permutation produced is identical to the serial libstdc++ loop (see the
// pairing argument at the top of this file).
template <int EM>
__device__ bool introselect_rounds_wave_reg(const SelShared& S, int& lo, int& hi, int& depth, int nth, int lane) {
  while (hi - lo > 3) {
    if (depth == 0) {
      if (lane == 0) { sel_heap_select(S, lo, nth + 1, hi); sel_swap(S, lo, nth); }
      sel_sync<64>();
      return true;
    }
    --depth;
    dbg_stamp(2000000 + (hi - lo));
    const int first = lo + 1, len = hi - first;
    const int E = (len + 63) >> 6;
    const int pa = lo + 1, pb = lo + (hi - lo) / 2, pc = hi - 1;
    // ---- trip 1: pivot candidates + this lane's elements
    const uint32_t klo = S.key[lo], ka = S.key[pa], kb = S.key[pb], kc = S.key[pc];
    const uint32_t ilo = S.idx[lo], ia = S.idx[pa], ib = S.idx[pb], ic = S.idx[pc];
    uint32_t k[EM], ix[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      k[e] = 0; ix[e] = 0;
      if (e < E) {
        const int p = first + e * 64 + lane;
        if (p < hi) { k[e] = S.key[p]; ix[e] = S.idx[p]; }
      }
    }
    // __move_median_to_first(lo, pa, pb, pc): which candidate is the median
    int msrc;
    uint32_t pk, ipk;
    if (ka < kb) {
      if (kb < kc) { msrc = pb; pk = kb; ipk = ib; } else if (ka < kc) { msrc = pc; pk = kc; ipk = ic; }
      else { msrc = pa; pk = ka; ipk = ia; }
    } else if (ka < kc) { msrc = pa; pk = ka; ipk = ia; }
    else if (kb < kc) { msrc = pc; pk = kc; ipk = ic; }
    else { msrc = pb; pk = kb; ipk = ib; }
    {   // iter_swap(lo, msrc): LDS by lane 0, register copy by the lane that holds msrc
      if (lane == 0) {
        S.key[lo] = pk; S.idx[lo] = uint16_t(ipk);
        S.key[msrc] = klo; S.idx[msrc] = uint16_t(ilo);
      }
      const int d = msrc - first, e0 = d >> 6, l0 = d & 63;
#pragma unroll
      for (int e = 0; e < EM; ++e)
        if (e == e0 && lane == l0) { k[e] = klo; ix[e] = ilo; }
    }
    // ---- pass 1: ranks from ballots, scatter stop positions (la: from the left; lb: B numbered from the left)
    int baseA = 0, baseB = 0;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      if (e < E) {
        const int p = first + e * 64 + lane;
        const bool in = p < hi;
        const bool A = in && k[e] >= pk, B = in && k[e] <= pk;
        const unsigned long long bA = __ballot(A), bB = __ballot(B);
        const int rA = baseA + __builtin_amdgcn_mbcnt_hi(uint32_t(bA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bA), 0));
        const int rB = baseB + __builtin_amdgcn_mbcnt_hi(uint32_t(bB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bB), 0));
        if (A) S.la[rA] = uint16_t(p);
        if (B) S.lb[rB] = uint16_t(p);
        baseA += __popcll(bA);
        baseB += __popcll(bB);
      }
    }
    const int totA = baseA, totB = baseB;
    sel_sync<64>();
    // ---- pass 2 (trip 2): partner by rank, validity (la[i] < lb[i]), the swaps themselves
    int m = 0;
    baseA = 0; baseB = 0;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      if (e < E) {
        const int p = first + e * 64 + lane;
        const bool in = p < hi;
        const bool A = in && k[e] >= pk, B = in && k[e] <= pk;
        const unsigned long long bA = __ballot(A), bB = __ballot(B);
        const int rA = baseA + __builtin_amdgcn_mbcnt_hi(uint32_t(bA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bA), 0));
        const int rBl = baseB + __builtin_amdgcn_mbcnt_hi(uint32_t(bB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bB), 0));
        const int rB = totB - 1 - rBl;                        // B rank counted from the right
        int dest = -1;
        bool vA = false;
        if (A && rA < totB) { const int q = S.lb[totB - 1 - rA]; vA = q > p; if (vA) dest = q; }
        if (B && !vA && rB < totA) { const int q = S.la[rB]; if (q < p) dest = q; }
        m += __popcll(__ballot(vA));
        if (dest >= 0) { S.key[dest] = k[e]; S.idx[dest] = uint16_t(ix[e]); }
        baseA += __popcll(bA);
        baseB += __popcll(bB);
      }
    }
    // ---- trip 3: the cut
    int cut;
    if (m == 0) {
      cut = totA > 0 ? int(S.la[0]) : hi;
    } else {
      const int a = m < totA ? int(S.la[m]) : hi;
      cut = min(a, int(S.lb[totB - m]));                      // lb (from the right) [m-1] = lbL[totB-1-(m-1)]
    }
    sel_sync<64>();
    if (cut <= nth) lo = cut; else hi = cut;
  }
  return false;
}

// ---- register-resident partition rounds for a whole workgroup ------------------------------
// For ranges of up to NT*EM elements: wave w owns a contiguous sub-range (slot e of lane l is position
// wb + e*64 + l), in-wave ranks come from ballots, cross-wave bases from one exchange of per-wave
// totals.  Three workgroup barriers per round.
template <int NT, int EM>
__device__ bool introselect_rounds_block_reg(const SelShared& S, int& lo, int& hi, int& depth, int nth,
                                             int stop_len, int tid) {
  constexpr int NW = NT / 64;
  const int lane = tid & 63, wave = tid >> 6;
  while (hi - lo > 3 && hi - lo > stop_len) {
    if (depth == 0) {
      if (tid == 0) { sel_heap_select(S, lo, nth + 1, hi); sel_swap(S, lo, nth); }
      __syncthreads();
      return true;
    }
    --depth;
    dbg_stamp(1000000 + (hi - lo));
    const int first = lo + 1, len = hi - first;
    const int L = (len + NW - 1) / NW;                     // positions per wave
    const int E = (L + 63) >> 6;                           // slots per lane (<= EM by precondition)
    const int wb = first + wave * L;
    const int we = min(hi, wb + L);
    const int pa = lo + 1, pb = lo + (hi - lo) / 2, pc = hi - 1;
    // ---- trip 1
    const uint32_t klo = S.key[lo], ka = S.key[pa], kb = S.key[pb], kc = S.key[pc];
    const uint32_t ilo = S.idx[lo], ia = S.idx[pa], ib = S.idx[pb], ic = S.idx[pc];
    uint32_t k[EM], ix[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      k[e] = 0; ix[e] = 0;
      if (e < E) {
        const int p = wb + e * 64 + lane;
        if (p < we) { k[e] = S.key[p]; ix[e] = S.idx[p]; }
      }
    }
    int msrc;
    uint32_t pk, ipk;
    if (ka < kb) {
      if (kb < kc) { msrc = pb; pk = kb; ipk = ib; } else if (ka < kc) { msrc = pc; pk = kc; ipk = ic; }
      else { msrc = pa; pk = ka; ipk = ia; }
    } else if (ka < kc) { msrc = pa; pk = ka; ipk = ia; }
    else if (kb < kc) { msrc = pc; pk = kc; ipk = ic; }
    else { msrc = pb; pk = kb; ipk = ib; }
    {
      const int d = msrc - first, wm = d / L, off = d - wm * L, e0 = off >> 6, l0 = off & 63;
#pragma unroll
      for (int e = 0; e < EM; ++e)
        if (wave == wm && e == e0 && lane == l0) { k[e] = klo; ix[e] = ilo; }
    }
    // ---- in-wave ranks, per-wave totals
    int rA[EM], rB[EM];
    int cA = 0, cB = 0;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      rA[e] = 0; rB[e] = 0;
      if (e < E) {
        const int p = wb + e * 64 + lane;
        const bool in = p < we;
        const bool A = in && k[e] >= pk, B = in && k[e] <= pk;
        const unsigned long long bA = __ballot(A), bB = __ballot(B);
        rA[e] = cA + __builtin_amdgcn_mbcnt_hi(uint32_t(bA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bA), 0));
        rB[e] = cB + __builtin_amdgcn_mbcnt_hi(uint32_t(bB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bB), 0));
        cA += __popcll(bA);
        cB += __popcll(bB);
      }
    }
    if (lane == 0) S.wtot[wave] = uint32_t(cA) | (uint32_t(cB) << 16);
    __syncthreads();                                        // also orders every trip-1 read before the swaps
    if (tid == 0) {                                         // iter_swap(lo, msrc)
      S.key[lo] = pk; S.idx[lo] = uint16_t(ipk);
      S.key[msrc] = klo; S.idx[msrc] = uint16_t(ilo);
    }
    int baseA = 0, baseB = 0, totA = 0, totB = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t t = S.wtot[w];
      if (w < wave) { baseA += int(t & 0xFFFFu); baseB += int(t >> 16); }
      totA += int(t & 0xFFFFu);
      totB += int(t >> 16);
    }
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      if (e < E) {
        const int p = wb + e * 64 + lane;
        const bool in = p < we;
        if (in && k[e] >= pk) S.la[baseA + rA[e]] = uint16_t(p);
        if (in && k[e] <= pk) S.lb[baseB + rB[e]] = uint16_t(p);     // B numbered from the left
      }
    }
    __syncthreads();
    // ---- partners, validity, swaps
    int cV = 0;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
      if (e < E) {
        const int p = wb + e * 64 + lane;
        const bool in = p < we;
        const bool A = in && k[e] >= pk, B = in && k[e] <= pk;
        const int ra = baseA + rA[e];
        const int rb = totB - 1 - (baseB + rB[e]);              // B rank counted from the right
        int dest = -1;
        bool vA = false;
        if (A && ra < totB) { const int q = S.lb[totB - 1 - ra]; vA = q > p; if (vA) dest = q; }
        if (B && !vA && rb < totA) { const int q = S.la[rb]; if (q < p) dest = q; }
        cV += __popcll(__ballot(vA));
        if (dest >= 0) { S.key[dest] = k[e]; S.idx[dest] = uint16_t(ix[e]); }
      }
    }
    if (lane == 0) S.wtot[16 + wave] = uint32_t(cV);
    __syncthreads();
    int m = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) m += int(S.wtot[16 + w]);
    int cut;
    if (m == 0) {
      cut = totA > 0 ? int(S.la[0]) : hi;
    } else {
      const int a = m < totA ? int(S.la[m]) : hi;
      cut = min(a, int(S.lb[totB - m]));
    }
    __syncthreads();
    if (cut <= nth) lo = cut; else hi = cut;
  }
  return false;
}

// One libstdc++ __unguarded_partition_pivot(lo, hi) executed by ONE wave on [lo, hi) (hi - lo > 3):
// median-of-3 to lo, then the Hoare partition of [lo+1, hi).  Nothing is cached in registers -- the two
// passes stream the range from LDS 64 positions at a time and ranks come from wave ballots.  The la/lb
// scratch is addressed inside [lo, hi) only, so several waves may partition DISJOINT ranges at once.
// Returns the cut (identical on every lane).
__device__ inline int wave_partition(const SelShared& S, int lo, int hi, int lane) {
  const int first = lo + 1, len = hi - first;
  const int E = (len + 63) >> 6;
  if (lane == 0) sel_median_to_first(S, lo, lo + 1, lo + (hi - lo) / 2, hi - 1);
  sel_sync<64>();
  const uint32_t pk = S.key[lo];
  uint16_t* la = S.la + first;
  uint16_t* lb = S.lb + first;
  int baseA = 0, baseB = 0;
  for (int e = 0; e < E; ++e) {
    const int p = first + e * 64 + lane;
    const bool in = p < hi;
    const uint32_t kk = in ? S.key[p] : 0u;
    const bool A = in && kk >= pk, B = in && kk <= pk;
    const unsigned long long bA = __ballot(A), bB = __ballot(B);
    const int rA = baseA + __builtin_amdgcn_mbcnt_hi(uint32_t(bA >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bA), 0));
    const int rB = baseB + __builtin_amdgcn_mbcnt_hi(uint32_t(bB >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bB), 0));
    if (A) la[rA] = uint16_t(p);
    if (B) lb[rB] = uint16_t(p);                       // B numbered from the left
    baseA += __popcll(bA);
    baseB += __popcll(bB);
  }
  const int totA = baseA, totB = baseB;
  sel_sync<64>();
  // number of swapped pairs: la[i] < lb_right[i] is a prefix property -> count it in parallel
  int m = 0;
  {
    const int np = min(totA, totB);
    for (int i0 = 0; i0 < np; i0 += 64) {
      const int i = i0 + lane;
      const bool v = i < np && la[i] < lb[totB - 1 - i];
      const unsigned long long b = __ballot(v);
      m += __popcll(b);
      if (b != ~0ull) break;
    }
  }
  int cut;
  if (m == 0) {
    cut = totA > 0 ? int(la[0]) : hi;
  } else {
    const int a = m < totA ? int(la[m]) : hi;
    cut = min(a, int(lb[totB - m]));
  }
  for (int i = lane; i < m; i += 64) sel_swap(S, la[i], lb[totB - 1 - i]);
  sel_sync<64>();
  return cut;
}

__device__ inline bool introselect_rounds_wave_stream(const SelShared& S, int& lo, int& hi, int& depth, int nth,
                                                      int stop_len, int lane) {
  while (hi - lo > 3 && hi - lo > stop_len) {
    if (depth == 0) {
      if (lane == 0) { sel_heap_select(S, lo, nth + 1, hi); sel_swap(S, lo, nth); }
      sel_sync<64>();
      return true;
    }
    --depth;
    dbg_stamp(3000000 + (hi - lo));
    const int cut = wave_partition(S, lo, hi, lane);
    if (cut <= nth) lo = cut; else hi = cut;
  }
  return false;
}

// std::nth_element(first, first + nth, first + n) on the (key, idx) array in LDS.  Long ranges
// (> 4096) start with workgroup-parallel rounds; then wave 0 finishes alone: streaming rounds down to
// 512 elements, register-resident rounds below.
// All NT threads of the workgroup must call this.
template <int NT>
__device__ void introselect_block(const SelShared& S, int n, int nth) {
  if (n == 0 || nth >= n) return;
  const int tid = threadIdx.x;
  int lo = 0, hi = n;
  int depth = 2 * (31 - __clz(n));                       // std::__lg(n) * 2
  bool done = false;
  if constexpr (NT > 64) {
    static_assert(NT <= 1024, "per-wave totals live in 16 slots");
    done = introselect_rounds<NT>(S, lo, hi, depth, nth, NT * 4, tid);          // only for n > NT*4
    if (!done) done = introselect_rounds_block_reg<NT, 4>(S, lo, hi, depth, nth, 512, tid);
  }
  if (!done && tid < 64) {
    done = introselect_rounds_wave_stream(S, lo, hi, depth, nth, 512, tid);
    if (!done) done = introselect_rounds_wave_reg<8>(S, lo, hi, depth, nth, tid);
    dbg_stamp(4000000 + (hi - lo));
    if (!done && tid == 0) sel_insertion_sort(S, lo, hi);
  }
  sel_sync<NT>();
  dbg_stamp(5000000);
}

// torch.topk(v, k, largest=False) SET: afterwards positions [0, k) of (key, idx) hold the kept
// elements (ATen/native/TopKImpl.h: partial_sort when k*64 <= n, else nth_element(k-1)).
template <int NT>
__device__ void topk_smallest_block(const SelShared& S, int n, int k) {
  if (k <= 0 || k >= n) return;                      // k == n: everything kept
  if (int64_t(k) * 64 <= int64_t(n)) {
    if (threadIdx.x == 0) sel_heap_select(S, 0, k, n);  // partial_sort = heap_select + sort_heap
    sel_sync<NT>();                                     // (sort_heap only permutes the first k)
  } else {
    introselect_block<NT>(S, n, k - 1);
  }
}

// bits/stl_heap.h __sort_heap(first, last) on a heap (serial)
__device__ inline void sel_sort_heap(const SelShared& S, int first, int last) {
  while (last - first > 1) {
    --last;
    const uint32_t vk = S.key[last];
    const uint16_t vi = S.idx[last];
    sel_move(S, last, first);
    sel_adjust_heap(S, first, 0, last - first, vk, vi);
  }
}

// ---- std::sort(first, first + n) replay (the `sorted=True` half of torch.topk, TopKImpl.h) ------------
// __introsort_loop: every segment longer than 16 is partitioned (same __unguarded_partition_pivot as
// above) and both halves recurse with depth_limit-1; at depth 0 a segment is heap-sorted instead.
// Segments of one recursion level are independent, so the workgroup processes the tree level by level,
// one wave per segment.  __final_insertion_sort then equals a STABLE sort inside every leaf segment
// (elements never cross a cut: everything left of a cut is <= everything right of it and the insertion
// uses a strict compare), done here by rank counting.  out_order[p] = original index at sorted position p.
struct SortScratch {
  uint32_t* segA;   // [kMaxSeg] packed (first | last << 13 | depth << 26)
  uint32_t* segB;
  int* cnt;         // [2]
  uint8_t* bnd;     // [n] 1 = a leaf starts here
};
constexpr int kMaxSeg = 1024;
__host__ __device__ inline size_t sort_scratch_bytes(int n) { return size_t(kMaxSeg) * 8 + 16 + size_t(n) + 16; }
__device__ __forceinline__ SortScratch sort_carve(unsigned char* p, int n) {
  SortScratch Q;
  Q.segA = reinterpret_cast<uint32_t*>(p);
  Q.segB = Q.segA + kMaxSeg;
  Q.cnt = reinterpret_cast<int*>(Q.segB + kMaxSeg);
  Q.bnd = reinterpret_cast<uint8_t*>(Q.cnt + 4);
  (void)n;
  return Q;
}

template <int NT>
__device__ void introsort_block(const SelShared& S, const SortScratch& Q, int n, int* __restrict__ out_order) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  for (int p = tid; p < n; p += NT) Q.bnd[p] = (p == 0) ? 1 : 0;
  if (tid == 0) {
    Q.cnt[0] = 0; Q.cnt[1] = 0;
    if (n > 16) { Q.segA[0] = uint32_t(0) | (uint32_t(n) << 13) | (uint32_t(2 * (31 - __clz(n))) << 26); Q.cnt[0] = 1; }
  }
  __syncthreads();
  uint32_t* cur = Q.segA;
  uint32_t* nxt = Q.segB;
  int ci = 0;
  while (true) {
    const int ns = Q.cnt[ci];
    dbg_stamp(6000000 + ns);
    if (ns == 0) break;
    for (int si = wave; si < ns; si += NW) {
      const uint32_t sg = cur[si];
      const int first = int(sg & 0x1FFFu), last = int((sg >> 13) & 0x1FFFu), depth = int(sg >> 26);
      if (depth == 0) {                                   // __partial_sort(first, last, last): heapsort
        if (lane == 0) { sel_heap_select(S, first, last, last); sel_sort_heap(S, first, last); }
        for (int p = first + lane; p < last; p += 64) Q.bnd[p] = 1;      // already final: one leaf per element
        sel_sync<64>();
      } else {
        const int cut = wave_partition(S, first, last, lane);
        if (lane == 0) {
          Q.bnd[cut] = 1;
          if (cut - first > 16) {
            const int j = atomicAdd(&Q.cnt[ci ^ 1], 1);
            nxt[j] = uint32_t(first) | (uint32_t(cut) << 13) | (uint32_t(depth - 1) << 26);
          }
          if (last - cut > 16) {
            const int j = atomicAdd(&Q.cnt[ci ^ 1], 1);
            nxt[j] = uint32_t(cut) | (uint32_t(last) << 13) | (uint32_t(depth - 1) << 26);
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) Q.cnt[ci] = 0;
    ci ^= 1;
    uint32_t* t = cur; cur = nxt; nxt = t;
    __syncthreads();
  }
  dbg_stamp(7000000);
  // stable sort inside each leaf == __final_insertion_sort.  Leaf id = prefix count of the boundary flags;
  // leaf starts are scattered by id (into la, free now), so every element finds [ls, le) in two reads.
  {
    const int E = (n + NT - 1) / NT;
    const int b = tid * E, e = min(n, b + E);
    uint32_t cnt = 0;
    for (int p = b; p < e; ++p) cnt += Q.bnd[p];
    uint32_t excl, tot;
    block_scan_pair<NT>(cnt, excl, tot, S.wtot);
    uint32_t id = excl;                                   // leaves before position b
    for (int p = b; p < e; ++p) {
      if (Q.bnd[p]) { S.la[id] = uint16_t(p); ++id; }
      S.lb[p] = uint16_t(id - 1);                         // leaf id of position p
    }
    if (tid == 0) S.la[tot] = uint16_t(n);                // sentinel end
    __syncthreads();
    for (int p = tid; p < n; p += NT) {
      const int lid = S.lb[p];
      const int ls = S.la[lid], le = S.la[lid + 1];
      const uint32_t kp = S.key[p];
      int r = ls;
#pragma unroll 4
      for (int q = ls; q < le; ++q) {
        const uint32_t kq = S.key[q];
        r += (kq < kp || (kq == kp && q < p)) ? 1 : 0;
      }
      out_order[r] = int(S.idx[p]);
    }
  }
  __syncthreads();
  dbg_stamp(8000000);
}

}  // namespace vc2
