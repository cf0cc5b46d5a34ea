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

struct SelShared {
  uint32_t* key;   // [n] total-order key (topk_key)
  uint16_t* idx;   // [n] original index
  uint16_t* la;    // [n] left-stop positions, ascending
  uint16_t* lb;    // [n] right-stop positions, descending
  uint32_t* wtot;  // [16] per-wave scan totals
};

__device__ __forceinline__ size_t sel_shared_bytes(int n) {
  return size_t(n) * (4 + 2 + 2 + 2) + 16 * 4 + 64;
}
__device__ __forceinline__ SelShared sel_carve(unsigned char* smem, int n) {
  SelShared S;
  S.key = reinterpret_cast<uint32_t*>(smem);
  S.wtot = S.key + n;
  S.idx = reinterpret_cast<uint16_t*>(S.wtot + 16);
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

// std::nth_element(first, first + nth, first + n) on the (key, idx) array in LDS.  The first rounds
// (long ranges) use all NT threads of the workgroup; once the range is <= kWaveTail elements wave 0
// finishes alone, wave-synchronously (no workgroup barriers).  All NT threads must call this.
constexpr int kWaveTail = 1024;

template <int NT>
__device__ void introselect_block(const SelShared& S, int n, int nth) {
  if (n == 0 || nth >= n) return;
  const int tid = threadIdx.x;
  int lo = 0, hi = n;
  int depth = 2 * (31 - __clz(n));                       // std::__lg(n) * 2
  bool done = false;
  if constexpr (NT > 64) {
    done = introselect_rounds<NT>(S, lo, hi, depth, nth, kWaveTail, tid);
  }
  if (!done && tid < 64) {
    done = introselect_rounds<64>(S, lo, hi, depth, nth, 3, tid);
    if (!done && tid == 0) sel_insertion_sort(S, lo, hi);
  }
  sel_sync<NT>();
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

}  // namespace vc2
