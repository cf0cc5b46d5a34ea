// vc2_oracle.cpp -- CPU restatement ("oracle") of the VidCom2 token-compression hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under vidcom2_amd/ may import, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
// the checker / the timed CPU baseline -- never as the product path.
//
// Parity pin: the reference ships no tests or golden vectors for this path,
// so this restatement is pinned against outputs of the reference itself, imported in the
// build container: tests/golden/*.json, produced by tests/golden/make_golden.py
// (committed).  Those fixtures are the parity anchor; see DESIGN.md "Numerics contract".
//
// Numerics contract restated here ("correctly-rounded op semantics"): the reference
// (token_compressor/vidcom2/vidcom2.py) runs every torch op in the input dtype T.  Each
// torch op is modelled as  RN_T(RN_f32(exact result of that one op))  -- i.e. exact
// (double-accumulated) reductions, IEEE fp32 for single arithmetic ops, then a
// round-to-nearest-even cast to T.  torch's CPU kernels differ from this only by their
// own fp32 accumulation noise (<= ~1e-6 relative), which changes a T-rounded result with
// probability ~2.5e-5 per bf16 value (measured).  Selection (torch.topk) is
// libstdc++ std::nth_element / std::partial_sort / std::sort over (value,index) pairs,
// exactly as ATen/native/TopKImpl.h does it -- this file calls the same libstdc++
// algorithms, so ties break identically.
//
// Build: g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).
// -ffp-contract=off matters: a fused multiply-add would skip the fp32 rounding of the
// product that the per-op model (and torch) performs.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace {

inline size_t Z(int64_t v) { return static_cast<size_t>(v); }

enum DType { F32 = 0, BF16 = 1, F16 = 2 };

// ---------------------------------------------------------------- dtype conversions
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline float bf16_to_f32(uint16_t h) { return bits_f32(uint32_t(h) << 16); }
inline uint16_t f32_to_bf16(float f) {  // RNE, NaN -> quiet NaN (c10::BFloat16 semantics)
  uint32_t u = f32_bits(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
  uint32_t lsb = (u >> 16) & 1u;
  return uint16_t((u + 0x7FFFu + lsb) >> 16);
}

inline float f16_to_f32(uint16_t h) {
  uint32_t sign = uint32_t(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
  if (exp == 0) {
    if (man == 0) return bits_f32(sign);
    float v = std::ldexp(float(man), -24);  // subnormal
    return sign ? -v : v;
  }
  if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
  return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}
inline uint16_t f32_to_f16(float f) {  // RNE incl. subnormals/overflow (c10::Half semantics)
  uint32_t u = f32_bits(f);
  uint16_t sign = uint16_t((u >> 16) & 0x8000u);
  uint32_t a = u & 0x7FFFFFFFu;
  if (a > 0x7F800000u) return uint16_t(sign | 0x7E00u);
  if (a >= 0x47800000u) {                       // >= 65536 -> may round to inf
    return uint16_t(sign | 0x7C00u);            // (65520..65536 handled below by rounding)
  }
  if (a < 0x33000000u) return sign;             // < 2^-25 -> 0
  int e = int(a >> 23) - 127;
  uint32_t m = (a & 0x7FFFFFu) | 0x800000u;     // 24-bit significand
  int shift;                                    // bits to drop
  if (e < -14) shift = 13 + (-14 - e); else shift = 13;
  uint32_t kept = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (kept & 1u))) kept++;
  uint32_t out;
  if (e < -14) out = kept;                      // subnormal (kept may carry into exp=1: fine)
  else out = (uint32_t(e + 15) << 10) + (kept - 0x400u);
  if (out >= 0x7C00u) out = 0x7C00u;
  return uint16_t(sign | out);
}

inline float load_T(const void* p, int64_t i, int dt) {
  switch (dt) {
    case F32: return static_cast<const float*>(p)[i];
    case BF16: return bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
    default: return f16_to_f32(static_cast<const uint16_t*>(p)[i]);
  }
}
inline void store_T(void* p, int64_t i, int dt, float v) {
  switch (dt) {
    case F32: static_cast<float*>(p)[i] = v; break;
    case BF16: static_cast<uint16_t*>(p)[i] = f32_to_bf16(v); break;
    default: static_cast<uint16_t*>(p)[i] = f32_to_f16(v); break;
  }
}
// RN_T of an fp32 value, returned widened to fp32.
inline float rT(float v, int dt) {
  switch (dt) {
    case F32: return v;
    case BF16: return bf16_to_f32(f32_to_bf16(v));
    default: return f16_to_f32(f32_to_f16(v));
  }
}
inline float rTd(double v, int dt) { return rT(float(v), dt); }  // RN_T(RN_f32(exact))

// ---------------------------------------------------------------- torch.topk (CPU) twin
// ATen/native/TopKImpl.h:31-92 (largest=False branch): pairs in index order, comparator
// "value < with NaN last", partial_sort when k*64 <= n, else nth_element (+ sort of the
// first k-1 when sorted=True).
using elem_t = std::pair<float, int64_t>;
inline bool less_nan_last(const elem_t& x, const elem_t& y) {
  return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first));
}
void topk_smallest(const float* v, int64_t n, int64_t k, bool sorted, int64_t* idx_out,
                   std::vector<elem_t>& q) {
  if (k == 0) return;
  q.resize(Z(n));
  for (int64_t j = 0; j < n; ++j) q[Z(j)] = {v[j], j};
  if (k * 64 <= n) {
    std::partial_sort(q.begin(), q.begin() + k, q.end(), less_nan_last);
  } else {
    std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), less_nan_last);
    if (sorted) std::sort(q.begin(), q.begin() + (k - 1), less_nan_last);
  }
  for (int64_t j = 0; j < k; ++j) idx_out[j] = q[Z(j)].second;
}

// ---------------------------------------------------------------- torch accumulation orders
// Mode 1 ("torch order") replays the fp32 accumulation ORDER of the two reductions whose rounding noise
// can flip a half-precision result (measured with the imported reference, tests/golden/make_golden.py):
//   * L2 norm over the last dim (ReduceOpsKernel.cpp norm_kernel_tensor_iterator_impl): fp32/bf16 use
//     8 interleaved fp32 FMA chains (element p goes to chain p % 8), chains then added 0..7, tail after;
//     fp16 takes the generic path: one sequential fp32 chain.
//   * sum over the last dim (SumKernel.cpp cascade_sum): 8 fp32 lanes x 4 interleaved vectors, 4 cascade
//     levels; half types first add the two 8-element halves of each 16-element chunk.
// Mode 0 ("exact", default) accumulates in double: the correctly rounded result.
int g_mode = 0;

inline int ceil_log2_i64(int64_t x) { int r = 0; while ((int64_t(1) << r) < x) ++r; return r; }

float norm_torch_order(const float* v, int64_t n, int dt) {
  if (dt == F16) {
    float s = 0.f;
    for (int64_t i = 0; i < n; ++i) s = s + v[i] * v[i];         // products of fp16 values are exact in fp32
    return std::sqrt(s);
  }
  const int64_t step = (dt == F32) ? 8 : 16;                     // Vec<T>::size(); fp32 lanes = 8
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t d = 0;
  for (; d + step <= n; d += step)
    for (int64_t h = 0; h < step; h += 8)
      for (int j = 0; j < 8; ++j) acc[j] = std::fmaf(v[d + h + j], v[d + h + j], acc[j]);
  float s = acc[0];
  for (int j = 1; j < 8; ++j) s = s + acc[j];
  for (; d < n; ++d) s = std::fmaf(v[d], v[d], s);
  return std::sqrt(s);
}

float sum_torch_order(const float* v, int64_t n, int dt) {
  constexpr int W = 8, ILP = 4, LEVELS = 4;
  const int64_t chunk = (dt == F32) ? 8 : 16;
  const int64_t vec_size = n / chunk;
  auto load = [&](int64_t c, float* out) {
    const float* p = v + c * chunk;
    if (dt == F32) { for (int j = 0; j < W; ++j) out[j] = p[j]; }
    else { for (int j = 0; j < W; ++j) out[j] = p[j] + p[W + j]; }
  };
  const int64_t size_ilp = vec_size / ILP;
  const int64_t level_power = std::max<int64_t>(4, ceil_log2_i64(size_ilp) / LEVELS);
  const int64_t level_step = int64_t(1) << level_power, level_mask = level_step - 1;
  float acc[LEVELS][ILP][W];
  std::memset(acc, 0, sizeof(acc));
  float t[W];
  int64_t i = 0;
  for (; i + level_step <= size_ilp;) {
    for (int64_t j = 0; j < level_step; ++j, ++i)
      for (int k = 0; k < ILP; ++k) { load(i * ILP + k, t); for (int l = 0; l < W; ++l) acc[0][k][l] += t[l]; }
    for (int j = 1; j < LEVELS; ++j) {
      for (int k = 0; k < ILP; ++k)
        for (int l = 0; l < W; ++l) { acc[j][k][l] += acc[j - 1][k][l]; acc[j - 1][k][l] = 0.f; }
      const int64_t mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < size_ilp; ++i)
    for (int k = 0; k < ILP; ++k) { load(i * ILP + k, t); for (int l = 0; l < W; ++l) acc[0][k][l] += t[l]; }
  for (int j = 1; j < LEVELS; ++j)
    for (int k = 0; k < ILP; ++k)
      for (int l = 0; l < W; ++l) acc[0][k][l] += acc[j][k][l];
  for (int64_t c = size_ilp * ILP; c < vec_size; ++c) { load(c, t); for (int l = 0; l < W; ++l) acc[0][0][l] += t[l]; }
  for (int k = 1; k < ILP; ++k)
    for (int l = 0; l < W; ++l) acc[0][0][l] += acc[0][k][l];
  float fin = 0.f;
  for (int64_t k = vec_size * chunk; k < n; ++k) fin += v[k];
  for (int l = 0; l < W; ++l) fin += acc[0][0][l];
  return fin;
}

// torch mean for every dtype on CPU: fp32 sum -> fp32 div by count -> cast
// (ATen/native/ReduceOps.cpp mean_out: "cast_fp32 -> sum -> div -> cast").
inline float mean_T(double exact_sum, int64_t count, int dt) {
  float s = float(exact_sum);
  return rT(s / float(count), dt);
}


// Column sum of a contiguous fp32 [rows, C] block as torch's outer reduction adds it
// (ATen/native/cpu/SumKernel.cpp: vectorized_outer_sum -> multi_row_sum for full groups of 32 columns,
// row_sum -- four row-interleaved chains -- for the columns after the last full group; which path the
// tail columns take depends on how torch splits the columns over threads, the single-call rule is used).
// `v` points at element (first row, column), `stride` = C.  Used for the half-precision centre means,
// where ReduceOps.cpp mean_out sums an fp32 copy, divides by the count in fp32 and casts.
float cascade_rows(const float* v, int64_t stride, int64_t step_rows, int64_t n) {   // multi_row_sum, 1 column
  const int64_t level_power = std::max<int64_t>(4, ceil_log2_i64(n) / 4);
  const int64_t level_step = int64_t(1) << level_power, level_mask = level_step - 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t i = 0;
  for (; i + level_step <= n;) {
    for (int64_t j = 0; j < level_step; ++j, ++i) acc[0] += v[i * step_rows * stride];
    for (int j = 1; j < 4; ++j) {
      acc[j] += acc[j - 1];
      acc[j - 1] = 0.f;
      const int64_t mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < n; ++i) acc[0] += v[i * step_rows * stride];
  for (int j = 1; j < 4; ++j) acc[0] += acc[j];
  return acc[0];
}
float col_sum_torch_order(const float* base, int64_t C, int64_t j, int64_t rows) {
  const int64_t group = (C >= 8) ? 32 : 4;
  const float* v = base + j;
  if (j < (C / group) * group) return cascade_rows(v, C, 1, rows);
  const int64_t size_ilp = rows / 4;                         // row_sum: rows as a (-1, 4) array
  float part[4];
  for (int k = 0; k < 4; ++k) part[k] = cascade_rows(v + k * C, C, 4, size_ilp);
  for (int64_t i = size_ilp * 4; i < rows; ++i) part[0] += v[i * C];
  for (int k = 1; k < 4; ++k) part[0] += part[k];
  return part[0];
}

}  // namespace

extern "C" {

// 0 = exact accumulation (default), 1 = torch's CPU accumulation order for the L2 norm and the row sums
int vc2o_set_mode(int mode) { g_mode = mode; return 0; }

// vidcom2.py:40  variances = x.var(dim=0, unbiased=False)  -> T[D]
int vc2o_chan_var(const void* x, int64_t R, int64_t D, int dt, void* var_out) {
  std::vector<double> s(Z(D), 0.0), m2(Z(D), 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t c0 = 0; c0 < D; c0 += 64) {
    int64_t c1 = std::min<int64_t>(D, c0 + 64);
    for (int64_t r = 0; r < R; ++r)
      for (int64_t c = c0; c < c1; ++c) s[Z(c)] += double(load_T(x, r * D + c, dt));
    for (int64_t c = c0; c < c1; ++c) s[Z(c)] /= double(R);
    for (int64_t r = 0; r < R; ++r)
      for (int64_t c = c0; c < c1; ++c) {
        double d = double(load_T(x, r * D + c, dt)) - s[Z(c)];
        m2[Z(c)] += d * d;
      }
  }
  for (int64_t c = 0; c < D; ++c) store_T(var_out, c, dt, float(m2[Z(c)] / double(R)));
  return 0;
}

// torch.topk(values, k, largest=False, sorted=sorted) on a T[n] vector -> int64[k]
int vc2o_topk_smallest(const void* vals, int64_t n, int64_t k, int sorted, int dt,
                       int64_t* idx_out) {
  std::vector<float> v(Z(n));
  for (int64_t j = 0; j < n; ++j) v[Z(j)] = load_T(vals, j, dt);
  std::vector<elem_t> q;
  topk_smallest(v.data(), n, k, sorted != 0, idx_out, q);
  return 0;
}

// vidcom2.py:38-43  select_low_var_channels: var -> topk(k=int(D*ratio), smallest) -> idx
int vc2o_select_low_var_channels(const void* x, int64_t R, int64_t D, int dt, int64_t k,
                                 void* var_out /*T[D] or null*/, int64_t* idx_out) {
  std::vector<uint8_t> tmp(Z(D) * 4);
  void* var = var_out ? var_out : static_cast<void*>(tmp.data());
  vc2o_chan_var(x, R, D, dt, var);
  return vc2o_topk_smallest(var, D, k, 1, dt, idx_out);
}

// vidcom2.py:45-62  compute_gaussian_scores(x[:, idx], tpf) -> (v_score, f_score) T[F,N]
//   x: T[R, D] row-major (R = F*N), idx: int64[C] the selected channels.
//   Optional debug outputs (may be null): norm T[R], vid_center T[C], frame_center T[F,C],
//   dist_v / dist_f T[R].
// Sharded form (frame-sharded multi-GPU): csum_out (double[C], may be null) receives
// this shard's sum of normalised tokens; csum_in (double[P][C], may be null) + R_total give the video
// centre of the WHOLE video instead of the local one.
int vc2o_gaussian_scores_ex(const void* x, int64_t R, int64_t D, int dt, const int64_t* idx,
                            int64_t C, int64_t tpf, const double* csum_in, int64_t P, int64_t R_total,
                            double* csum_out, void* v_out, void* f_out, void* norm_out,
                            void* vc_out, void* fc_out, void* dv_out, void* df_out) {
  if (tpf <= 0 || R % tpf != 0) return -2;  // torch: .view(-1, tpf, C) RuntimeError
  const int64_t F = R / tpf, N = tpf;
  // F.normalize: x / x.norm(2, -1, keepdim).clamp_min(1e-12).expand_as(x)
  std::vector<float> xh(Z(R) * Z(C));
  std::vector<float> nrm(Z(R));
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    float norm;
    if (g_mode == 1) {
      std::vector<float> row(Z(C));
      for (int64_t j = 0; j < C; ++j) row[Z(j)] = load_T(x, r * D + idx[j], dt);
      norm = rT(norm_torch_order(row.data(), C, dt), dt);
    } else {
      double n2 = 0.0;
      for (int64_t j = 0; j < C; ++j) {
        double v = double(load_T(x, r * D + idx[j], dt));
        n2 += v * v;
      }
      norm = rTd(std::sqrt(n2), dt);
    }
    float den = rT(std::max(norm, 1e-12f), dt);      // clamp_min(eps) computed in fp32 -> T
    if (std::isnan(norm)) den = norm;
    nrm[Z(r)] = norm;
    for (int64_t j = 0; j < C; ++j)
      xh[Z(r * C + j)] = rT(load_T(x, r * D + idx[j], dt) / den, dt);
  }
  // centres (vidcom2.py:51-52)
  std::vector<double> fsum(Z(F) * Z(C), 0.0);
  std::vector<float> fc(Z(F) * Z(C)), vc(Z(C));
#pragma omp parallel for schedule(static)
  for (int64_t f = 0; f < F; ++f)
    for (int64_t n = 0; n < N; ++n)
      for (int64_t j = 0; j < C; ++j)
        fsum[Z(f * C + j)] += double(xh[Z((f * N + n) * C + j)]);
  // torch mode, half precision: the fp32 sum inside mean() is order dependent (mean_out casts to fp32,
  // sums with the outer-reduction cascade, divides, casts back); replayed for the unsharded pass.  The
  // frame-sharded pass only has per-rank sums of the video centre and keeps the exact sum there.
  const bool replay = g_mode == 1 && dt != F32;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < C; ++j) {
    double t = 0.0;
    for (int64_t f = 0; f < F; ++f) {
      t += fsum[Z(f * C + j)];
      fc[Z(f * C + j)] = replay ? rT(col_sum_torch_order(xh.data() + Z(f * N * C), C, j, N) / float(N), dt)
                                : mean_T(fsum[Z(f * C + j)], N, dt);
    }
    if (csum_out) csum_out[j] = t;
    if (csum_in) {
      t = 0.0;
      for (int64_t p = 0; p < P; ++p) t += csum_in[p * C + j];
      vc[Z(j)] = mean_T(t, R_total, dt);
    } else {
      vc[Z(j)] = replay ? rT(col_sum_torch_order(xh.data(), C, j, R) / float(R), dt) : mean_T(t, R, dt);
    }
  }
  // _multi_scale_gaussian (vidcom2.py:59-62), alphas = 2^-3..2^1
  static const float two_a[5] = {0.25f, 0.5f, 1.0f, 2.0f, 4.0f};
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const int64_t f = r / N;
    float dist[2];
    if (g_mode == 1) {
      std::vector<float> qa(Z(C)), qb(Z(C));
      for (int64_t j = 0; j < C; ++j) {
        float xv = xh[Z(r * C + j)];
        float a = rT(xv - vc[Z(j)], dt);
        float b = rT(xv - fc[Z(f * C + j)], dt);
        qa[Z(j)] = rT(a * a, dt);
        qb[Z(j)] = rT(b * b, dt);
      }
      dist[0] = rT(sum_torch_order(qa.data(), C, dt), dt);
      dist[1] = rT(sum_torch_order(qb.data(), C, dt), dt);
    } else {
      double sv = 0.0, sf = 0.0;
      for (int64_t j = 0; j < C; ++j) {
        float xv = xh[Z(r * C + j)];
        float a = rT(xv - vc[Z(j)], dt);
        float b = rT(xv - fc[Z(f * C + j)], dt);
        sv += double(rT(a * a, dt));
        sf += double(rT(b * b, dt));
      }
      dist[0] = rTd(sv, dt);
      dist[1] = rTd(sf, dt);
    }
    float score[2];
    for (int w = 0; w < 2; ++w) {
      float acc = 0.0f;
      for (int a = 0; a < 5; ++a) {
        float arg = rT((-dist[w]) / two_a[a], dt);
        float e = rTd(std::exp(double(arg)), dt);
        acc = (a == 0) ? e : rT(acc + e, dt);        // Python sum(): 0 + t1 is exact
      }
      score[w] = acc;
    }
    store_T(v_out, r, dt, score[0]);
    store_T(f_out, r, dt, score[1]);
    if (dv_out) store_T(dv_out, r, dt, dist[0]);
    if (df_out) store_T(df_out, r, dt, dist[1]);
  }
  if (norm_out) for (int64_t r = 0; r < R; ++r) store_T(norm_out, r, dt, nrm[Z(r)]);
  if (vc_out) for (int64_t j = 0; j < C; ++j) store_T(vc_out, j, dt, vc[Z(j)]);
  if (fc_out) for (int64_t i = 0; i < F * C; ++i) store_T(fc_out, i, dt, fc[Z(i)]);
  return 0;
}

// _multi_scale_gaussian(x, center, alphas) as a standalone call (vidcom2.py:59-62).  x: T[R, C]
// (rows of N per frame), centre: T[n_centres, C] with n_centres == 1 (video centre) or R / N (one per
// frame); two_a[i] = fp32(2 * alpha_i).  out: T[R].
int vc2o_multi_scale_gaussian(const void* x, int64_t R, int64_t C, int dt, const void* centre,
                              int64_t n_centres, int64_t N, const float* two_a, int na, void* out) {
  if (!x || !centre || !out || !two_a || R < 0 || C <= 0 || N <= 0 || na <= 0) return -1;
  if (n_centres != 1 && n_centres * N != R) return -2;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const int64_t cb = (n_centres == 1 ? 0 : r / N) * C;
    float dist;
    if (g_mode == 1) {
      std::vector<float> q(Z(C));
      for (int64_t j = 0; j < C; ++j) {
        float a = rT(load_T(x, r * C + j, dt) - load_T(centre, cb + j, dt), dt);
        q[Z(j)] = rT(a * a, dt);
      }
      dist = rT(sum_torch_order(q.data(), C, dt), dt);
    } else {
      double sv = 0.0;
      for (int64_t j = 0; j < C; ++j) {
        float a = rT(load_T(x, r * C + j, dt) - load_T(centre, cb + j, dt), dt);
        sv += double(rT(a * a, dt));
      }
      dist = rTd(sv, dt);
    }
    float acc = 0.0f;
    for (int a = 0; a < na; ++a) {
      float arg = rT((-dist) / two_a[a], dt);
      float e = rTd(std::exp(double(arg)), dt);
      acc = (a == 0) ? e : rT(acc + e, dt);          // Python sum(): 0 + t1 is exact
    }
    store_T(out, r, dt, acc);
  }
  return 0;
}

int vc2o_gaussian_scores(const void* x, int64_t R, int64_t D, int dt, const int64_t* idx,
                         int64_t C, int64_t tpf, void* v_out, void* f_out, void* norm_out,
                         void* vc_out, void* fc_out, void* dv_out, void* df_out) {
  return vc2o_gaussian_scores_ex(x, R, D, dt, idx, C, tpf, nullptr, 0, 0, nullptr, v_out, f_out, norm_out,
                                 vc_out, fc_out, dv_out, df_out);
}

// vidcom2.py:32-33 call-site expressions: s = -vid_score.mean(-1) (T[F]); total = v + f (T[F,N])
int vc2o_fuse(const void* v, const void* f, int64_t F, int64_t N, int dt, void* s_out,
              void* total_out) {
  for (int64_t i = 0; i < F; ++i) {
    double t = 0.0;
    for (int64_t n = 0; n < N; ++n) {
      float a = load_T(v, i * N + n, dt), b = load_T(f, i * N + n, dt);
      t += double(a);
      store_T(total_out, i * N + n, dt, a + b);
    }
    store_T(s_out, i, dt, -mean_T(t, N, dt));
  }
  return 0;
}

// vidcom2.py:64-68  compute_scales(scores, base, temp=0.01) -> T[F]
int vc2o_compute_scales(const void* s, int64_t F, double base, double temp, int dt,
                        void* scales_out) {
  if (F <= 0) return 0;
  std::vector<float> z(Z(F));
  float mx = load_T(s, 0, dt);
  bool any_nan = std::isnan(mx);
  for (int64_t i = 1; i < F; ++i) {
    float v = load_T(s, i, dt);
    if (std::isnan(v)) any_nan = true;
    if (v > mx) mx = v;
  }
  if (any_nan) mx = NAN;
  const float tf = float(temp);                       // python scalar -> opmath (fp32)
  for (int64_t i = 0; i < F; ++i) {
    float d = rT(load_T(s, i, dt) - mx, dt);          // scores - scores.max()
    z[Z(i)] = rT(d / tf, dt);                    // / temp
  }
  // F.softmax(dim=0): fp32 internally: exp(z - max z) / sum, one cast at the end
  float zmax = z[0];
  for (int64_t i = 1; i < F; ++i) if (z[Z(i)] > zmax) zmax = z[Z(i)];
  std::vector<double> e(Z(F));
  double esum = 0.0;
  for (int64_t i = 0; i < F; ++i) {
    e[Z(i)] = double(float(std::exp(double(z[Z(i)] - zmax))));
    esum += e[Z(i)];
  }
  std::vector<float> p(Z(F));
  double psum = 0.0;
  for (int64_t i = 0; i < F; ++i) {
    p[Z(i)] = rTd(e[Z(i)] / esum, dt);
    psum += double(p[Z(i)]);
  }
  const float pmean = mean_T(psum, F, dt);            // probs.mean()
  const float bf = float(base);
  for (int64_t i = 0; i < F; ++i) {
    float t = rT(1.0f + p[Z(i)], dt);            // 1 + probs
    t = rT(t - pmean, dt);                            // - probs.mean()
    t = rT(bf * t, dt);                               // base * (...)
    if (t > 1.0f) t = 1.0f;                           // clamp(max=1.0); NaN propagates
    store_T(scales_out, i, dt, t);
  }
  return 0;
}

// vidcom2.py:72  ks = (scales * tpf).round().long().clamp(min=1)
int vc2o_ks(const void* scales, int64_t F, int64_t tpf, int dt, int64_t* ks_out) {
  for (int64_t i = 0; i < F; ++i) {
    float t = rT(load_T(scales, i, dt) * float(tpf), dt);
    t = std::nearbyintf(t);                           // round-half-even (default FE_TONEAREST)
    int64_t k = std::isnan(t) ? INT64_MIN : int64_t(t);
    ks_out[i] = k < 1 ? 1 : k;
  }
  return 0;
}

// vidcom2.py:74-77  per frame: topk(scores[i], k, largest=False, sorted=False) then index sort.
// idx_out: concatenated ascending local indices (sum ks entries).
int vc2o_select_outliers(const void* total, int64_t F, int64_t N, int dt, const int64_t* ks,
                         int64_t* idx_out) {
  std::vector<int64_t> off(Z(F) + 1, 0);
  for (int64_t i = 0; i < F; ++i) {
    if (ks[i] > N) return -3;                         // torch.topk raises: k out of range
    off[Z(i) + 1] = off[Z(i)] + ks[i];
  }
#pragma omp parallel
  {
    std::vector<elem_t> q;
    std::vector<float> v(Z(N));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < F; ++i) {
      for (int64_t n = 0; n < N; ++n) v[Z(n)] = load_T(total, i * N + n, dt);
      int64_t* o = idx_out + off[Z(i)];
      topk_smallest(v.data(), N, ks[i], false, o, q);
      std::sort(o, o + ks[i]);
    }
  }
  return 0;
}

// vidcom2.py:99-103 / :105-115  index mappers
int vc2o_map_linear(const int64_t* local_idx, const int64_t* ks, int64_t F, int64_t stride,
                    int64_t* out) {
  int64_t p = 0;
  for (int64_t i = 0; i < F; ++i)
    for (int64_t j = 0; j < ks[i]; ++j, ++p) out[p] = local_idx[p] + i * stride;
  return 0;
}
int vc2o_map_grid_vid(const int64_t* local_idx, const int64_t* ks, int64_t F, int64_t h,
                      int64_t* out) {
  const int64_t w_new = h + 1, stride = h * w_new;
  int64_t p = 0, o = 0;
  for (int64_t i = 0; i < F; ++i) {
    const int64_t start = i * stride;
    for (int64_t j = 0; j < ks[i]; ++j, ++p)
      out[o++] = start + (local_idx[p] / h) * w_new + (local_idx[p] % h);
    for (int64_t a = 0; a < h; ++a) out[o++] = start + a * w_new + h;
  }
  return 0;
}

// Whole pass up to the kept global (linear) indices -- vidcom2.py:27-33 + :99-103.
// Outputs: chan_idx int64[C], v/f/total T[F*N], s/scales T[F], ks int64[F],
// gidx int64[>= sum ks] (caller sizes it F*N), K_out.
int vc2o_compress_indices(const void* x, int64_t F, int64_t N, int64_t D, int dt, double base,
                          int64_t* chan_idx, void* v, void* f, void* total, void* s,
                          void* scales, int64_t* ks, int64_t* gidx, int64_t* K_out) {
  const int64_t R = F * N, C = int64_t(double(D) * 0.5);
  int rc = vc2o_select_low_var_channels(x, R, D, dt, C, nullptr, chan_idx);
  if (rc) return rc;
  rc = vc2o_gaussian_scores(x, R, D, dt, chan_idx, C, N, v, f, nullptr, nullptr, nullptr,
                            nullptr, nullptr);
  if (rc) return rc;
  vc2o_fuse(v, f, F, N, dt, s, total);
  vc2o_compute_scales(s, F, base, 0.01, dt, scales);
  vc2o_ks(scales, F, N, dt, ks);
  std::vector<int64_t> local(Z(R));
  rc = vc2o_select_outliers(total, F, N, dt, ks, local.data());
  if (rc) return rc;
  int64_t K = 0;
  for (int64_t i = 0; i < F; ++i) K += ks[i];
  vc2o_map_linear(local.data(), ks, F, N, gidx);
  *K_out = K;
  return 0;
}

// exp exactly as the Gaussian kernel evaluates it (vidcom2.py:62): out = RN_T(RN_f32(exp(x))) -- KAT hook
int vc2o_exp_T(const void* in, int64_t n, int dt, void* out) {
  for (int64_t i = 0; i < n; ++i) store_T(out, i, dt, float(std::exp(double(load_T(in, i, dt)))));
  return 0;
}

// flat[global_idx] row gather (vidcom2.py:91 / :96)
int vc2o_gather_rows(const void* src, const int64_t* idx, int64_t K, int64_t D, int dt,
                     void* dst) {
  const size_t es = (dt == F32) ? 4 : 2;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < K; ++j)
    std::memcpy(static_cast<char*>(dst) + Z(j) * Z(D) * es,
                static_cast<const char*>(src) + Z(idx[j]) * Z(D) * es, Z(D) * es);
  return 0;
}

}  // extern "C"
