// vc2_device.h -- device-side helpers shared by the VidCom2 hot-path kernels (gfx950 only).
//
// Numerics helpers implement the "every torch op rounds to the input dtype T" behaviour of the
// reference (token_compressor/vidcom2/vidcom2.py runs entirely in T; SURVEY.md finding 2):
//   rnT<DT>(v)   = RN_T(v) widened back to fp32        (v is the fp32 result of ONE IEEE op)
// fp32 single ops are IEEE (this TU is built with -ffp-contract=off so a*b+c is never fused);
// reductions accumulate in fp64 so that the value handed to rnT is the correctly rounded one.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vc2.h"

namespace vc2 {

constexpr int kWave = 64;

template <int DT> struct Tr;
template <> struct Tr<VC2_F32>  { static constexpr int ES = 4; static constexpr int VEC = 4; };
template <> struct Tr<VC2_BF16> { static constexpr int ES = 2; static constexpr int VEC = 8; };
template <> struct Tr<VC2_F16>  { static constexpr int ES = 2; static constexpr int VEC = 8; };

// ---- RN_T(fp32) widened to fp32 ------------------------------------------------------
template <int DT> __device__ __forceinline__ float rnT(float v);
template <> __device__ __forceinline__ float rnT<VC2_F32>(float v) { return v; }
template <> __device__ __forceinline__ float rnT<VC2_BF16>(float v) {
  return static_cast<float>(static_cast<__bf16>(v));        // v_cvt_pk_bf16_f32 (RNE) + shift
}
template <> __device__ __forceinline__ float rnT<VC2_F16>(float v) {
  return static_cast<float>(static_cast<_Float16>(v));      // v_cvt_f16_f32 (RNE) + v_cvt_f32_f16
}

// RN_T of two values at once (one v_cvt_pk_* instead of two)
template <int DT> __device__ __forceinline__ void rnT2(float a, float b, float& ra, float& rb) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  if constexpr (DT == VC2_F32) { ra = a; rb = b; }
  else if constexpr (DT == VC2_BF16) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const f2_t w = __builtin_convertvector(__builtin_convertvector((f2_t){a, b}, b2_t), f2_t);
    ra = w.x; rb = w.y;
  } else {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const f2_t w = __builtin_convertvector(__builtin_convertvector((f2_t){a, b}, h2_t), f2_t);
    ra = w.x; rb = w.y;
  }
}

// the same on a register pair
typedef float f2_t __attribute__((ext_vector_type(2)));
template <int DT> __device__ __forceinline__ f2_t rnT2v(f2_t v) {
  if constexpr (DT == VC2_F32) {
    return v;
  } else if constexpr (DT == VC2_BF16) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(__builtin_convertvector(v, b2_t), f2_t);
  } else {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(__builtin_convertvector(v, h2_t), f2_t);
  }
}

// (a, b) -> (a*a, b*b) in ONE packed instruction (the compiler scalarises the vector multiply here)
__device__ __forceinline__ f2_t pk_square(f2_t v) {
  f2_t r;
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(r) : "v"(v));
  return r;
}


// ---- scalar T load / store -----------------------------------------------------------
template <int DT> __device__ __forceinline__ float ldT(const void* p, int64_t i) {
  if constexpr (DT == VC2_F32) {
    return static_cast<const float*>(p)[i];
  } else if constexpr (DT == VC2_BF16) {
    return __uint_as_float(static_cast<uint32_t>(static_cast<const uint16_t*>(p)[i]) << 16);
  } else {
    return static_cast<float>(static_cast<const _Float16*>(p)[i]);
  }
}
template <int DT> __device__ __forceinline__ void stT(void* p, int64_t i, float v) {  // rounds RNE
  if constexpr (DT == VC2_F32) {
    static_cast<float*>(p)[i] = v;
  } else if constexpr (DT == VC2_BF16) {
    static_cast<__bf16*>(p)[i] = static_cast<__bf16>(v);
  } else {
    static_cast<_Float16*>(p)[i] = static_cast<_Float16>(v);
  }
}

// ---- 16-byte vector of VEC elements, unpacked to fp32 --------------------------------
// VEC == Tr<DT>::VEC : one 16-byte load per lane (coalesced 1 KiB per wave);
// VEC == 1           : scalar fallback for rows whose byte length is not a multiple of 16.
template <int DT, int VEC> struct RawVec;
template <int DT> struct RawVec<DT, 1> { float v; };
template <> struct RawVec<VC2_F32, 4> { float4 v; };
template <> struct RawVec<VC2_BF16, 8> { uint4 v; };
template <> struct RawVec<VC2_F16, 8> { uint4 v; };

template <int DT, int VEC>
__device__ __forceinline__ RawVec<DT, VEC> load_raw(const void* __restrict__ x, int64_t elem) {
  RawVec<DT, VEC> r;
  if constexpr (VEC == 1) {
    r.v = ldT<DT>(x, elem);
  } else if constexpr (DT == VC2_F32) {
    r.v = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + elem);
  } else {
    r.v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(x) + elem);
  }
  return r;
}
template <int DT, int VEC> __device__ __forceinline__ RawVec<DT, VEC> zero_raw() {
  RawVec<DT, VEC> r;
  if constexpr (VEC == 1) r.v = 0.f;
  else if constexpr (DT == VC2_F32) r.v = make_float4(0.f, 0.f, 0.f, 0.f);
  else r.v = make_uint4(0u, 0u, 0u, 0u);
  return r;
}
template <int DT, int VEC>
__device__ __forceinline__ void unpack(const RawVec<DT, VEC>& r, float (&o)[VEC]) {
  if constexpr (VEC == 1) {
    o[0] = r.v;
  } else if constexpr (DT == VC2_F32) {
    o[0] = r.v.x; o[1] = r.v.y; o[2] = r.v.z; o[3] = r.v.w;
  } else if constexpr (DT == VC2_BF16) {
    const uint32_t w[4] = {r.v.x, r.v.y, r.v.z, r.v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
  } else {
    union { uint4 u; _Float16 h[8]; } c;
    c.u = r.v;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = static_cast<float>(c.h[i]);
  }
}

// ---- wave / block reductions (fp64, fixed order => deterministic) ---------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Wave total of an fp64 value with DPP lane moves (VALU latency) instead of ds_bpermute (LDS-crossbar
// latency): quad swaps, half-row / row mirrors, then the two gfx9 row broadcasts; the total lands in
// lane 63 and is broadcast through an SGPR pair.  Every lane returns the same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return v + __hiloint2double(thi, tlo);
}
__device__ __forceinline__ double wave_sum_bcast(double v) {
  v = dpp_add<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);    // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);    // row_mirror        -> every lane holds its 16-lane row total
  v = dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// The same for fp32 (the "torch order" sweeps accumulate in fp32 with a proven error bound):
// a fixed reduction tree, so every lane returns the same bits and the result is run-to-run identical.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_addf(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_bcast_f32(float v) {
  v = dpp_addf<0xB1, 0xF>(v);
  v = dpp_addf<0x4E, 0xF>(v);
  v = dpp_addf<0x141, 0xF>(v);
  v = dpp_addf<0x140, 0xF>(v);
  v = dpp_addf<0x142, 0xA>(v);
  v = dpp_addf<0x143, 0xC>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t wave_max_bcast_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}

__device__ __forceinline__ float wave_max_nanprop(float v) {  // NaN wins (torch max semantics)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o, 64);
    v = (v != v) ? v : ((t != t) ? t : fmaxf(v, t));
  }
  return v;
}

// the same through DPP lane moves (VALU latency instead of six LDS-crossbar round trips); every lane returns the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_nanprop(float v) {
  const float t = __int_as_float(__builtin_amdgcn_update_dpp(int(0xFF800000u), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
  return (v != v) ? v : ((t != t) ? t : fmaxf(v, t));          // (rows outside ROW_MASK see -inf: the identity)
}
__device__ __forceinline__ float wave_max_nanprop_bcast(float v) {
  v = dpp_max_nanprop<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v = dpp_max_nanprop<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v = dpp_max_nanprop<0x141, 0xF>(v);    // row_half_mirror
  v = dpp_max_nanprop<0x140, 0xF>(v);    // row_mirror        -> every lane holds its 16-lane row maximum
  v = dpp_max_nanprop<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_max_nanprop<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// torch mean (ReduceOps.cpp mean_out): fp32 sum -> fp32 divide by the count -> cast to T.
template <int DT> __device__ __forceinline__ float mean_T(double exact_sum, int64_t count) {
  const float s = static_cast<float>(exact_sum);
  return rnT<DT>(s / static_cast<float>(count));
}

// total-order key for torch.topk(largest=False)'s comparator
//   less(x,y) = (!isnan(x) && isnan(y)) || x < y          (ATen/native/TopKImpl.h)
// so that  less(x,y) <=> key(x) < key(y)  and  "equivalent" <=> equal keys  (-0 == +0, all NaNs equal).
__device__ __forceinline__ uint32_t topk_key(float f) {
  if (f != f) return 0xFFFFFFFFu;
  if (f == 0.f) return 0x80000000u;
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace vc2
