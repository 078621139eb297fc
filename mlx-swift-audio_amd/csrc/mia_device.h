// mia_device.h -- device-side helpers shared by the gfx950 kernels (wave64, MFMA fragments, 16-bit types).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define MIA_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;      // 8 x 16-bit payload (bf16 MFMA operand)
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;   // 8 x f16 (f16 MFMA operand)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- 16-bit storage types: bf16 (tag 2) and f16 (tag 1) share every kernel via this trait --------
struct BF16 {
  static constexpr int tag = 2;
  static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);  // RNE, NaN preserving (v_cvt_pk_bf16_f32)
    return *reinterpret_cast<uint16_t*>(&b);
  }
  static __device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  }
};

struct F16 {
  static constexpr int tag = 1;
  static __device__ __forceinline__ float to_f32(uint16_t v) {
    _Float16 h = __builtin_bit_cast(_Float16, v);
    return (float)h;
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    _Float16 h = (_Float16)f;  // RNE
    return __builtin_bit_cast(uint16_t, h);
  }
  static __device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)T::from_f32(lo) | ((uint32_t)T::from_f32(hi) << 16);
}
// bf16: ONE v_cvt_pk_bf16_f32 (RNE, NaN preserving -- the instruction from_f32 uses, so the bits are the same); written as two
// scalar conversions + shift + or, hipcc emits four instructions per pair
template <>
__device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

// ---- wave64 reductions ----------------------------------------------------------------------
// All-reduce over the 64 lanes; every lane gets the result.  The 16-lane rows are reduced with DPP row operations (full-rate VALU
// modifiers: quad swap, quad pair swap, half-row mirror, row mirror), the four row totals are then read as scalars and combined in
// a fixed order.  `__shfl_xor` compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar per butterfly step, six dependent
// ones per reduction -- which showed in the decode chain's row kernels (two to eight reductions each inside ~3 us of work).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {     // v_readlane_b32 on the bits (the builtin's operand is an int)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);    // row_half_mirror
  v += dpp_f32<0x140>(v);    // row_mirror: every lane of a row holds the row's sum
  const float r0 = lane_f32(v, 0), r1 = lane_f32(v, 16), r2 = lane_f32(v, 32), r3 = lane_f32(v, 48);
  return (r0 + r1) + (r2 + r3);
}
// partial butterflies on DPP for the in-row steps of hand-written reductions: xor 1 / xor 2 (quad permutes), the half-row mirror
// (lane i <-> 7 - i of its group of 8: after the two quad steps it completes an 8-lane all-reduce exactly like xor 4) and the
// 8-lane rotation inside a 16-lane row (== xor 8)
__device__ __forceinline__ float dpp_xor1(float v) { return dpp_f32<0xB1>(v); }
__device__ __forceinline__ float dpp_xor2(float v) { return dpp_f32<0x4E>(v); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return dpp_f32<0x141>(v); }
__device__ __forceinline__ float dpp_xor8(float v) { return dpp_f32<0x128>(v); }     // row_ror:8
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  const float r0 = lane_f32(v, 0), r1 = lane_f32(v, 16), r2 = lane_f32(v, 32), r3 = lane_f32(v, 48);
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// exact (erf) GELU, MLXNN GELU() default.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far
// below the 16-bit rounding of every consumer): one v_exp, one v_rcp and a 5-term Horner chain instead of libm's branchy
// erff, which cost ~45 % of the MLP1 GEMM when run 245 M times per layer in its epilogue.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  // v_rcp_f32 (1 ulp): __frcp_rn is the correctly rounded quotient, which hipcc expands into the ten-instruction IEEE division sequence
  // (div_scale x2, rcp, four fma, div_fmas, div_fixup) -- per GELU value, in the fc1 epilogue -- for a result whose last bit the
  // 1.5e-7 approximation error swamps anyway
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// order-preserving float <-> int key (for atomicMax on floats of either sign)
__device__ __forceinline__ int float_to_ordered(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int k) {
  int i = k >= 0 ? k : k ^ 0x7fffffff;
  return __int_as_float(i);
}
