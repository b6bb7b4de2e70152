// ssdk_flow_common.h -- shared by the register-flow inverted-residual kernels (ssdk_mbflow.hip: one wave owns every
// hidden channel of its strip pair; ssdk_mbsplit.hip: the hidden channels of a strip pair are split over the waves of a
// workgroup): parameter block, MFMA / conversion helpers, the DPP neighbour shifts and the stride-2 strip merge.
#pragma once
#include "ssdk_common.h"

namespace ssdk {

typedef __bf16 fl_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fl_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 fl_h2 __attribute__((ext_vector_type(2)));

struct FlowParams {
  const u16* x;
  u16* y;
  const u16* we;
  const float* se;
  const float* be;
  const u16* wd;
  const u16* bd;
  const u16* wp;
  const float* sp;
  const float* bp;
  int N, H, W, Cin, Chid, Cout, Ho, Wo, residual;
  int strips, segs, rs;  // strips per row, row segments per image, output rows per segment
  // STEM instances: x is the image [N][Cimg<=3][Himg][Wimg] (layout 1, NCHW) or [N][Himg][Wimg][Cimg] (2, NHWC); the
  // "expand" GEMM is the 3x3 / stride 2 / pad 1 stem convolution, K = (ci, ky, kx) = 27 of 32; H, W = its output grid
  int Himg, Wimg, Cimg, layout;
  // STEM: row segments that touch the top / bottom of the image run in their own launch (the YE instance masks rows per
  // value; the interior instance has a branch-free gather): bit s of seg_mask = segment s belongs to THIS launch
  unsigned long long seg_mask;
  // SSDK_MB_DBG=1 (debug): shader-clock stamps of one wave in the middle of the grid, four per input row: row entry, x row
  // arrived (an explicit wait), chunk loop done, projection + stores issued
  unsigned long long* dbg;
};

template <int DT>
__device__ __forceinline__ f32x4 fl_mfma(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (DT == SSDK_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fl_bf16x8, a), __builtin_bit_cast(fl_bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fl_f16x8, a), __builtin_bit_cast(fl_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ fl_h2 fl_as_h2(u32 w) { return __builtin_bit_cast(fl_h2, w); }
__device__ __forceinline__ u32 fl_as_u32(fl_h2 v) { return __builtin_bit_cast(u32, v); }
template <int DT> __device__ __forceinline__ u32 fl_to16(float v) {
  if constexpr (DT == SSDK_BF16) {
    u32 b = __builtin_bit_cast(u32, v);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
  } else {
    _Float16 h = (_Float16)v;
    return (u32)__builtin_bit_cast(u16, h);
  }
}
typedef float fl_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fl_bf16x2 __attribute__((ext_vector_type(2)));
template <int DT> __device__ __forceinline__ u32 fl_pack2(float a, float b) {
  const fl_f32x2 v = {a, b};
  if constexpr (DT == SSDK_BF16) return __builtin_bit_cast(u32, __builtin_convertvector(v, fl_bf16x2));
  else return __builtin_bit_cast(u32, __builtin_convertvector(v, fl_h2));
}
template <int DT> __device__ __forceinline__ float fl_from16(u32 h) {
  if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
  else return f16_bits_to_f32(h);
}
// neighbour pixel inside the 16-lane row; the row's first / last lane gets 0 (those lanes are halo pixels)
__device__ __forceinline__ u32 fl_from_left(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }   // row_shr:1
__device__ __forceinline__ u32 fl_from_right(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }  // row_shl:1

// ---- ReLU6 in units of six (round 4) ---------------------------------------------------------------------------------
// relu6(v) = 6 * clamp(v / 6, 0, 1), and clamping to [0, 1] is a free output modifier of the VALU.  The flow kernels keep both
// internal tensors in units of six: E' = clamp(e / 6) comes out of ONE packed fp32 multiply per two accumulators (its factor
// is 1/6, or 0 for a pixel outside the image: the zero padding of the expanded tensor) followed by the packed conversion,
// instead of two v_med3_f32 per pair; the depthwise sum of E' with the bias / 6 is D / 6 and the LAST packed FMA of an output
// row clamps it (instead of a v_pk_max + v_pk_min pass over the finished row); the factor 6 returns in the fp32 scale of the
// projection epilogue.  Taps and weights are untouched, so the only new rounding is that of bias / 6 to fp16.
typedef float fl_f2 __attribute__((ext_vector_type(2)));
constexpr float kFlSixth = 1.0f / 6.0f;
__device__ __forceinline__ u32 fl_unit_pack(float a, float b, fl_f2 k) {
  const fl_f2 v = {a, b};
  fl_f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "v"(k));
  return __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(r[0], r[1]));
}
__device__ __forceinline__ fl_h2 fl_fma_clamp01(fl_h2 a, fl_h2 b, fl_h2 c) {
  fl_h2 r;
  asm("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// a packed pair of fp16 values divided by six (fp32 arithmetic, one rounding)
__device__ __forceinline__ u32 fl_sixth_h2(u32 w) {
  return fl_to16<SSDK_F16>(fl_from16<SSDK_F16>(w & 0xffffu) * kFlSixth) | (fl_to16<SSDK_F16>(fl_from16<SSDK_F16>(w >> 16) * kFlSixth) << 16);
}

}  // namespace ssdk
