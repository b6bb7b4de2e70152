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

// ReLU6 of N packed-fp16 words in place, as two passes of independent instructions.  Written with __builtin_elementwise_max /
// min the compiler (a) canonicalises every input first (v_pk_max v, v, v: the values went through an asm pin, so it no longer
// knows they are FMA results) and (b) emits the three dependent packed instructions of a word back to back, each pair separated
// by the s_nop a dependent VOP3P pair needs: 5 issue slots per word, 180 per output row of the 144-channel block.  Here: 2.
template <int N>
__device__ __forceinline__ void fl_relu6_words(fl_h2* v) {
  const u32 six = 0x46004600u;
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(v[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(v[i]) : "v"(six));
}

// Stride 2, two strips per wave: only the ODD lanes 1..13 of a strip own an output pixel, so the taps of strip `a` stay in
// the odd lanes and those of strip `b` move one lane to the left into the even lanes 0..12 -- one register set then holds
// the 14 outputs of both strips and every packed FMA, the bias / ReLU6 pass, the projection MFMAs and the epilogue run once
// for the pair instead of once per strip (half of their lanes idle).  Lane j of the merged registers:
//   odd  j: left a[j-1], centre a[j],   right a[j+1]      (strip a, output (j-1)/2)
//   even j: left b[j],   centre b[j+1], right b[j+2]      (strip b, output j/2)
// A select whose one arm is a lane shift is ONE instruction (v_cndmask_b32_dpp: VCC ? src1 : dpp(src0)); the compiler
// does not form it (it branches around a v_mov_b32_dpp instead, which reads disabled lanes), hence the asm block: VCC holds
// the odd-lane mask, then the even-lane mask; s_nop 1 = the two wait states between a VALU write and a DPP read of it.
__device__ __forceinline__ void fl_merge_s2(u32 a0, u32 a1, u32 b0, u32 b1, u32& l0, u32& l1, u32& c0, u32& c1, u32& r0, u32& r1) {
  u32 t0, t1;
  asm volatile(
      "s_nop 1\n\t"
      "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\t"
      "s_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
      "v_cndmask_b32_dpp %2, %10, %8, vcc row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %3, %11, %9, vcc row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %7, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %4, %10, %6, vcc row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %5, %11, %7, vcc row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_mov_b32 vcc_lo, 0x55555555\n\t"
      "s_mov_b32 vcc_hi, 0x55555555\n\t"
      "v_cndmask_b32_dpp %0, %8, %10, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %1, %9, %11, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      : "=&v"(l0), "=&v"(l1), "=&v"(c0), "=&v"(c1), "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc");
}

}  // namespace ssdk
