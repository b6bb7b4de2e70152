// ssdk_conv_common.h -- parameter block and device helpers shared by the convolution kernels
// (ssdk_conv.hip: 128-tile / 256-tile implicit GEMM, depthwise, stem; ssdk_conv3x3.hip: halo-tile 3x3).
#pragma once
#include <stdio.h>

#include "ssdk_common.h"

namespace ssdk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { LAYOUT_NCHW = 0, LAYOUT_NHWC = 1 };

struct ConvParams {
  const void* x;
  const void* w;
  const void* w_frag;  // optional fragment-major image of w (ssdk.h: ssdk_weight_frag_bytes)
  const float* scale;
  const float* bias;
  const void* res;  // residual, same layout/dtype as y (NHWC only)
  void* y;
  void* y2;
  int N, Cin, H, W, Cout, k, stride, pad, Ho, Wo;
  int M;           // N*Ho*Wo
  int cin_chunks;  // ceil(Cin/32)
  int KT;          // k*k*cin_chunks
  int act, act2, split;  // channels >= split use act2 and go to y2 (split == Cout: single output)
  int in_layout, out_layout;
  int post;      // activation applied after the residual add (res_mode bit 1), else SSDK_ACT_NONE
  int res_mode;  // bit 0: the residual tensor is half resolution (nearest x2 upsample); bit 1: activation AFTER the add
  // split-K (small-M layers: too few output tiles to fill 256 CUs and a long, latency-bound k-loop):
  // blockIdx.z owns k-tiles [z*kt_per, (z+1)*kt_per); partial accumulators go to fp32 slabs in fragment order,
  // the last workgroup to arrive on a tile (agent-scope release/acquire on a counter) sums them and runs
  // the epilogue.  Counters are zero on entry and reset by the last arriver.
  int ksplits, kt_per;
  float* slabs;
  unsigned* counters;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SSDK_ACT_RELU: return v > 0.f ? v : 0.f;
    case SSDK_ACT_RELU6: return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    case SSDK_ACT_SILU: return v / (1.0f + __expf(-v));
    case SSDK_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    default: return v;
  }
}

template <int DT> __device__ __forceinline__ u32 f32_to_bits16(float v);
template <> __device__ __forceinline__ u32 f32_to_bits16<SSDK_BF16>(float v) {
  u32 b = __builtin_bit_cast(u32, v);
  if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;  // quiet NaN
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;                  // round to nearest even
}
template <> __device__ __forceinline__ u32 f32_to_bits16<SSDK_F16>(float v) {
  _Float16 h = (_Float16)v;
  return (u32)__builtin_bit_cast(u16, h);
}
template <int DT> __device__ __forceinline__ float bits16_to_f32(u32 h) {
  if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
  else return f16_bits_to_f32(h);
}

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (DT == SSDK_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}


// ---- epilogue arithmetic: y = act(acc * scale + bias) -> 16-bit, four accumulator rows at a time -------------
// The activation id can differ per lane (split heads: loc columns are linear, conf columns sigmoid), so it is
// turned into per-lane clamp bounds + a sigmoid/silu selector once per column; the transcendental uses the
// hardware exp2/rcp (1 ulp each, far below the 16-bit output rounding) and the conversion the packed
// v_cvt_pk_bf16_f32 / v_cvt_f16_f32 (round to nearest even, NaN preserved).
struct ActSel {
  float lo, hi;
  int mode;  // 0 linear/clamp, 1 sigmoid, 2 silu
};
__device__ __forceinline__ ActSel act_sel(int act) {
  ActSel a;
  a.lo = -__builtin_inff();
  a.hi = __builtin_inff();
  a.mode = 0;
  if (act == SSDK_ACT_RELU) a.lo = 0.f;
  else if (act == SSDK_ACT_RELU6) {
    a.lo = 0.f;
    a.hi = 6.f;
  } else if (act == SSDK_ACT_SIGMOID) a.mode = 1;
  else if (act == SSDK_ACT_SILU) a.mode = 2;
  return a;
}
__device__ __forceinline__ bool act_is_sig(int act) { return act == SSDK_ACT_SIGMOID || act == SSDK_ACT_SILU; }
__device__ __forceinline__ bool act_is_clamp(int act) { return act == SSDK_ACT_RELU || act == SSDK_ACT_RELU6; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ u32 pack2_16(float a, float b) {
  const f32x2 v = {a, b};
  if constexpr (DT == SSDK_BF16) return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2));
  else return __builtin_bit_cast(u32, __builtin_convertvector(v, f16x2));
}
// any_sig / any_clamp are workgroup-uniform (does ANY column of this launch use a sigmoid-type / clamp-type act)
template <int DT>
__device__ __forceinline__ uint2 epilogue4(const f32x4 acc, float sc, float bi, const ActSel a, bool any_sig,
                                           bool any_clamp) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r] * sc + bi;
  if (any_sig) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
      v[r] = a.mode == 1 ? sg : (a.mode == 2 ? v[r] * sg : v[r]);
    }
  }
  if (any_clamp) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], a.lo), a.hi);
  }
  return make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
}

// ---- residual add at the NHWC store stage (values already rounded to 16 bit, like the framework's tensor add) ----
// element offset of channel 0 of the residual pixel that belongs to output pixel m = (b, oy, ox)
__device__ __forceinline__ size_t res_pixel_offset(const ConvParams& p, u32 b, u32 oy, u32 ox) {
  if (p.res_mode & 1) return (((size_t)b * (u32)(p.Ho >> 1) + (oy >> 1)) * (u32)(p.Wo >> 1) + (ox >> 1)) * (u32)p.Cout;
  return (((size_t)b * (u32)p.Ho + oy) * (u32)p.Wo + ox) * (u32)p.Cout;
}
__device__ __forceinline__ size_t res_pixel_offset(const ConvParams& p, u32 m) {
  if (!(p.res_mode & 1)) return (size_t)m * (u32)p.Cout;
  const u32 hw = (u32)(p.Ho * p.Wo);
  const u32 b = m / hw, r = m % hw;
  return res_pixel_offset(p, b, r / (u32)p.Wo, r % (u32)p.Wo);
}
__device__ __forceinline__ float post_act(float v, int act) {  // res_mode bit 1: relu / relu6 after the add
  if (act == SSDK_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SSDK_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
  return v;
}
template <int DT>
__device__ __forceinline__ u32x4 add_residual8(u32x4 v, const u16* res, int post) {  // 8 channels, 16-byte aligned
  const u32x4 rv = *reinterpret_cast<const u32x4*>(res);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 a = v[e], r = rv[e];
    const float lo = post_act(bits16_to_f32<DT>(a & 0xffffu) + bits16_to_f32<DT>(r & 0xffffu), post);
    const float hi = post_act(bits16_to_f32<DT>(a >> 16) + bits16_to_f32<DT>(r >> 16), post);
    v[e] = pack2_16<DT>(lo, hi);
  }
  return v;
}
template <int DT>
__device__ __forceinline__ u32 add_residual1(u32 v, const u16* res, int post) {
  return f32_to_bits16<DT>(post_act(bits16_to_f32<DT>(v) + bits16_to_f32<DT>((u32)*res), post));
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) const unsigned char glb_u8;

// 16-byte zero page: masked chunks of a direct-to-LDS load read it (a masked-off lane would leave stale LDS)
static __device__ __attribute__((aligned(16))) unsigned g_zero16[4];  // one copy per translation unit

// async global -> LDS copy of 16 bytes per lane; the LDS destination is wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((glb_u8*)g, (lds_u8*)l, 16, 0, 0);
}

// grouped 3x3 convolution, 16 channels per group (ssdk_gconv.hip)
int launch_gconv3x3_g16(const ssdk_conv_desc* d, int Ho, int Wo, hipStream_t stream);

// halo-tile 3x3 kernel with split-K over its channel slabs: slices to use (1: no split) and the workspace they need
int halo_splitk_plan(int N, int Cin, int Ho, int Wo, int Cout, bool nchw, size_t* ws_bytes);
// halo-tile 3x3 kernel (ssdk_conv3x3.hip); returns SSDK_OK, or 1 when the layer does not fit it
size_t halo_ws_bytes(int N, int Cin, int Ho, int Wo, int Cout, bool nchw, int ksplits);
int launch_conv3x3_halo(const ConvParams& p, int dtype, hipStream_t stream, bool allow_underfill);
int launch_conv3x3_short(const ConvParams& p, int dtype, hipStream_t stream);
int launch_conv_pwflow(const ConvParams& p, int dtype, hipStream_t stream);  // ssdk_pwflow.hip: 1x1, Cin <= 256, streaming
// 1x1, long K (Cin >= 256), NHWC: persistent NT GEMM (ssdk_gemmp.hip); 1: not taken
int launch_conv_gemmp(const ConvParams& p, int dtype, hipStream_t stream);
constexpr int kSmallmapGroupMax = 4;
int launch_conv_smallmap_group(const ConvParams* ps, int n, int dtype, hipStream_t stream);  // ssdk_smallmap.hip  // ssdk_conv3x3s.hip: Cin <= 128, needs p.w_frag

}  // namespace ssdk
