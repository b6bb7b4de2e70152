// ssdk_common.h -- shared device/host helpers for the gfx950 kernels of libssdk.so.
// CDNA4 only: 64-lane wavefronts are assumed everywhere (no multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ssdk.h"

namespace ssdk {

using u32 = uint32_t;
using u64 = unsigned long long;
using u16 = uint16_t;

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// ---- host-side error plumbing (ssdk_api.cpp) -------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
void lds_poison(hipStream_t stream);  // ssdk_debug.hip: no-op unless SSDK_LDS_POISON=1

// ---- order-preserving float <-> u32 (total order on non-NaN floats) ---------------------------
__host__ __device__ __forceinline__ u32 ord_f32(float f) {
  u32 b = __builtin_bit_cast(u32, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float unord_f32(u32 u) {
  u32 b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, b);
}
// 64-bit composite key: descending key order == (score descending, index ascending).
__device__ __forceinline__ u64 make_key(float s, u32 idx) {
  return ((u64)ord_f32(s) << 32) | (u64)(~idx);
}
__device__ __forceinline__ float key_score(u64 k) { return unord_f32((u32)(k >> 32)); }
__device__ __forceinline__ u32 key_index(u64 k) { return ~(u32)k; }

// ---- dtype helpers --------------------------------------------------------------------------
template <int DT> struct DType;
template <> struct DType<SSDK_F32> { static constexpr int size = 4; static constexpr int vec = 4; };
template <> struct DType<SSDK_BF16> { static constexpr int size = 2; static constexpr int vec = 8; };
template <> struct DType<SSDK_F16> { static constexpr int size = 2; static constexpr int vec = 8; };

__device__ __forceinline__ float bf16_bits_to_f32(u32 h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ float f16_bits_to_f32(u32 h) {
  _Float16 x = __builtin_bit_cast(_Float16, (u16)h);
  return (float)x;
}

// element e (compile-time) of a 16-byte vector holding DType<DT>::vec elements
template <int DT, int E>
__device__ __forceinline__ float vec_elem(const u32x4& v) {
  if constexpr (DT == SSDK_F32) {
    const u32 w = v[E];  // (bit_cast straight from the vector-element lvalue reads element 0)
    return __builtin_bit_cast(float, w);
  } else {
    u32 w = v[E >> 1];
    u32 h = (E & 1) ? (w >> 16) : (w & 0xffffu);
    if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
    else return f16_bits_to_f32(h);
  }
}

// scalar load of element i of a tensor of dtype dt (runtime), upcast to fp32
__device__ __forceinline__ float load_as_f32(const void* p, size_t i, int dt) {
  if (dt == SSDK_F32) return ((const float*)p)[i];
  u32 h = ((const u16*)p)[i];
  return dt == SSDK_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}

// ---- wave helpers ---------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 mbcnt(u64 m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int d) {
  u32 lo = __shfl_xor((u32)v, d), hi = __shfl_xor((u32)(v >> 32), d);
  return ((u64)hi << 32) | lo;
}

// NaN-propagating min / max (torch.min / torch.max / clamp semantics; fminf/fmaxf would drop the NaN): v_min / v_max
// plus a NaN mask built from integer arithmetic, no compare-and-select.  The subtraction is opaque to the optimiser on
// purpose: LLVM would otherwise turn the sign test back into a compare.  (Round 1 introduced this form on the theory
// that a v_cndmask read a stale VCC under multi-queue load; tools/repro/vcc_select_repro.hip does not reproduce that
// -- 0 wrong lanes of 1.3e9 -- and DESIGN.md 8.4 retracts it.  The form stays because it is branch- and VCC-free.)
__device__ __forceinline__ u32 nan_or_mask(float a, float b) {  // 0x7fc00000 when a or b is NaN, else 0
  const u32 ba = __builtin_bit_cast(u32, a) & 0x7fffffffu, bb = __builtin_bit_cast(u32, b) & 0x7fffffffu;
  u32 ta, tb;
  asm("v_sub_u32 %0, 0x7f800000, %1" : "=v"(ta) : "v"(ba));  // negative iff |bits| > inf, i.e. NaN
  asm("v_sub_u32 %0, 0x7f800000, %1" : "=v"(tb) : "v"(bb));
  return (u32)((int)(ta | tb) >> 31) & 0x7fc00000u;
}
__device__ __forceinline__ float tmin(float a, float b) {
  return __builtin_bit_cast(float, __builtin_bit_cast(u32, __builtin_fminf(a, b)) | nan_or_mask(a, b));
}
__device__ __forceinline__ float tmax(float a, float b) {
  return __builtin_bit_cast(float, __builtin_bit_cast(u32, __builtin_fmaxf(a, b)) | nan_or_mask(a, b));
}

}  // namespace ssdk
