// ssdk_bntrain.hip -- BatchNorm2d training forward / backward (batch statistics), NCHW, fp32 | bf16 | f16 data with
// fp32 parameters and statistics, on gfx950.
//
// Why: after the depthwise convolutions, MIOpenBatchNorm{Fwd,Bwd}Spatial are the largest item of the reference's DDP
// training step on SSD-MobileNetV2 (22 of 57 ms at batch 64; they move ~28 GB where ~7 ms of HBM time would do).
// Both directions are two HBM-bound passes:
//   forward   (1) per-channel sum / sum of squares about a pivot (the channel's first element: avoids the
//                 cancellation of E[x^2] - E[x]^2), partials per (channel, image-slice) in a fixed order
//             (2) y = x * a[c] + b[c],  a = gamma * invstd, b = beta - mean * a;  running statistics updated like
//                 torch (momentum, unbiased variance)
//   backward  (1) per-channel sum(dy), sum(dy * xhat)            -> dbeta, dgamma
//             (2) dx = a * dy + x * k1[c] + k0[c]   (the usual formula with the per-channel scalars folded)
// An activation that follows the BatchNorm (ReLU6 / ReLU: every Conv-BN-ReLU6 of MobileNetV2) can ride along
// (`act`): the forward apply clamps before it stores, and both backward passes mask dy where the rounded
// pre-activation x*a+b left the open interval -- recomputed, nothing extra is saved.  This removes the clamp and
// hardtanh_backward launches and one read + one write of every activation tensor in each direction.
// A workgroup of a reduction pass owns (channel c, slice s): the planes n = s, s + SPLIT, ... of that channel, read
// with 16-byte vectors; partials are combined by one thread per channel in index order (bit-reproducible).
#include <atomic>

#include "ssdk_conv_common.h"

namespace ssdk {

struct BnParams {
  const void* x;
  const void* dy;
  void* out;               // y | dx
  const float* weight;
  const float* bias;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* dweight;
  float* dbias;
  int no_apply;            // forward: statistics + coefficients only (the consumer applies them on load: ssdk_dwconv_fwd_affine)
  const float* sums;       // optional: [C][2] = (sum x, sum x^2) over N * HW computed by the PRODUCER of x (ssdk_pw_forward_stats):
                           // the forward pass then has no reduction of its own
  float* partial;          // [C][SPLIT][2]
  float* coef;             // [C][4] per-channel scalars of the apply pass
  int N, C, HW, split, dtype;
  float momentum, eps;
  int act;                 // 0 none | 1 ReLU6 | 2 ReLU fused behind the normalisation
  u32* tickets;            // [C] arrival counters of the reduction's workgroups (zero between launches): the LAST workgroup of a
                           // channel does the finalize arithmetic itself and there is no finalize launch; nullptr: the launch
};

// the activation's pass-through mask on the pre-activation value as the forward pass stored it (rounded to DT)
template <int DT> __device__ __forceinline__ bool bn_act_open(float pre, int act) {
  float v = pre;
  if constexpr (DT != SSDK_F32) v = bits16_to_f32<DT>(f32_to_bits16<DT>(pre));
  return act == 1 ? (v > 0.f && v < 6.f) : v > 0.f;   // hardtanh_backward / threshold_backward: open interval
}

template <int DT> struct BnVec { static constexpr int n = DT == SSDK_F32 ? 4 : 8; };

template <int DT>
__device__ __forceinline__ void bn_load(const void* src, size_t i, float (&v)[8]) {
  if constexpr (DT == SSDK_F32) {
    const f32x4 q = *reinterpret_cast<const f32x4*>((const float*)src + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = q[e];
  } else {
    const u32x4 q = *reinterpret_cast<const u32x4*>((const u16*)src + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bits16_to_f32<DT>(q[e] & 0xffffu);
      v[2 * e + 1] = bits16_to_f32<DT>(q[e] >> 16);
    }
  }
}
template <int DT> __device__ __forceinline__ float bn_ld1(const void* p, size_t i) {
  if constexpr (DT == SSDK_F32) return ((const float*)p)[i];
  else return bits16_to_f32<DT>(((const u16*)p)[i]);
}

__device__ __forceinline__ void block_sum2(float& a, float& b, float (*red)[2]) {  // fixed butterfly + wave order
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    a += __shfl_xor(a, d);
    b += __shfl_xor(b, d);
  }
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave][0] = a;
    red[wave][1] = b;
  }
  __syncthreads();
  a = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
  b = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
}

template <int DT, int MODE> __device__ __forceinline__ void bn_fold_finalize(const BnParams& p, int c, int s, float a, float b);  // (below the finalize kernels)

// MODE 0: sums of (x - pivot), (x - pivot)^2.  MODE 1: sums of dy, dy * (x - mean) * invstd.
template <int DT, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const BnParams p) {
  __shared__ float red[4][2];
  constexpr int VN = BnVec<DT>::n;
  const int c = blockIdx.x, s = blockIdx.y;
  const bool vec = (p.HW % VN) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.dy)) & 15u) == 0;
  const float k = MODE == 0 ? bn_ld1<DT>(p.x, (size_t)c * p.HW) : p.save_mean[c];
  const float istd = MODE == 0 ? 1.f : p.save_invstd[c];
  const bool masked = MODE == 1 && p.act != 0;
  const float fa = masked ? (p.weight ? p.weight[c] : 1.f) * istd : 0.f;                 // forward y = x*fa + fb
  const float fb = masked ? (p.bias ? p.bias[c] : 0.f) - p.save_mean[c] * fa : 0.f;
  float a = 0.f, b = 0.f;
  for (int n = s; n < p.N; n += p.split) {
    const size_t base = ((size_t)n * p.C + c) * p.HW;
    if (vec) {
      for (int i = threadIdx.x * VN; i < p.HW; i += 256 * VN) {
        float xv[8], gv[8];
        bn_load<DT>(p.x, base + i, xv);
        if (MODE == 1) bn_load<DT>(p.dy, base + i, gv);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          const float d = xv[e] - k;
          if (MODE == 0) {
            a += d;
            b += d * d;
          } else {
            const float g = (masked && !bn_act_open<DT>(xv[e] * fa + fb, p.act)) ? 0.f : gv[e];
            a += g;
            b += g * d * istd;
          }
        }
      }
    } else {
      for (int i = threadIdx.x; i < p.HW; i += 256) {
        const float xs = bn_ld1<DT>(p.x, base + i);
        const float d = xs - k;
        if (MODE == 0) {
          a += d;
          b += d * d;
        } else {
          float g = bn_ld1<DT>(p.dy, base + i);
          // the mask is evaluated on x itself like the vector branch, the flat kernels and the forward clamp
          // ((x - k) + k is not always x in floating point)
          if (masked && !bn_act_open<DT>(xs * fa + fb, p.act)) g = 0.f;
          a += g;
          b += g * d * istd;
        }
      }
    }
  }
  block_sum2(a, b, red);
  if (p.tickets) {  // (workgroup-uniform)
    bn_fold_finalize<DT, MODE>(p, c, s, a, b);
  } else if (threadIdx.x == 0) {
    p.partial[((size_t)c * p.split + s) * 2 + 0] = a;
    p.partial[((size_t)c * p.split + s) * 2 + 1] = b;
  }
}

// forward statistics -> mean, invstd, running stats, apply coefficients (a, b) of channel c from its sums (s1, s2)
template <int DT>
__device__ __forceinline__ void bn_fwd_finish(const BnParams& p, int c, float s1, float s2) {
  const float M = (float)p.N * (float)p.HW;
  const float pivot = p.sums ? 0.f : bn_ld1<DT>(p.x, (size_t)c * p.HW);
  const float m1 = s1 / M;
  const float mean = pivot + m1;
  float var = s2 / M - m1 * m1;  // biased
  var = var < 0.f ? 0.f : var;
  const float invstd = 1.0f / sqrtf(var + p.eps);
  p.save_mean[c] = mean;
  p.save_invstd[c] = invstd;
  if (p.running_mean) {
    const float unbiased = M > 1.f ? var * (M / (M - 1.f)) : var;
    p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
    p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * unbiased;
  }
  const float g = p.weight ? p.weight[c] : 1.f, bt = p.bias ? p.bias[c] : 0.f;
  const float a = g * invstd;   // (the backward passes rebuild exactly this a and b for the activation mask)
  p.coef[c * 4 + 0] = a;
  p.coef[c * 4 + 1] = bt - mean * a;
  p.coef[c * 4 + 2] = 0.f;
}
template <int DT>
__global__ __launch_bounds__(64) void bn_fwd_finalize_kernel(const BnParams p) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= p.C) return;
  float s1 = 0.f, s2 = 0.f;
  if (p.sums) {  // raw sums from the producing convolution: pivot 0
    s1 = p.sums[2 * c + 0];
    s2 = p.sums[2 * c + 1];
  } else {
    for (int s = 0; s < p.split; ++s) {
      s1 += p.partial[((size_t)c * p.split + s) * 2 + 0];
      s2 += p.partial[((size_t)c * p.split + s) * 2 + 1];
    }
  }
  bn_fwd_finish<DT>(p, c, s1, s2);
}

// backward sums -> dgamma, dbeta, coefficients of dx = a*dy + k1*x + k0
__device__ __forceinline__ void bn_bwd_finish(const BnParams& p, int c, float sg, float sgx) {
  if (p.dweight) p.dweight[c] = sgx;
  if (p.dbias) p.dbias[c] = sg;
  const float M = (float)p.N * (float)p.HW;
  const float g = p.weight ? p.weight[c] : 1.f;
  const float invstd = p.save_invstd[c], mean = p.save_mean[c];
  const float a = g * invstd;
  const float k1 = -a * invstd * sgx / M;
  p.coef[c * 4 + 0] = a;
  p.coef[c * 4 + 1] = -a * sg / M - k1 * mean;  // k0
  p.coef[c * 4 + 2] = k1;
  p.coef[c * 4 + 3] = (p.bias ? p.bias[c] : 0.f) - mean * a;  // forward offset (activation mask of the apply pass)
}
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const BnParams p) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= p.C) return;
  float sg = 0.f, sgx = 0.f;
  for (int s = 0; s < p.split; ++s) {
    sg += p.partial[((size_t)c * p.split + s) * 2 + 0];
    sgx += p.partial[((size_t)c * p.split + s) * 2 + 1];
  }
  bn_bwd_finish(p, c, sg, sgx);
}

// The finalize arithmetic WITHOUT its launch (round 6: 90 finalize launches of ~5 us per training step).  Every workgroup of
// a reduction publishes its partial sums (a, b) and draws a ticket of its channel; the one that draws the last ticket reads
// the channel's partials back, adds them in INDEX order -- the order of the finalize kernels, so the statistics keep their
// bits -- and does what the finalize kernel's thread of that channel does.  It leaves the ticket at zero for the next launch.
// The partials travel between CUs of different XCDs (one L2 each) as agent-scope relaxed atomics: the store goes through to
// memory, the load does not take a line from the reader's L2.  (NOT fences: an agent-scope release is a write-back of the
// whole L2, full of the neighbouring passes' dirty lines -- measured, 17.6 -> 22-29 ms per step.)  split <= 64 (bn_split).
template <int DT, int MODE>
__device__ __forceinline__ void bn_fold_finalize(const BnParams& p, int c, int s, float a, float b) {
  __shared__ u32 s_last;
  __shared__ float s_part[64][2];
  const u32 tid = threadIdx.x;
  if (tid == 0) {
    float* q = p.partial + ((size_t)c * p.split + s) * 2;
    __hip_atomic_store(q + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both stores have been acknowledged before the ticket is drawn
    s_last = __hip_atomic_fetch_add(p.tickets + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u32)p.split - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_last == 0u) return;
  if (tid < (u32)p.split) {
    const float* q = p.partial + ((size_t)c * p.split + tid) * 2;
    s_part[tid][0] = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_part[tid][1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(p.tickets + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < p.split; ++j) {
      s1 += s_part[j][0];
      s2 += s_part[j][1];
    }
    if constexpr (MODE == 0) bn_fwd_finish<DT>(p, c, s1, s2);
    else bn_bwd_finish(p, c, s1, s2);
  }
}

// MODE 0: out = x*a + b.  MODE 1: out = dy*a + x*k1 + k0.   grid (chunks of a plane, N*C planes)
template <int DT, int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnParams p) {
  constexpr int VN = BnVec<DT>::n;
  const int plane = blockIdx.y;
  const int c = plane % p.C;
  const float a = p.coef[c * 4 + 0], k0 = p.coef[c * 4 + 1], k1 = p.coef[c * 4 + 2], fb = p.coef[c * 4 + 3];
  const int act = p.act;
  auto fwd = [&](float x) {  // MODE 0
    float o = x * a + k0;
    if (act) o = fmaxf(o, 0.f);   // (a NaN input stays NaN in torch's clamp; fmaxf drops it -- BatchNorm of a NaN
    if (act == 1) o = fminf(o, 6.f);  //  batch has NaN statistics anyway, every output is NaN before it gets here)
    return o;
  };
  auto bwd = [&](float x, float g) {  // MODE 1: the forward scale is the same `a`
    if (act && !bn_act_open<DT>(x * a + fb, act)) g = 0.f;
    return g * a + x * k1 + k0;
  };
  const size_t base = (size_t)plane * p.HW;
  const bool vec = (p.HW % VN) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.dy) | ((uintptr_t)p.out)) & 15u) == 0;
  if (vec) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * VN;
    if (i >= p.HW) return;
    float xv[8], gv[8], o[8];
    bn_load<DT>(p.x, base + i, xv);
    if (MODE == 1) bn_load<DT>(p.dy, base + i, gv);
#pragma unroll
    for (int e = 0; e < VN; ++e) o[e] = MODE == 0 ? fwd(xv[e]) : bwd(xv[e], gv[e]);
    if constexpr (DT == SSDK_F32) {
      *reinterpret_cast<f32x4*>((float*)p.out + base + i) = f32x4{o[0], o[1], o[2], o[3]};
    } else {
      *reinterpret_cast<u32x4*>((u16*)p.out + base + i) =
          u32x4{pack2_16<DT>(o[0], o[1]), pack2_16<DT>(o[2], o[3]), pack2_16<DT>(o[4], o[5]), pack2_16<DT>(o[6], o[7])};
    }
  } else {
    for (int e = 0; e < VN; ++e) {
      const int i = (blockIdx.x * 256 + threadIdx.x) * VN + e;
      if (i >= p.HW) return;
      const float xv = bn_ld1<DT>(p.x, base + i);
      const float o = MODE == 0 ? fwd(xv) : bwd(xv, bn_ld1<DT>(p.dy, base + i));
      if constexpr (DT == SSDK_F32) ((float*)p.out)[base + i] = o;
      else ((u16*)p.out)[base + i] = (u16)f32_to_bits16<DT>(o);
    }
  }
}

// ---- round 2: the same two passes over FLAT element ranges ---------------------------------------------------------------
// The kernels above keep one 16-byte load in flight per thread (the reduction's loop) or handle one vector per thread and
// 256 * 8 elements per workgroup (the apply pass), use vectors only when the plane size is a multiple of the vector width
// (never at 300 px: 150^2 ... 10^2), and leave most of a workgroup idle on small planes (16 x 16: 32 of 256 threads).
// Measured: ~4 TB/s on the large planes, far less on the small ones; BatchNorm was 7.2 of the 22 ms of kernel time of the
// training step once the depthwise kernels were rewritten (profiles/r02_train_kernel_split_v2.txt).
// Here every access is a 16-byte vector at the element type's own alignment (unaligned access mode, as in
// ssdk_dwplane.hip), four per thread in flight, over index spaces that do not care about plane boundaries:
//   reduce: workgroup (c, s) runs over the vectors of ALL its planes n = s, s + split, ... as one list (+ the < 8 tail
//           elements of each plane one by one);
//   apply:  a workgroup owns 8192 consecutive elements -- a chunk of one large plane, or several whole small planes, whose
//           channel is recovered per vector from the element's offset (a vector that straddles two planes takes its
//           coefficients element by element).
typedef u32x4 bn_u32x4_a2 __attribute__((aligned(2)));
typedef u32x4 bn_u32x4_a4 __attribute__((aligned(4)));
constexpr int kBnU = 4;            // vectors per thread in flight
constexpr int kBnChunk = 8192;     // elements per workgroup of the apply pass

template <int DT> __device__ __forceinline__ u32x4 bn_load_raw(const void* src, size_t i) {
  if constexpr (DT == SSDK_F32) return *reinterpret_cast<const bn_u32x4_a4*>((const u32*)src + i);
  else return *reinterpret_cast<const bn_u32x4_a2*>((const u16*)src + i);
}
template <int DT> __device__ __forceinline__ void bn_unpack(const u32x4 q, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 t = q[e];  // (a scalar first: __builtin_bit_cast of a vector element reads element 0)
    if constexpr (DT == SSDK_F32) {
      v[e] = __builtin_bit_cast(float, t);
    } else {
      v[2 * e] = bits16_to_f32<DT>(t & 0xffffu);
      v[2 * e + 1] = bits16_to_f32<DT>(t >> 16);
    }
  }
}
template <int DT> __device__ __forceinline__ void bn_store_raw(void* dst, size_t i, const float (&o)[8]) {
  if constexpr (DT == SSDK_F32) {
    const u32x4 q = {__builtin_bit_cast(u32, o[0]), __builtin_bit_cast(u32, o[1]), __builtin_bit_cast(u32, o[2]), __builtin_bit_cast(u32, o[3])};
    *reinterpret_cast<bn_u32x4_a4*>((u32*)dst + i) = q;
  } else {
    const u32x4 q = {pack2_16<DT>(o[0], o[1]), pack2_16<DT>(o[2], o[3]), pack2_16<DT>(o[4], o[5]), pack2_16<DT>(o[6], o[7])};
    *reinterpret_cast<bn_u32x4_a2*>((u16*)dst + i) = q;
  }
}
template <int DT> __device__ __forceinline__ void bn_st1(void* p, size_t i, float v) {
  if constexpr (DT == SSDK_F32) ((float*)p)[i] = v;
  else ((u16*)p)[i] = (u16)f32_to_bits16<DT>(v);
}
// v / d for 0 <= v < 2^20, rcp = 1.0f / d (exact: the +0.5 keeps the product half a step away from every integer)
__device__ __forceinline__ int bn_div_small(int v, float rcp) { return (int)(((float)v + 0.5f) * rcp); }

template <int DT, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_flat_kernel(const BnParams p) {
  __shared__ float red[4][2];
  constexpr int VN = BnVec<DT>::n;
  const int c = blockIdx.x, s = blockIdx.y;
  const float k = MODE == 0 ? bn_ld1<DT>(p.x, (size_t)c * p.HW) : p.save_mean[c];
  const float istd = MODE == 0 ? 1.f : p.save_invstd[c];
  const bool masked = MODE == 1 && p.act != 0;
  const float fa = masked ? (p.weight ? p.weight[c] : 1.f) * istd : 0.f;  // forward y = x*fa + fb
  const float fb = masked ? (p.bias ? p.bias[c] : 0.f) - p.save_mean[c] * fa : 0.f;
  const int act = p.act;
  float a = 0.f, b = 0.f;
  auto acc = [&](float x, float g) {
    const float d = x - k;
    if (MODE == 0) {
      a += d;
      b += d * d;
    } else {
      if (masked && !bn_act_open<DT>(x * fa + fb, act)) g = 0.f;
      a += g;
      b += g * d * istd;
    }
  };
  const u32 np = (u32)((p.N - s + p.split - 1) / p.split);  // planes of this workgroup: n = s + j * split
  const u32 vpp = (u32)p.HW / VN, tail = (u32)p.HW - vpp * VN;
  const u32 W = np * vpp;
  const size_t nstride = (size_t)p.split * p.C * p.HW, first = ((size_t)s * p.C + c) * p.HW;
  for (u32 v0 = threadIdx.x; v0 < W; v0 += 256 * kBnU) {
    u32x4 xr[kBnU], gr[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const u32 v = v0 + u * 256;
      xr[u] = u32x4{0u, 0u, 0u, 0u};
      gr[u] = u32x4{0u, 0u, 0u, 0u};
      if (v < W) {
        const u32 j = v / vpp;
        const size_t off = first + j * nstride + (size_t)(v - j * vpp) * VN;
        xr[u] = bn_load_raw<DT>(p.x, off);
        if (MODE == 1) gr[u] = bn_load_raw<DT>(p.dy, off);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // all the loads of the trip in flight before the first one is waited for
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      if (v0 + u * 256 >= W) continue;
      float xv[8], gv[8];
      bn_unpack<DT>(xr[u], xv);
      bn_unpack<DT>(gr[u], gv);
#pragma unroll
      for (int e = 0; e < VN; ++e) acc(xv[e], gv[e]);
    }
  }
  if (tail) {
    for (u32 t = threadIdx.x; t < np * tail; t += 256) {
      const u32 j = t / tail;
      const size_t off = first + j * nstride + (size_t)vpp * VN + (t - j * tail);
      acc(bn_ld1<DT>(p.x, off), MODE == 1 ? bn_ld1<DT>(p.dy, off) : 0.f);
    }
  }
  block_sum2(a, b, red);
  if (p.tickets) {  // (workgroup-uniform)
    bn_fold_finalize<DT, MODE>(p, c, s, a, b);
  } else if (threadIdx.x == 0) {
    p.partial[((size_t)c * p.split + s) * 2 + 0] = a;
    p.partial[((size_t)c * p.split + s) * 2 + 1] = b;
  }
}

struct BnFlat {
  int G, chunks;       // whole planes per workgroup (chunks == 1) | chunks of kBnChunk elements per plane (G == 1)
  float rcpHW, rcpC;
};

// MODE 0: out = x*a + b (+ activation).  MODE 1: out = dy*a + x*k1 + k0 (dy masked by the activation).
template <int DT, int MODE>
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const BnParams p, const BnFlat f) {
  constexpr int VN = BnVec<DT>::n;
  const int act = p.act;
  const long NC = (long)p.N * p.C;
  const u32 bid = blockIdx.x;
  const long P0 = (long)(bid / (u32)f.chunks) * f.G;                     // first plane of this workgroup
  const int lo = (int)(bid % (u32)f.chunks) * kBnChunk;                   // element range [lo, hi) from the start of plane P0
  const long left = NC - P0;
  const int hi = f.chunks > 1 ? (p.HW - lo < kBnChunk ? p.HW : lo + kBnChunk) : (int)(left < f.G ? left : f.G) * p.HW;
  const int c0 = (int)(P0 % p.C);
  const size_t base = (size_t)P0 * p.HW;
  auto channel = [&](int le) {  // channel of the element at offset le from the start of plane P0
    if (f.chunks > 1) return c0;
    const int cc = c0 + bn_div_small(le, f.rcpHW);
    return cc - bn_div_small(cc, f.rcpC) * p.C;
  };
  auto one = [&](float x, float g, const float4 k) {  // k = (a, k0, k1, fb)
    if (MODE == 0) {
      float o = x * k.x + k.y;
      if (act) o = fmaxf(o, 0.f);
      if (act == 1) o = fminf(o, 6.f);
      return o;
    }
    if (act && !bn_act_open<DT>(x * k.x + k.w, act)) g = 0.f;
    return g * k.x + x * k.z + k.y;
  };
  const float4* coef = reinterpret_cast<const float4*>(p.coef);
  const int nvec = (hi - lo) / VN;
  for (int vb = 0; vb < nvec; vb += 256 * kBnU) {
    u32x4 xr[kBnU], gr[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int v = vb + u * 256 + (int)threadIdx.x;
      xr[u] = u32x4{0u, 0u, 0u, 0u};
      gr[u] = u32x4{0u, 0u, 0u, 0u};
      if (v < nvec) {
        const size_t off = base + lo + (size_t)v * VN;
        xr[u] = bn_load_raw<DT>(p.x, off);
        if (MODE == 1) gr[u] = bn_load_raw<DT>(p.dy, off);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
      const int v = vb + u * 256 + (int)threadIdx.x;
      if (v >= nvec) continue;
      const int le = lo + v * VN;
      float xv[8], gv[8], o[8];
      bn_unpack<DT>(xr[u], xv);
      bn_unpack<DT>(gr[u], gv);
      const int pl = f.chunks > 1 ? 0 : bn_div_small(le, f.rcpHW);
      if (f.chunks > 1 || le + VN <= (pl + 1) * p.HW) {  // one plane: one set of coefficients
        const float4 k = coef[channel(le)];
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = one(xv[e], gv[e], k);
      } else {  // the vector straddles planes (plane sizes that are not a multiple of the vector width)
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = one(xv[e], gv[e], coef[channel(le + e)]);
      }
      bn_store_raw<DT>(p.out, base + le, o);
    }
  }
  const int rest = (hi - lo) - nvec * VN;  // < VN elements at the end of the range
  if ((int)threadIdx.x < rest) {
    const int le = lo + nvec * VN + (int)threadIdx.x;
    const float x = bn_ld1<DT>(p.x, base + le), g = MODE == 1 ? bn_ld1<DT>(p.dy, base + le) : 0.f;
    bn_st1<DT>(p.out, base + le, one(x, g, coef[channel(le)]));
  }
}

// (A second version of the flat apply pass -- coefficients through scalar loads where a workgroup's range lies inside one
//  plane, fetched with the data otherwise -- was written at the end of round 2, run in round 3 (57 BatchNorm tests green) and
//  A/B-timed on the 512 px training step: 24.66 / 24.52 ms without, 24.77 / 24.58 ms with it.  No gain: deleted.)

static int bn_split(int N, int C) {
  int s = (1024 + C - 1) / C;
  if (s > N) s = N;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

// Ticket words of the folded finalize: kBnTicketRegions regions of kBnTicketC channels in device memory, zero when the module
// is loaded and left at zero by every launch that uses them.  Consecutive launches take consecutive regions, so BatchNorms
// that run concurrently on two streams do not share counters unless more than kBnTicketRegions of them are in flight.
constexpr int kBnTicketC = 2048, kBnTicketRegions = 16;
__device__ u32 g_bn_tickets[kBnTicketRegions * kBnTicketC];

static u32* bn_tickets(int C) {
  if (C > kBnTicketC) return nullptr;  // (A/B on the 512 px training step, tools/run/r06_s34.sh: 17.43 / 17.44 ms folded, 17.56 / 17.62 ms with the launch)
  static u32* base[16] = {};  // per device (the symbol has one address per device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!base[dev]) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_bn_tickets)) != hipSuccess || !a) return nullptr;
    base[dev] = (u32*)a;
  }
  static std::atomic<unsigned> turn{0};
  return base[dev] + (size_t)(turn.fetch_add(1u) % (unsigned)kBnTicketRegions) * kBnTicketC;
}

template <int MODE>
static void bn_launch(const BnParams& p0, hipStream_t st) {
  BnParams p = p0;
  const bool has_reduce = !(MODE == 0 && p.sums);
  p.tickets = has_reduce ? bn_tickets(p.C) : nullptr;  // the reduction finalizes (no finalize launch)
  // 0: the per-plane kernels, 1: the flat kernels, 2 (default): the flat reduction always, the flat apply pass only where
  // the per-plane one cannot use vectors (plane size not a multiple of the vector width, or a misaligned tensor: there it
  // works element by element).  Per launch on the 512 px step, bf16 (profiles/r02_train_kernel_split_v2.txt against a trace
  // with SSDK_BN_FLAT=1): reduction 21.2 -> 18.1 us forward, 38.8 -> 36.6 us backward; apply 26.3 -> 25.8 us forward but
  // 38.0 -> 45.0 us backward (eight vectors in flight per thread, 84 VGPRs) -- hence the split.
  static const int env_flat = getenv("SSDK_BN_FLAT") ? atoi(getenv("SSDK_BN_FLAT")) : 2;
  const dim3 rgrid((unsigned)p.C, (unsigned)p.split);
  const int vn = p.dtype == SSDK_F32 ? 4 : 8;
  const bool vec_ok = (p.HW % vn) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.dy) | ((uintptr_t)p.out)) & 15u) == 0;
  const bool fits = (long)p.N * p.HW < (1l << 30) && p.HW < (1 << 20) && p.C < (1 << 19);
  const bool flat_r = env_flat != 0 && fits;
  const bool flat = (env_flat == 1 || (env_flat == 2 && !vec_ok)) && fits;
  BnFlat f;
  f.G = p.HW >= kBnChunk ? 1 : kBnChunk / p.HW;
  f.chunks = p.HW >= kBnChunk ? (p.HW + kBnChunk - 1) / kBnChunk : 1;
  f.rcpHW = 1.0f / (float)p.HW;
  f.rcpC = 1.0f / (float)p.C;
  const long NC = (long)p.N * p.C;
  const long fgrid = f.chunks > 1 ? NC * f.chunks : (NC + f.G - 1) / f.G;
  const dim3 agrid((unsigned)((p.HW + 256 * vn - 1) / (256 * vn)), (unsigned)NC);
#define SSDK_BN(DT)                                                                                         \
  do {                                                                                                      \
    if (MODE == 0 && p.sums) {                                                                              \
    } else if (flat_r) hipLaunchKernelGGL((bn_reduce_flat_kernel<DT, MODE>), rgrid, dim3(256), 0, st, p);   \
    else hipLaunchKernelGGL((bn_reduce_kernel<DT, MODE>), rgrid, dim3(256), 0, st, p);                      \
    if (p.tickets) {                                                                                        \
    } else if (MODE == 0) hipLaunchKernelGGL((bn_fwd_finalize_kernel<DT>), dim3((unsigned)((p.C + 63) / 64)), dim3(64), 0, st, p); \
    else hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((p.C + 63) / 64)), dim3(64), 0, st, p);  \
    if (MODE == 0 && p.no_apply) {                                                                           \
    } else if (flat && fgrid < (1l << 31)) hipLaunchKernelGGL((bn_apply_flat_kernel<DT, MODE>), dim3((unsigned)fgrid), dim3(256), 0, st, p, f); \
    else hipLaunchKernelGGL((bn_apply_kernel<DT, MODE>), agrid, dim3(256), 0, st, p);                        \
  } while (0)
  if (p.dtype == SSDK_F32) SSDK_BN(SSDK_F32);
  else if (p.dtype == SSDK_BF16) SSDK_BN(SSDK_BF16);
  else SSDK_BN(SSDK_F16);
#undef SSDK_BN
}

}  // namespace ssdk

using namespace ssdk;

extern "C" size_t ssdk_bn_workspace_bytes(int N, int C) {
  // partial sums [C][split][2], rounded up to a multiple of 4 floats (the coefficient table behind it stays 16-byte aligned)
  return ((((size_t)C * bn_split(N, C) * 2 + 3) & ~(size_t)3) + (size_t)C * 4) * sizeof(float);
}

static int bn_common(BnParams& p, const char* what, int N, int C, int HW, int dtype, void* workspace, size_t workspace_bytes) {
  if (N < 1 || C < 1 || HW < 1 || (dtype != SSDK_F32 && dtype != SSDK_BF16 && dtype != SSDK_F16) || (long)N * C > 2147483647l) {
    set_error("%s: bad arguments N=%d C=%d HW=%d dtype=%d", what, N, C, HW, dtype);
    return SSDK_E_BADARG;
  }
  if (!workspace || workspace_bytes < ssdk_bn_workspace_bytes(N, C) || ((uintptr_t)workspace & 15)) {
    set_error("%s: workspace too small or misaligned", what);
    return SSDK_E_BADARG;
  }
  p.N = N;
  p.C = C;
  p.HW = HW;
  p.dtype = dtype;
  p.split = bn_split(N, C);
  p.partial = (float*)workspace;
  p.coef = p.partial + (((size_t)C * p.split * 2 + 3) & ~(size_t)3);
  return SSDK_OK;
}

extern "C" int ssdk_bn_act_train_fwd(const void* x, const float* weight, const float* bias, float* running_mean,
                                     float* running_var, void* y, float* save_mean, float* save_invstd, void* workspace,
                                     size_t workspace_bytes, int N, int C, int HW, float momentum, float eps, int act,
                                     int dtype, void* stream) {
  if (act < 0 || act > 2) {
    set_error("bn_train_fwd: act must be 0 (none), 1 (ReLU6) or 2 (ReLU)");
    return SSDK_E_BADARG;
  }
  if (!x || !y || !save_mean || !save_invstd || (!running_mean) != (!running_var)) {
    set_error("bn_train_fwd: null pointer");
    return SSDK_E_BADARG;
  }
  BnParams p;
  memset(&p, 0, sizeof(p));
  const int rc = bn_common(p, "bn_train_fwd", N, C, HW, dtype, workspace, workspace_bytes);
  if (rc) return rc;
  p.x = x;
  p.dy = x;
  p.out = y;
  p.weight = weight;
  p.bias = bias;
  p.running_mean = running_mean;
  p.running_var = running_var;
  p.save_mean = save_mean;
  p.save_invstd = save_invstd;
  p.momentum = momentum;
  p.eps = eps;
  p.act = act;
  bn_launch<0>(p, (hipStream_t)stream);
  return check_launch("bn_train_fwd");
}

extern "C" int ssdk_bn_act_train_fwd_sums(const void* x, const float* sums, const float* weight, const float* bias,
                                          float* running_mean, float* running_var, void* y, float* save_mean, float* save_invstd,
                                          void* workspace, size_t workspace_bytes, int N, int C, int HW, float momentum, float eps,
                                          int act, int dtype, void* stream) {
  if (act < 0 || act > 2 || !sums) {
    set_error("bn_train_fwd_sums: act must be 0 (none), 1 (ReLU6) or 2 (ReLU); sums must not be null");
    return SSDK_E_BADARG;
  }
  if (!x || !y || !save_mean || !save_invstd || (!running_mean) != (!running_var)) {
    set_error("bn_train_fwd_sums: null pointer");
    return SSDK_E_BADARG;
  }
  BnParams p;
  memset(&p, 0, sizeof(p));
  const int rc = bn_common(p, "bn_train_fwd_sums", N, C, HW, dtype, workspace, workspace_bytes);
  if (rc) return rc;
  p.x = x;
  p.dy = x;
  p.out = y;
  p.sums = sums;
  p.weight = weight;
  p.bias = bias;
  p.running_mean = running_mean;
  p.running_var = running_var;
  p.save_mean = save_mean;
  p.save_invstd = save_invstd;
  p.momentum = momentum;
  p.eps = eps;
  p.act = act;
  bn_launch<0>(p, (hipStream_t)stream);
  return check_launch("bn_train_fwd_sums");
}

// The forward pass WITHOUT its apply pass: batch statistics (from `sums` when the producer provided them, else by the reduction
// over x), running statistics, save_mean / save_invstd and coef_out [C][4] = (a, b, 0, 0) with y = act(a x + b).  The consumer of
// the BatchNorm's output applies the coefficients when it loads x (ssdk_dwconv_fwd_affine): y is never written.
extern "C" int ssdk_bn_act_train_stats(const void* x, const float* sums, const float* weight, const float* bias, float* running_mean,
                                       float* running_var, float* save_mean, float* save_invstd, float* coef_out, void* workspace,
                                       size_t workspace_bytes, int N, int C, int HW, float momentum, float eps, int dtype,
                                       void* stream) {
  if (!x || !save_mean || !save_invstd || !coef_out || ((uintptr_t)coef_out & 15) || (!running_mean) != (!running_var)) {
    set_error("bn_train_stats: null / misaligned pointer");
    return SSDK_E_BADARG;
  }
  BnParams p;
  memset(&p, 0, sizeof(p));
  const int rc = bn_common(p, "bn_train_stats", N, C, HW, dtype, workspace, workspace_bytes);
  if (rc) return rc;
  p.x = x;
  p.dy = x;
  p.out = nullptr;
  p.sums = sums;
  p.no_apply = 1;
  p.coef = coef_out;
  p.weight = weight;
  p.bias = bias;
  p.running_mean = running_mean;
  p.running_var = running_var;
  p.save_mean = save_mean;
  p.save_invstd = save_invstd;
  p.momentum = momentum;
  p.eps = eps;
  bn_launch<0>(p, (hipStream_t)stream);
  return check_launch("bn_train_stats");
}

extern "C" int ssdk_bn_train_fwd(const void* x, const float* weight, const float* bias, float* running_mean,
                                 float* running_var, void* y, float* save_mean, float* save_invstd, void* workspace,
                                 size_t workspace_bytes, int N, int C, int HW, float momentum, float eps, int dtype,
                                 void* stream) {
  return ssdk_bn_act_train_fwd(x, weight, bias, running_mean, running_var, y, save_mean, save_invstd, workspace,
                               workspace_bytes, N, C, HW, momentum, eps, 0, dtype, stream);
}

extern "C" int ssdk_bn_act_train_bwd(const void* x, const void* dy, const float* weight, const float* bias,
                                     const float* save_mean, const float* save_invstd, void* dx, float* dweight,
                                     float* dbias, void* workspace, size_t workspace_bytes, int N, int C, int HW, int act,
                                     int dtype, void* stream) {
  if (act < 0 || act > 2) {
    set_error("bn_train_bwd: act must be 0 (none), 1 (ReLU6) or 2 (ReLU)");
    return SSDK_E_BADARG;
  }
  if (!x || !dy || !dx || !save_mean || !save_invstd) {
    set_error("bn_train_bwd: null pointer");
    return SSDK_E_BADARG;
  }
  BnParams p;
  memset(&p, 0, sizeof(p));
  const int rc = bn_common(p, "bn_train_bwd", N, C, HW, dtype, workspace, workspace_bytes);
  if (rc) return rc;
  p.x = x;
  p.dy = dy;
  p.out = dx;
  p.weight = weight;
  p.bias = bias;
  p.act = act;
  p.save_mean = const_cast<float*>(save_mean);
  p.save_invstd = const_cast<float*>(save_invstd);
  p.dweight = dweight;
  p.dbias = dbias;
  bn_launch<1>(p, (hipStream_t)stream);
  return check_launch("bn_train_bwd");
}

extern "C" int ssdk_bn_train_bwd(const void* x, const void* dy, const float* weight, const float* save_mean,
                                 const float* save_invstd, void* dx, float* dweight, float* dbias, void* workspace,
                                 size_t workspace_bytes, int N, int C, int HW, int dtype, void* stream) {
  return ssdk_bn_act_train_bwd(x, dy, weight, nullptr, save_mean, save_invstd, dx, dweight, dbias, workspace,
                               workspace_bytes, N, C, HW, 0, dtype, stream);
}
