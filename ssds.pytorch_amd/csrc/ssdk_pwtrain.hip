// ssdk_pwtrain.hip -- the dense 1x1 convolutions of the TRAINING step on the NCHW tensors themselves: forward, input gradient
// and weight gradient on the matrix cores, no layout change, no library.
//
// Reference: the pointwise convolutions of torchvision's InvertedResidual / ConvBNReLU inside MobileNetV2 (nets/mobilenet.py:56,
// 78, 180-192) and of the SSD extras (layers/basic_layers.py:40-57), forward + backward under Apex AMP O1 in the reference's DDP
// step (pipeline/pipeline_anchor_apex.py:103-130, utils/train_ddp.py:106-108).  Rounds 2-5 ran them as hipBLASLt strided-batched
// GEMMs through torch.matmul / torch.bmm ("library dispatch is not implemented").
//
//     y[b]  = W   x[b]            [Cout,Cin] [Cin,HW]     forward            pw_gemm_kernel   (a = W)
//     dx[b] = W^T dy[b]           [Cin,Cout] [Cout,HW]    input gradient     pw_gemm_kernel   (a = W^T, prepared once per step)
//     dW    = sum_b dy[b] x[b]^T  [Cout,HW]  [HW,Cin]     weight gradient    pw_wgrad_kernel + pw_wgrad_reduce_kernel
//
// All three are STREAMS at the shapes of this network (16 ... 960 channels): in + out bytes decide, so nothing is staged through
// LDS except the (small) weight matrix.  What makes NCHW natural for v_mfma_f32_16x16x32:
//   * y = a x: the B operand of an MFMA is "8 consecutive k of one column per lane".  In NCHW the 8 k (channels) of a pixel are
//     HW elements apart -- but WHICH pixel a lane's column is, is free.  A lane loads 16 bytes = 8 consecutive PIXELS of one
//     channel for each of its 8 channels (8 loads, every wave-level load is 16 lanes x 16 B = 256 contiguous bytes of a channel
//     row), transposes the 8 x 8 block in its own registers with 32 v_perm_b32, and owns the B operands of EIGHT MFMAs: MFMA t
//     multiplies the pixel set {p0 + 8 fr + t}.  Its accumulators then hold, per output channel row, 8 consecutive pixels =
//     one 16-byte store into the NCHW output.  No LDS transpose on either side.
//   * dW = dy x^T contracts over PIXELS, which are contiguous in both operands: A fragments of dy and B fragments of x are plain
//     16-byte loads: a k-step is 32 consecutive pixels, each load instruction reads 64 contiguous bytes of each of its 16 rows,
//     four k-steps (128 pixels) are in flight per chunk.
//   * the weight gradient is split over (tile of the [Cout, Cin] matrix) x (range of pixels); every wave writes its fp32 partial
//     tile to the workspace and pw_wgrad_reduce_kernel adds the partials in index order: no float atomics, bit-reproducible.
// HBM bytes: forward (Cin + Cout) * 2 per pixel, input gradient the same, weight gradient (Cin + Cout) * 2 per pixel read.
#include "ssdk_conv_common.h"

namespace ssdk {

typedef u32x4 pw_u32x4_a2 __attribute__((aligned(2)));
constexpr int PWT_THREADS = 256;
constexpr int PWT_KC = 8;  // k-steps (of 32 channels) of weights staged per chunk in the long-K kernel

struct PwtParams {
  const u16* x;       // [B, K, HW]
  const u16* a;       // [M, K] row-major, 16 bit
  const float* bias;  // [M] or null
  u16* y;             // [B, M, HW]
  int B, K, M, HW;
  int KS;             // ceil(K / 32)
  int nf;             // output fragments (16 rows) per workgroup slice
  u32 gpi;            // 128-pixel groups per image
  u32 groups;         // B * gpi
  u32 wg_iters;       // (P) iterations of a workgroup (4 groups each)
  float* stats;       // optional [gridDim.x * 4 waves][M][2]: per wave (sum y, sum y^2) of its outputs (fp32, before the store's rounding)
};

// 8 consecutive 16-bit elements at p (any 2-byte alignment); lanes with nvalid < 8 read element by element and zero-fill
template <bool TAIL>
__device__ __forceinline__ u32x4 pw_load8(const u16* p, int nvalid) {
  if constexpr (!TAIL) {
    return *reinterpret_cast<const pw_u32x4_a2*>(p);
  } else {
    if (nvalid >= 8) return *reinterpret_cast<const pw_u32x4_a2*>(p);
    u32 h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = e < nvalid ? (u32)p[e] : 0u;
    return u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  }
}
template <bool TAIL>
__device__ __forceinline__ void pw_store8(u16* p, const u32x4 v, int nvalid) {
  if constexpr (!TAIL) {
    *reinterpret_cast<pw_u32x4_a2*>(p) = v;
  } else {
    if (nvalid >= 8) {
      *reinterpret_cast<pw_u32x4_a2*>(p) = v;
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < nvalid) p[e] = (u16)((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
  }
}

// raw[j] = 8 pixels of channel j of the lane's 8 channels  ->  bop[t] = the 8 channels of pixel t (MFMA B operand of pixel set t)
__device__ __forceinline__ void pw_transpose8(const u32x4 (&raw)[8], u32x4 (&bop)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const u32 lo = raw[2 * q][h], hi = raw[2 * q + 1][h];
      bop[2 * h][q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);      // low halves: pixel 2h of channels 2q, 2q + 1
      bop[2 * h + 1][q] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // high halves: pixel 2h + 1
    }
  }
}

// the lane's 8 x 8 block of k-step ks: channels 32 ks + 8 fg + (0..7), pixels px .. px + 7 of image base xb
template <bool TAIL>
__device__ __forceinline__ void pw_load_raw(const u16* xb, u32 HW, u32 K, u32 chan0, int nvalid, u32x4 (&raw)[8]) {
  if (chan0 < K) {  // (K is a multiple of 8: all eight channels or none)
    const u16* p = xb + (size_t)chan0 * HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = pw_load8<TAIL>(p + (size_t)j * HW, nvalid);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = u32x4{0u, 0u, 0u, 0u};
  }
}

// A fragment (f, ks) of the slice: lane (fr, fg) holds a[m0 + 16 f + fr][32 ks + 8 fg .. + 7], zero outside the matrix
__device__ __forceinline__ void pw_stage_a(unsigned char* lds, const u16* a, u32 M, u32 K, u32 m0, u32 nf, u32 ks0, u32 nks,
                                           u32 tid) {
  for (u32 i = tid; i < nf * nks * 64u; i += PWT_THREADS) {
    const u32 l = i & 63u, fk = i >> 6, ks = fk % nks, f = fk / nks;
    const u32 row = m0 + 16u * f + (l & 15u), k0 = (ks0 + ks) * 32u + (l >> 4) * 8u;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < M && k0 < K) v = *reinterpret_cast<const u32x4*>(a + (size_t)row * K + k0);
    *reinterpret_cast<u32x4*>(lds + (size_t)i * 16u) = v;
  }
}

// (sum, sum of squares) of row i of the fragment over the lane's 8 pixels that lie inside the plane -- of the fp32 accumulators,
// BEFORE the rounding of the store: the statistics of a BatchNorm over 10^5 ... 10^6 elements do not see 2^-9 of zero-mean
// rounding per element, and unpacking the stored words cost as much as the reduction pass it was meant to save (round 6, first
// version: forward + statistics 1 517 vs 1 088 us per step)
__device__ __forceinline__ void pw_row_moments(const f32x4 (&acc)[8], int i, int nvalid, float& s1, float& s2) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float v = t < nvalid ? acc[t][i] : 0.f;
    s1 += v;
    s2 += v * v;
  }
}
// sum over the 16 lanes fr of a lane group (all four groups at once)
__device__ __forceinline__ float pw_sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// finished accumulators of fragment f (rows 4 fg + i of it, pixels 8 fr + t) -> four 16-byte stores
// STATS: m1[i] / m2[i] += the lane's share of (sum, sum of squares) of row i, over the pixels inside the plane
template <int DT, bool TAIL, bool STATS = false>
__device__ __forceinline__ void pw_store_frag(const f32x4 (&acc)[8], u16* yb, u32 HW, u32 M, u32 row0, int nvalid, float* m1 = nullptr,
                                              float* m2 = nullptr) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (row0 + (u32)i < M) {
      const u32x4 o = {pack2_16<DT>(acc[0][i], acc[1][i]), pack2_16<DT>(acc[2][i], acc[3][i]),
                       pack2_16<DT>(acc[4][i], acc[5][i]), pack2_16<DT>(acc[6][i], acc[7][i])};
      pw_store8<TAIL>(yb + (size_t)(row0 + (u32)i) * HW, o, nvalid);
      if constexpr (STATS) pw_row_moments(acc, i, TAIL ? nvalid : 8, m1[i], m2[i]);
    }
  }
}

// ---- (E) short K (<= 96 channels): the B operands of a 128-pixel group stay in registers, the wave walks over ALL output
// fragments of the slice.  Expansion layers (16 -> 96 ... 96 -> 576) and the input gradients of the projections. ---------------
template <int DT, int KS, bool STATS = false>
__global__ __launch_bounds__(PWT_THREADS) void pw_gemm_short_kernel(const PwtParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4;
  const u32 M = (u32)p.M, K = (u32)p.K, HW = (u32)p.HW;
  const u32 m0 = blockIdx.y * (u32)(16 * p.nf);
  const u32 nf = min((u32)p.nf, (M - m0 + 15u) / 16u);
  pw_stage_a(smem, p.a, M, K, m0, nf, 0u, (u32)KS, tid);
  // STATS: this wave's (sum, sum of squares) per output row of the slice, behind the A fragments: [nf * 16 rows][2]
  float* wst = reinterpret_cast<float*>(smem + (size_t)p.nf * KS * 1024) + (size_t)wave * (u32)p.nf * 32u;
  if constexpr (STATS) {
    for (u32 i = lane; i < (u32)p.nf * 32u; i += 64u) wst[i] = 0.f;
  }
  __syncthreads();
  const u32 gstride = gridDim.x * 4u;
  for (u32 g = blockIdx.x * 4u + wave; g < p.groups; g += gstride) {
    const u32 b = g / p.gpi, p0 = (g - b * p.gpi) * 128u + 8u * fr;
    const bool tail = (g - b * p.gpi) * 128u + 128u > HW;  // wave-uniform
    const int nvalid = (int)HW - (int)p0;
    const u16* xb = p.x + (size_t)b * K * HW + p0;
    u16* yb = p.y + (size_t)b * M * HW + p0;
    u32x4 bop[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw[8];
      if (tail) pw_load_raw<true>(xb, HW, K, (u32)ks * 32u + fg * 8u, nvalid, raw);
      else pw_load_raw<false>(xb, HW, K, (u32)ks * 32u + fg * 8u, 8, raw);
      pw_transpose8(raw, bop[ks]);
    }
    for (u32 f = 0; f < nf; ++f) {
      f32x4 acc[8];
      const u32 row0 = m0 + 16u * f + 4u * fg;
      f32x4 init = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) init[i] = row0 + (u32)i < M ? p.bias[row0 + (u32)i] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = init;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 wa = *reinterpret_cast<const u32x4*>(smem + ((size_t)(f * (u32)KS + (u32)ks) * 64u + lane) * 16u);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = mfma16<DT>(wa, bop[ks][t], acc[t]);
      }
      if constexpr (STATS) {
        float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
        if (tail) pw_store_frag<DT, true, true>(acc, yb, HW, M, row0, nvalid, m1, m2);
        else pw_store_frag<DT, false, true>(acc, yb, HW, M, row0, 8, m1, m2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a1 = pw_sum16(m1[i]), a2 = pw_sum16(m2[i]);
          if (fr == 0u) {  // (the wave's own words: no atomics)
            wst[(16u * f + 4u * fg + (u32)i) * 2u + 0u] += a1;
            wst[(16u * f + 4u * fg + (u32)i) * 2u + 1u] += a2;
          }
        }
      } else {
        if (tail) pw_store_frag<DT, true>(acc, yb, HW, M, row0, nvalid);
        else pw_store_frag<DT, false>(acc, yb, HW, M, row0, 8);
      }
    }
  }
  if constexpr (STATS) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* dst = p.stats + ((size_t)(blockIdx.x * 4u + wave) * M + m0) * 2u;
    for (u32 i = lane; i < nf * 32u; i += 64u)
      if (m0 + (i >> 1) < M) dst[i] = wst[i];
  }
}

// ---- (P) long K: NF output fragments x 8 pixel sets of accumulators per wave, k-steps outermost, the weights of the slice
// staged in chunks of PWT_KC k-steps (a workgroup's four waves walk through the chunks together).  Projection layers
// (96 ... 960 -> 16 ... 320) and the input gradients of the expansions. -------------------------------------------------------
template <int DT, int NF, bool STATS = false>
__global__ __launch_bounds__(PWT_THREADS) void pw_gemm_long_kernel(const PwtParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4;
  const u32 M = (u32)p.M, K = (u32)p.K, HW = (u32)p.HW, KS = (u32)p.KS;
  const u32 m0 = blockIdx.y * (u32)(16 * NF);
  const bool one_chunk = KS <= (u32)PWT_KC;
  if (one_chunk) {
    pw_stage_a(smem, p.a, M, K, m0, (u32)NF, 0u, KS, tid);
    __syncthreads();
  }
  float sm1[STATS ? NF : 1][4], sm2[STATS ? NF : 1][4];  // STATS: the lane's share of (sum, sum of squares) per output row
#pragma unroll
  for (int f = 0; f < (STATS ? NF : 1); ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) sm1[f][i] = sm2[f][i] = 0.f;
  for (u32 it = blockIdx.x; it < p.wg_iters; it += gridDim.x) {
    const u32 g = it * 4u + wave;
    const bool live = g < p.groups;  // wave-uniform; a dead wave still stages and meets the barriers
    const u32 gg = live ? g : p.groups - 1u;
    const u32 b = gg / p.gpi, p0 = (gg - b * p.gpi) * 128u + 8u * fr;
    const bool tail = (gg - b * p.gpi) * 128u + 128u > HW;
    const int nvalid = (int)HW - (int)p0;
    const u16* xb = p.x + (size_t)b * K * HW + p0;
    u16* yb = p.y + (size_t)b * M * HW + p0;
    f32x4 acc[NF][8];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const u32 row0 = m0 + 16u * (u32)f + 4u * fg;
      f32x4 init = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) init[i] = row0 + (u32)i < M ? p.bias[row0 + (u32)i] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[f][t] = init;
    }
    u32x4 raw[8];
    if (tail) pw_load_raw<true>(xb, HW, K, fg * 8u, nvalid, raw);
    else pw_load_raw<false>(xb, HW, K, fg * 8u, 8, raw);
    for (u32 c0 = 0; c0 < KS; c0 += (u32)PWT_KC) {
      const u32 nks = min((u32)PWT_KC, KS - c0);
      if (!one_chunk) {
        __syncthreads();  // every wave is done with the previous chunk's fragments
        pw_stage_a(smem, p.a, M, K, m0, (u32)NF, c0, nks, tid);
        __syncthreads();
      }
      for (u32 ks = 0; ks < nks; ++ks) {
        u32x4 bop[8];
        pw_transpose8(raw, bop);
        const u32 kn = c0 + ks + 1u;
        if (kn < KS) {  // the next k-step's block travels under this one's MFMAs
          if (tail) pw_load_raw<true>(xb, HW, K, kn * 32u + fg * 8u, nvalid, raw);
          else pw_load_raw<false>(xb, HW, K, kn * 32u + fg * 8u, 8, raw);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const u32x4 wa = *reinterpret_cast<const u32x4*>(smem + ((size_t)((u32)f * nks + ks) * 64u + lane) * 16u);
#pragma unroll
          for (int t = 0; t < 8; ++t) acc[f][t] = mfma16<DT>(wa, bop[t], acc[f][t]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const u32 row0 = m0 + 16u * (u32)f + 4u * fg;
        if constexpr (STATS) {
          if (tail) pw_store_frag<DT, true, true>(acc[f], yb, HW, M, row0, nvalid, sm1[f], sm2[f]);
          else pw_store_frag<DT, false, true>(acc[f], yb, HW, M, row0, 8, sm1[f], sm2[f]);
        } else {
          if (tail) pw_store_frag<DT, true>(acc[f], yb, HW, M, row0, nvalid);
          else pw_store_frag<DT, false>(acc[f], yb, HW, M, row0, 8);
        }
      }
    }
  }
  if constexpr (STATS) {
    float* dst = p.stats + (size_t)(blockIdx.x * 4u + wave) * M * 2u;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a1 = pw_sum16(sm1[f][i]), a2 = pw_sum16(sm2[f][i]);
        const u32 row = m0 + 16u * (u32)f + 4u * fg + (u32)i;
        if (fr == 0u && row < M) {
          dst[row * 2u + 0u] = a1;
          dst[row * 2u + 1u] = a2;
        }
      }
  }
}

struct PwtPlan {
  bool is_short;
  int slices, nf;
  unsigned gx;
  size_t lds;
};

// geometry of a y = a x launch (shared by the launch and by the statistics workspace query)
static PwtPlan pw_gemm_plan(PwtParams& p, int B, int K, int M, int HW, bool stats) {
  p.B = B;
  p.K = K;
  p.M = M;
  p.HW = HW;
  p.KS = (K + 31) / 32;
  p.gpi = (u32)((HW + 127) / 128);
  p.groups = (u32)B * p.gpi;
  p.wg_iters = (p.groups + 3u) / 4u;
  const int mf = (M + 15) / 16;
  static const int force_long = getenv("SSDK_PW_LONG") ? atoi(getenv("SSDK_PW_LONG")) : 0;
  PwtPlan pl;
  pl.is_short = p.KS <= 3 && !force_long;
  if (pl.is_short) {
    // slice: all output fragments if their A fragments fit 64 KiB, else equal slices
    int slices = (mf * p.KS + 63) / 64;
    p.nf = (mf + slices - 1) / slices;
    pl.slices = (mf + p.nf - 1) / p.nf;
    pl.nf = p.nf;
    pl.lds = (size_t)p.nf * p.KS * 1024 + (stats ? (size_t)4 * p.nf * 32 * sizeof(float) : 0);
    pl.gx = (unsigned)(256 * 4 / pl.slices);  // ~4 workgroups per CU over all slices
  } else {
    // long K: NF <= 4 fragments per slice, equal slices -- and MORE slices (each re-reads x, from L2) when the pixels alone give the
    // chip too few workgroups: 960 -> 160 on 16 x 16 maps at batch 64 is 32 workgroup iterations; as 3 slices of 4 fragments it
    // ran on 96 workgroups in 33 us (library GEMM: 22), as 5 slices of 2 on 160
    int slices = (mf + 3) / 4;
    const int wanted = (int)((256u + p.wg_iters - 1u) / p.wg_iters);
    if (slices < wanted) slices = wanted < mf ? wanted : mf;
    const int nf = (mf + slices - 1) / slices;
    pl.slices = (mf + nf - 1) / nf;
    pl.nf = p.nf = nf;
    const int nks = p.KS < PWT_KC ? p.KS : PWT_KC;
    pl.lds = (size_t)nf * nks * 1024;
    pl.gx = (unsigned)(256 * 3 / pl.slices);
  }
  if (pl.gx < 1u) pl.gx = 1u;
  if (pl.gx > p.wg_iters) pl.gx = p.wg_iters;
  return pl;
}

// stats_ws (optional): [gx * 4 waves][M][2] fp32 per-wave moments of the stored outputs (the caller reduces them)
static int pw_gemm_launch(const void* x, const void* a, const float* bias, void* y, int B, int K, int M, int HW, int dtype,
                          hipStream_t stream, float* stats_ws = nullptr, unsigned* rows_out = nullptr) {
  PwtParams p;
  p.x = (const u16*)x;
  p.a = (const u16*)a;
  p.bias = bias;
  p.y = (u16*)y;
  p.stats = stats_ws;
  const bool st = stats_ws != nullptr;
  const PwtPlan pl = pw_gemm_plan(p, B, K, M, HW, st);
  if (rows_out) *rows_out = pl.gx * 4u;
  const dim3 grid(pl.gx, (unsigned)pl.slices);
  const size_t lds = pl.lds;
  const int nf = pl.nf;
  if (pl.is_short) {
#define SSDK_PWS(DT, KS_, ST_)                                                                                         \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_short_kernel<DT, KS_, ST_>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    hipLaunchKernelGGL((pw_gemm_short_kernel<DT, KS_, ST_>), grid, dim3(PWT_THREADS), lds, stream, p);                \
  } while (0)
#define SSDK_PWSK(DT, KS_)             \
  do {                                 \
    if (st) SSDK_PWS(DT, KS_, true);   \
    else SSDK_PWS(DT, KS_, false);     \
  } while (0)
#define SSDK_PWSD(DT)                     \
  do {                                    \
    if (p.KS == 1) SSDK_PWSK(DT, 1);      \
    else if (p.KS == 2) SSDK_PWSK(DT, 2); \
    else SSDK_PWSK(DT, 3);                \
  } while (0)
    if (dtype == SSDK_BF16) SSDK_PWSD(SSDK_BF16);
    else SSDK_PWSD(SSDK_F16);
#undef SSDK_PWSD
#undef SSDK_PWSK
#undef SSDK_PWS
    return check_launch("pw_gemm_short_kernel");
  }
#define SSDK_PWL(DT, NF_, ST_)                                                                                        \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_long_kernel<DT, NF_, ST_>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    hipLaunchKernelGGL((pw_gemm_long_kernel<DT, NF_, ST_>), grid, dim3(PWT_THREADS), lds, stream, p);                 \
  } while (0)
#define SSDK_PWLN(DT, NF_)            \
  do {                                \
    if (st) SSDK_PWL(DT, NF_, true);  \
    else SSDK_PWL(DT, NF_, false);    \
  } while (0)
#define SSDK_PWLD(DT)                    \
  do {                                   \
    if (nf == 1) SSDK_PWLN(DT, 1);       \
    else if (nf == 2) SSDK_PWLN(DT, 2);  \
    else if (nf == 3) SSDK_PWLN(DT, 3);  \
    else SSDK_PWLN(DT, 4);               \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_PWLD(SSDK_BF16);
  else SSDK_PWLD(SSDK_F16);
#undef SSDK_PWLD
#undef SSDK_PWLN
#undef SSDK_PWL
  return check_launch("pw_gemm_long_kernel");
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
struct PwgParams {
  const u16* dy;  // [B, Cout, HW]
  const u16* x;   // [B, Cin, HW]
  float* ws;      // [S][Cout][Cin] fp32 partials
  int B, Cout, Cin, HW;
  u32 cpi;        // 128-pixel chunks per image
  u32 chunks;     // B * cpi
  u32 mt, nt;     // tiles of the [Cout, Cin] matrix
  u32 splits;     // pixel ranges
  u32 cps;        // chunks per split
};

template <int DT, int MFW, int NFW>
__global__ __launch_bounds__(PWT_THREADS) void pw_wgrad_kernel(const PwgParams p) {
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4;
  const u32 tiles = p.mt * p.nt;
  const u32 item = blockIdx.x * 4u + wave;
  if (item >= tiles * p.splits) return;
  const u32 tile = item % tiles, s = item / tiles;
  const u32 m0 = (tile / p.nt) * (u32)(16 * MFW), n0 = (tile % p.nt) * (u32)(16 * NFW);
  const u32 Cout = (u32)p.Cout, Cin = (u32)p.Cin, HW = (u32)p.HW;
  f32x4 acc[MFW][NFW];
#pragma unroll
  for (int i = 0; i < MFW; ++i)
#pragma unroll
    for (int j = 0; j < NFW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32 c_end = min(p.chunks, (s + 1u) * p.cps);
  for (u32 c = s * p.cps; c < c_end; ++c) {
    // k-step u of the chunk = its pixels 32 u .. 32 u + 31, lane group fg the 8 pixels 32 u + 8 fg ..: ONE load instruction
    // reads 64 contiguous bytes of each of its 16 rows.  (As first written a lane took 32 CONSECUTIVE pixels over its four loads,
    // i.e. every instruction touched sixteen 16-byte pieces 64 bytes apart per row: the 32 x 32 layers ran at 0.19 of the roof.)
    const u32 b = c / p.cpi, p0 = (c - b * p.cpi) * 128u + 8u * fg;
    const bool tail = (c - b * p.cpi) * 128u + 128u > HW;  // wave-uniform
    u32x4 av[MFW][4], bv[NFW][4];
#pragma unroll
    for (int i = 0; i < MFW; ++i) {
      const u32 row = m0 + 16u * (u32)i + fr;
      const u16* src = p.dy + ((size_t)b * Cout + row) * HW + p0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row < Cout) av[i][u] = tail ? pw_load8<true>(src + 32 * u, (int)HW - (int)p0 - 32 * u) : pw_load8<false>(src + 32 * u, 8);
        else av[i][u] = u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
      const u32 row = n0 + 16u * (u32)j + fr;
      const u16* src = p.x + ((size_t)b * Cin + row) * HW + p0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row < Cin) bv[j][u] = tail ? pw_load8<true>(src + 32 * u, (int)HW - (int)p0 - 32 * u) : pw_load8<false>(src + 32 * u, 8);
        else bv[j][u] = u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < MFW; ++i)
#pragma unroll
        for (int j = 0; j < NFW; ++j) acc[i][j] = mfma16<DT>(av[i][u], bv[j][u], acc[i][j]);
  }
  // partial tile -> workspace: D[row 4 fg + r of fragment i][column fr of fragment j]
  float* ws = p.ws + (size_t)s * Cout * Cin;
#pragma unroll
  for (int i = 0; i < MFW; ++i)
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
      const u32 ci = n0 + 16u * (u32)j + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const u32 co = m0 + 16u * (u32)i + 4u * fg + (u32)r;
        if (co < Cout && ci < Cin) ws[(size_t)co * Cin + ci] = acc[i][j][r];
      }
    }
}

// ---- weight gradient of the layers whose [Cout, Cin] matrix is large on BOTH sides (the 32 x 32 / 16 x 16 blocks: 64 ... 960
// channels either side).  The streaming kernel above re-reads dy once per column tile and x once per row tile -- 4.3 x the bytes
// on 96 -> 576 -- in 16-row x 64-byte pieces: 0.2 of the roof there.  Here a workgroup owns a 128 x 128 tile of dW and walks
// over pixels in chunks of 128: both operand tiles are staged in LDS with full-row 256-byte global reads (12 independent
// 16-byte loads per thread per chunk, the next chunk's requested before this chunk's MFMAs), a wave reads its 4 + 4 fragments
// per k-step from LDS (row stride 288 bytes = 18 x 16: conflict-free for the lane groups of ds_read_b128).
constexpr u32 PWG_RS = 288;
constexpr size_t PWG_LDS = 2 * 128 * PWG_RS;

template <int DT>
__global__ __launch_bounds__(PWT_THREADS) void pw_wgrad_tiled_kernel(const PwgParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* la = smem;
  unsigned char* lb = smem + 128 * PWG_RS;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4, wm = wave >> 1, wn = wave & 1u;
  const u32 tiles = p.mt * p.nt;
  const u32 tile = blockIdx.x % tiles, s = blockIdx.x / tiles;
  const u32 m0 = (tile / p.nt) * 128u, n0 = (tile % p.nt) * 128u;
  const u32 Cout = (u32)p.Cout, Cin = (u32)p.Cin, HW = (u32)p.HW;
  const u32 r16 = tid >> 4, c16 = tid & 15u;  // staging role: row r16 (+ 16 per pass), 16-byte column c16 of the 256-byte row
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32 c_begin = s * p.cps, c_end = min(p.chunks, (s + 1u) * p.cps);
  u32x4 va[8], vb[8];
  auto fetch = [&](u32 c) {
    const u32 b = c / p.cpi, p0 = (c - b * p.cpi) * 128u + 8u * c16;
    const int nvalid = (int)HW - (int)p0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const u32 ra = m0 + (u32)q * 16u + r16, rb = n0 + (u32)q * 16u + r16;
      va[q] = ra < Cout ? pw_load8<true>(p.dy + ((size_t)b * Cout + ra) * HW + p0, nvalid) : u32x4{0u, 0u, 0u, 0u};
      vb[q] = rb < Cin ? pw_load8<true>(p.x + ((size_t)b * Cin + rb) * HW + p0, nvalid) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (u32 c = c_begin; c < c_end; ++c) {
    __syncthreads();  // every wave is done with the previous chunk's fragments
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      *reinterpret_cast<u32x4*>(la + ((u32)q * 16u + r16) * PWG_RS + c16 * 16u) = va[q];
      *reinterpret_cast<u32x4*>(lb + ((u32)q * 16u + r16) * PWG_RS + c16 * 16u) = vb[q];
    }
    __syncthreads();
    if (c + 1u < c_end) fetch(c + 1u);  // travels under this chunk's MFMAs
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const u32x4*>(la + (64u * wm + 16u * (u32)i + fr) * PWG_RS + (u32)u * 64u + fg * 16u);
        bf[i] = *reinterpret_cast<const u32x4*>(lb + (64u * wn + 16u * (u32)i + fr) * PWG_RS + (u32)u * 64u + fg * 16u);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(af[i], bf[j], acc[i][j]);
    }
  }
  float* ws = p.ws + (size_t)s * Cout * Cin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 ci = n0 + 64u * wn + 16u * (u32)j + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const u32 co = m0 + 64u * wm + 16u * (u32)i + 4u * fg + (u32)r;
        if (co < Cout && ci < Cin) ws[(size_t)co * Cin + ci] = acc[i][j][r];
      }
    }
}

// rows [by * 64, by * 64 + 64) of a [rows][n] fp32 matrix summed per element in a FIXED order: thread (element, quarter kq) adds its
// 16 rows one after the other (sixteen independent coalesced loads), the four quarters are added in order 0..3.  Applied until one
// row is left (<= 4096 partials: two passes).  (As first written ONE thread per element walked all <= 2048 partials: 6 workgroups,
// 4.4 ms per training step in 34 launches.)
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* in, float* out, u32 n, u32 rows) {
  __shared__ float part[4][64];
  const u32 el = threadIdx.x & 63u, kq = threadIdx.x >> 6;
  const u32 i = blockIdx.x * 64u + el, r0 = blockIdx.y * 64u + kq * 16u;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (i < n && r0 + (u32)k < rows) ? in[(size_t)(r0 + (u32)k) * n + i] : 0.f;
  float s = v[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += v[k];
  part[kq][el] = s;
  __syncthreads();
  if (kq == 0 && i < n) out[(size_t)blockIdx.y * n + i] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
}

// -> true: the LDS-tiled kernel (128 x 128 tiles of dW), false: the streaming kernel (wave tiles of mfw x nfw fragments)
static bool pw_wgrad_plan(int B, int Cout, int Cin, int HW, PwgParams* p, int* mfw, int* nfw) {
  const int mf = (Cout + 15) / 16, nf = (Cin + 15) / 16;
  p->cpi = (u32)((HW + 127) / 128);
  p->chunks = (u32)B * p->cpi;
  static const int env_tiled = getenv("SSDK_PW_WGRAD_TILED") ? atoi(getenv("SSDK_PW_WGRAD_TILED")) : 1;
  const bool tiled = env_tiled && mf >= 4 && nf >= 4;
  u32 tiles, target;
  if (tiled) {
    *mfw = *nfw = 8;
    p->mt = (u32)((Cout + 127) / 128);
    p->nt = (u32)((Cin + 127) / 128);
    tiles = p->mt * p->nt;
    target = 512u;  // workgroups: two per CU
  } else {
    // wave tile: 3 x 2 fragments (20 vectors of 16 bytes in flight per lane), 3 x 1 for narrow inputs
    *mfw = mf >= 3 ? 3 : mf;
    *nfw = nf >= 2 ? 2 : 1;
    p->mt = (u32)((mf + *mfw - 1) / *mfw);
    p->nt = (u32)((nf + *nfw - 1) / *nfw);
    tiles = p->mt * p->nt;
    target = 4096u;  // waves: ~16 per CU
  }
  u32 splits = target / tiles;
  if (splits < 1u) splits = 1u;
  if (splits > p->chunks) splits = p->chunks;
  p->cps = (p->chunks + splits - 1u) / splits;
  p->splits = (p->chunks + p->cps - 1u) / p->cps;
  return tiled;
}

}  // namespace ssdk

using namespace ssdk;

static int pw_check(const void* a, const void* b, const void* c, int B, int K, int M, int HW, int dtype, const char* what) {
  if (!a || !b || !c || B < 1 || K < 8 || M < 1 || HW < 1 || (K % 8) || (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("%s: bad argument (16-bit tensors, input channels a multiple of 8)", what);
    return SSDK_E_BADARG;
  }
  if ((size_t)B * (size_t)(K > M ? K : M) * (size_t)HW >= ((size_t)1 << 32)) {
    set_error("%s: tensor too large", what);
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}

// sum of `rows` rows of n floats in a fixed order -> dst (<= 4096 rows: two passes through `mid`)
static int pw_reduce_rows(const float* src, float* mid, float* dst, u32 n, u32 rows, hipStream_t st) {
  while (true) {
    const u32 out_rows = (rows + 63u) / 64u;
    float* out = out_rows == 1u ? dst : mid;
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((n + 63u) / 64u, out_rows), dim3(256), 0, st, src, out, n, rows);
    if (out_rows == 1u) break;
    src = mid;  // (<= 4096 rows: the second pass is the last one; a third would need another buffer)
    rows = out_rows;
    if (rows > 64u) {
      set_error("ssdk_pw: more than 4096 partial rows");
      return SSDK_E_BADARG;
    }
  }
  return check_launch("pw_wgrad_reduce_kernel");
}

namespace ssdk {
int reduce_rows_fixed_order(const float* src, float* mid, float* dst, unsigned n, unsigned rows, hipStream_t st) {
  return pw_reduce_rows(src, mid, dst, n, rows, st);
}
}  // namespace ssdk

extern "C" int ssdk_pw_forward(const void* x, const void* a, const float* bias, void* y, int B, int K, int M, int HW, int dtype,
                               void* stream) {
  int rc = pw_check(x, a, y, B, K, M, HW, dtype, "ssdk_pw_forward");
  if (rc) return rc;
  if (((uintptr_t)a) & 15) {
    set_error("ssdk_pw_forward: the weight matrix must be 16-byte aligned");
    return SSDK_E_BADARG;
  }
  return pw_gemm_launch(x, a, bias, y, B, K, M, HW, dtype, (hipStream_t)stream);
}

// forward + the per-channel (sum y, sum y^2) of the stored outputs: what the BatchNorm behind the convolution needs
extern "C" size_t ssdk_pw_stats_workspace_bytes(int B, int K, int M, int HW) {
  if (B < 1 || K < 1 || M < 1 || HW < 1) return 0;
  PwtParams p;
  const PwtPlan pl = pw_gemm_plan(p, B, K, M, HW, true);
  const size_t rows = (size_t)pl.gx * 4;
  return (rows + (rows + 63) / 64) * (size_t)M * 2 * sizeof(float);
}

static int pw_reduce_rows(const float* src, float* mid, float* dst, u32 n, u32 rows, hipStream_t st);

extern "C" int ssdk_pw_forward_stats(const void* x, const void* a, const float* bias, void* y, float* sums, void* workspace,
                                     size_t workspace_bytes, int B, int K, int M, int HW, int dtype, void* stream) {
  int rc = pw_check(x, a, y, B, K, M, HW, dtype, "ssdk_pw_forward_stats");
  if (rc) return rc;
  if (!sums || !workspace || (((uintptr_t)a) & 15) || (((uintptr_t)workspace) & 15)) {
    set_error("ssdk_pw_forward_stats: null / misaligned pointer");
    return SSDK_E_BADARG;
  }
  if (workspace_bytes < ssdk_pw_stats_workspace_bytes(B, K, M, HW)) {
    set_error("ssdk_pw_forward_stats: workspace too small");
    return SSDK_E_WORKSPACE;
  }
  unsigned rows = 0;
  rc = pw_gemm_launch(x, a, bias, y, B, K, M, HW, dtype, (hipStream_t)stream, (float*)workspace, &rows);
  if (rc) return rc;
  const u32 n = (u32)M * 2u;
  return pw_reduce_rows((const float*)workspace, (float*)workspace + (size_t)rows * n, sums, n, rows, (hipStream_t)stream);
}

extern "C" size_t ssdk_pw_wgrad_workspace_bytes(int B, int Cout, int Cin, int HW) {
  if (B < 1 || Cout < 1 || Cin < 1 || HW < 1) return 0;
  PwgParams p;
  int mfw, nfw;
  pw_wgrad_plan(B, Cout, Cin, HW, &p, &mfw, &nfw);
  return ((size_t)p.splits + (size_t)((p.splits + 63u) / 64u)) * (size_t)Cout * (size_t)Cin * sizeof(float);
}

extern "C" int ssdk_pw_wgrad(const void* dy, const void* x, float* dw, void* ws, size_t ws_bytes, int B, int Cout, int Cin, int HW,
                             int dtype, void* stream) {
  if (!dy || !x || !dw || !ws || B < 1 || Cout < 1 || Cin < 1 || HW < 1 || (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("ssdk_pw_wgrad: bad argument");
    return SSDK_E_BADARG;
  }
  if ((size_t)B * (size_t)(Cout > Cin ? Cout : Cin) * (size_t)HW >= ((size_t)1 << 32)) {
    set_error("ssdk_pw_wgrad: tensor too large");
    return SSDK_E_BADARG;
  }
  PwgParams p;
  int mfw, nfw;
  const bool tiled = pw_wgrad_plan(B, Cout, Cin, HW, &p, &mfw, &nfw);
  if (ws_bytes < ((size_t)p.splits + (size_t)((p.splits + 63u) / 64u)) * (size_t)Cout * (size_t)Cin * sizeof(float)) {
    set_error("ssdk_pw_wgrad: workspace too small");
    return SSDK_E_WORKSPACE;
  }
  p.dy = (const u16*)dy;
  p.x = (const u16*)x;
  p.ws = (float*)ws;
  p.B = B;
  p.Cout = Cout;
  p.Cin = Cin;
  p.HW = HW;
  const u32 items = p.mt * p.nt * p.splits;
  const dim3 grid(tiled ? items : (items + 3u) / 4u);
  hipStream_t st = (hipStream_t)stream;
  if (tiled) {
    if (dtype == SSDK_BF16) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_tiled_kernel<SSDK_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PWG_LDS);
      hipLaunchKernelGGL((pw_wgrad_tiled_kernel<SSDK_BF16>), grid, dim3(PWT_THREADS), PWG_LDS, st, p);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_tiled_kernel<SSDK_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PWG_LDS);
      hipLaunchKernelGGL((pw_wgrad_tiled_kernel<SSDK_F16>), grid, dim3(PWT_THREADS), PWG_LDS, st, p);
    }
  } else {
#define SSDK_PWG(DT, MF_, NF_) hipLaunchKernelGGL((pw_wgrad_kernel<DT, MF_, NF_>), grid, dim3(PWT_THREADS), 0, st, p)
#define SSDK_PWGD(DT)                                  \
  do {                                                 \
    if (mfw == 3 && nfw == 2) SSDK_PWG(DT, 3, 2);      \
    else if (mfw == 3) SSDK_PWG(DT, 3, 1);             \
    else if (mfw == 2 && nfw == 2) SSDK_PWG(DT, 2, 2); \
    else if (mfw == 2) SSDK_PWG(DT, 2, 1);             \
    else if (nfw == 2) SSDK_PWG(DT, 1, 2);             \
    else SSDK_PWG(DT, 1, 1);                           \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_PWGD(SSDK_BF16);
  else SSDK_PWGD(SSDK_F16);
#undef SSDK_PWGD
#undef SSDK_PWG
  }
  int rc = check_launch(tiled ? "pw_wgrad_tiled_kernel" : "pw_wgrad_kernel");
  if (rc) return rc;
  const u32 n = (u32)Cout * (u32)Cin;
  return pw_reduce_rows((const float*)ws, (float*)ws + (size_t)p.splits * n, dw, n, p.splits, st);
}

// w32 [Cout, Cin] fp32 (the master weights) -> w16 [Cout, Cin] and wt16 [Cin, Cout] in the compute dtype: the cast autocast
// would launch anyway, plus the transposed copy the input gradient reads as ITS row-major matrix.
namespace ssdk {
template <int DT>
__global__ __launch_bounds__(256) void pw_prepare_kernel(const float* w32, u16* w16, u16* wt16, u32 cout, u32 cin) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= cout * cin) return;
  const u32 co = i / cin, ci = i - co * cin;
  const u16 v = (u16)f32_to_bits16<DT>(w32[i]);
  w16[i] = v;
  wt16[(size_t)ci * cout + co] = v;
}
}  // namespace ssdk

extern "C" int ssdk_pw_prepare(const float* w32, void* w16, void* wt16, int Cout, int Cin, int dtype, void* stream) {
  if (!w32 || !w16 || !wt16 || Cout < 1 || Cin < 1 || (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("ssdk_pw_prepare: bad argument");
    return SSDK_E_BADARG;
  }
  const u32 n = (u32)Cout * (u32)Cin;
  if (dtype == SSDK_BF16)
    hipLaunchKernelGGL(pw_prepare_kernel<SSDK_BF16>, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, w32, (u16*)w16,
                       (u16*)wt16, (u32)Cout, (u32)Cin);
  else
    hipLaunchKernelGGL(pw_prepare_kernel<SSDK_F16>, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, w32, (u16*)w16,
                       (u16*)wt16, (u32)Cout, (u32)Cin);
  return check_launch("pw_prepare_kernel");
}

// ---- 3x3 convolutions of the training step (stem, extras, multibox heads) on the same kernels -----------------------------------
// A dense 3x3 / pad 1 / stride 1 | 2 convolution in NCHW is y[b] = W2 col[b] with W2 = weight.view(Cout, Cin * 9) -- torch's own
// flattening, k = ci * 9 + ky * 3 + kx -- and col[b][k][p] = x[b][ci][s oy - 1 + ky][s ox - 1 + kx] (zero outside the plane).  On a
// chip with 8 TB/s the nine-fold copy is cheap next to what it buys: forward, input gradient and weight gradient of every 3x3
// layer ARE the 1x1 kernels above (ssdk_pw_forward with a = W2 / W2^T, ssdk_pw_wgrad against col), the weight gradient comes out
// in the parameter's own layout, and the only new code is two index kernels:
//   im2col3x3_kernel   x [B, C, H, W] -> col [B, Kp, Ho Wo], Kp = C * 9 rounded up to 8 (zero rows; the 3-channel stem: 27 -> 32)
//   col2im3x3_kernel   dcol [B, Kp, Ho Wo] -> dx [B, C, H, W]: every input pixel GATHERS its <= 9 contributions (fp32 sum in tap
//                      order, one rounding): no atomics, bit-reproducible
// Reference: the 3x3 convolutions of ssd.py:77-104 (extras through basic_layers.py:40-57, heads ssd.py:100-103) and the stem of
// mobilenet.py:78 in the step of pipeline_anchor_apex.py:103-130, which PyTorch-ROCm sends to MIOpen's igemm_{fwd,bwd,wrw}_gtcx35
// kernels between batched_transpose launches.
namespace ssdk {

struct ColParams {
  const u16* x;
  u16* col;
  int B, C, H, W, Ho, Wo, stride, Kp;
  size_t kstride, bstride;  // elements between two k rows of an image / between two images: [B][Kp][HWo], or FOLDED [Kp][B][HWo]
};

__global__ __launch_bounds__(256) void im2col3x3_kernel(const ColParams p) {
  const u32 HWo = (u32)(p.Ho * p.Wo), vpr = (HWo + 7u) / 8u;  // 8-pixel vectors per col row
  const size_t total = (size_t)p.B * p.Kp * vpr;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
    const u32 v = (u32)(i % vpr), k = (u32)((i / vpr) % (u32)p.Kp), b = (u32)(i / ((size_t)vpr * p.Kp));
    const u32 p0 = v * 8u;
    const int nvalid = (int)HWo - (int)p0;
    u16* dst = p.col + (size_t)b * p.bstride + (size_t)k * p.kstride + p0;
    u32 h[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (k < (u32)(p.C * 9)) {
      const u32 c = k / 9u, t = k - c * 9u, ky = t / 3u, kx = t - ky * 3u;
      const u16* plane = p.x + ((size_t)b * p.C + c) * p.H * p.W;
      const u32 oy0 = p0 / (u32)p.Wo, ox0 = p0 - oy0 * (u32)p.Wo;
      const int iy0 = (int)oy0 * p.stride - 1 + (int)ky, ix0 = (int)ox0 * p.stride - 1 + (int)kx;
      if (p.stride == 1 && ox0 + 7u < (u32)p.Wo && nvalid >= 8 && ix0 >= 0 && ix0 + 7 < p.W) {  // one output row, interior columns
        if ((unsigned)iy0 < (unsigned)p.H) {
          *reinterpret_cast<pw_u32x4_a2*>(dst) = *reinterpret_cast<const pw_u32x4_a2*>(plane + (size_t)iy0 * p.W + ix0);
          continue;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const u32 op = p0 + (u32)e, oy = op / (u32)p.Wo, ox = op - oy * (u32)p.Wo;
          const int iy = (int)oy * p.stride - 1 + (int)ky, ix = (int)ox * p.stride - 1 + (int)kx;
          if (e < nvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) h[e] = (u32)plane[(size_t)iy * p.W + ix];
        }
      }
    }
    pw_store8<true>(dst, u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)}, nvalid);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const ColParams p) {  // p.col = dcol (read), p.x = dx (written)
  const u32 HW = (u32)(p.H * p.W), vpr = (HW + 7u) / 8u;
  const size_t total = (size_t)p.B * p.C * vpr;
  u16* dx = const_cast<u16*>(p.x);
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
    const u32 v = (u32)(i % vpr), c = (u32)((i / vpr) % (u32)p.C), b = (u32)(i / ((size_t)vpr * p.C));
    const u32 p0 = v * 8u;
    const int nvalid = (int)HW - (int)p0;
    const u16* rows = p.col + (size_t)b * p.bstride + (size_t)c * 9u * p.kstride;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32 iy0 = p0 / (u32)p.W, ix0 = p0 - iy0 * (u32)p.W;
    const bool one_row = p.stride == 1 && ix0 + 7u < (u32)p.W && nvalid >= 8;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ky = t / 3, kx = t % 3;
      const u16* row = rows + (size_t)t * p.kstride;
      if (one_row) {  // stride 1: the 8 input pixels of a row meet 8 consecutive output pixels of row iy + 1 - ky
        const int oy = (int)iy0 + 1 - ky, ox = (int)ix0 + 1 - kx;
        if ((unsigned)oy >= (unsigned)p.Ho) continue;
        if (ox >= 0 && ox + 7 < p.Wo) {
          const u32x4 d = *reinterpret_cast<const pw_u32x4_a2*>(row + (size_t)oy * p.Wo + ox);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += bits16_to_f32<DT>((e & 1) ? (d[e >> 1] >> 16) : (d[e >> 1] & 0xffffu));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if ((unsigned)(ox + e) < (unsigned)p.Wo) acc[e] += bits16_to_f32<DT>((u32)row[(size_t)oy * p.Wo + ox + e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const u32 ip = p0 + (u32)e, iy = ip / (u32)p.W, ix = ip - iy * (u32)p.W;
          const int ny = (int)iy + 1 - ky, nx = (int)ix + 1 - kx;  // = stride * (oy, ox)
          if (e >= nvalid || ny < 0 || nx < 0) continue;
          if (p.stride == 2 && ((ny | nx) & 1)) continue;
          const int oy = ny / p.stride, ox = nx / p.stride;
          if (oy < p.Ho && ox < p.Wo) acc[e] += bits16_to_f32<DT>((u32)row[(size_t)oy * p.Wo + ox]);
        }
      }
    }
    const u32x4 o = {pack2_16<DT>(acc[0], acc[1]), pack2_16<DT>(acc[2], acc[3]), pack2_16<DT>(acc[4], acc[5]), pack2_16<DT>(acc[6], acc[7])};
    pw_store8<true>(dx + ((size_t)b * p.C + c) * HW + p0, o, nvalid);
  }
}

}  // namespace ssdk

static int col_check(const void* a, const void* b, int B, int C, int H, int W, int stride, int dtype, const char* what) {
  if (!a || !b || B < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2) || (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("%s: bad argument (16-bit NCHW tensors, stride 1 | 2)", what);
    return SSDK_E_BADARG;
  }
  const size_t kp = ((size_t)C * 9 + 7) / 8 * 8;
  if ((size_t)B * kp * (size_t)H * (size_t)W >= ((size_t)1 << 32)) {
    set_error("%s: tensor too large", what);
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}

static ColParams col_params(const void* x, void* col, int B, int C, int H, int W, int stride, int fold = 0) {
  ColParams p;
  p.x = (const u16*)x;
  p.col = (u16*)col;
  p.B = B;
  p.C = C;
  p.H = H;
  p.W = W;
  p.stride = stride;
  p.Ho = (H + 2 - 3) / stride + 1;
  p.Wo = (W + 2 - 3) / stride + 1;
  p.Kp = (C * 9 + 7) / 8 * 8;
  const size_t hwo = (size_t)p.Ho * p.Wo;
  p.kstride = fold ? (size_t)B * hwo : hwo;
  p.bstride = fold ? hwo : (size_t)p.Kp * hwo;
  return p;
}

extern "C" int ssdk_im2col3x3(const void* x, void* col, int B, int C, int H, int W, int stride, int dtype, void* stream) {
  int rc = col_check(x, col, B, C, H, W, stride, dtype, "ssdk_im2col3x3");
  if (rc) return rc;
  const ColParams p = col_params(x, col, B, C, H, W, stride);
  const size_t total = (size_t)B * p.Kp * (((size_t)p.Ho * p.Wo + 7) / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("im2col3x3_kernel");
}

extern "C" int ssdk_col2im3x3(const void* dcol, void* dx, int B, int C, int H, int W, int stride, int dtype, void* stream) {
  int rc = col_check(dcol, dx, B, C, H, W, stride, dtype, "ssdk_col2im3x3");
  if (rc) return rc;
  ColParams p = col_params(dx, const_cast<void*>(dcol), B, C, H, W, stride);
  const size_t total = (size_t)B * C * (((size_t)H * W + 7) / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (dtype == SSDK_BF16) hipLaunchKernelGGL(col2im3x3_kernel<SSDK_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(col2im3x3_kernel<SSDK_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("col2im3x3_kernel");
}

// The same two with the batch FOLDED into the pixel dimension: col / dcol are [Kp][B][Ho * Wo] = one "image" of B * Ho * Wo pixels
// for the 1x1 kernels, which tile the pixels of ONE image in groups of 128 -- layers with a few pixels per image (the SSD extras:
// 64 / 16 / 4 / 1) would fill those groups to 50 ... 0.8 %.
extern "C" int ssdk_im2col3x3_folded(const void* x, void* col, int B, int C, int H, int W, int stride, int dtype, void* stream) {
  int rc = col_check(x, col, B, C, H, W, stride, dtype, "ssdk_im2col3x3_folded");
  if (rc) return rc;
  const ColParams p = col_params(x, col, B, C, H, W, stride, 1);
  const size_t total = (size_t)B * p.Kp * (((size_t)p.Ho * p.Wo + 7) / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("im2col3x3_kernel");
}

extern "C" int ssdk_col2im3x3_folded(const void* dcol, void* dx, int B, int C, int H, int W, int stride, int dtype, void* stream) {
  int rc = col_check(dcol, dx, B, C, H, W, stride, dtype, "ssdk_col2im3x3_folded");
  if (rc) return rc;
  ColParams p = col_params(dx, const_cast<void*>(dcol), B, C, H, W, stride, 1);
  const size_t total = (size_t)B * C * (((size_t)H * W + 7) / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (dtype == SSDK_BF16) hipLaunchKernelGGL(col2im3x3_kernel<SSDK_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(col2im3x3_kernel<SSDK_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("col2im3x3_kernel");
}

// ---- weights of a 3x3 layer (or of the loc | conf PAIR of an SSD level) from the fp32 master tensors into the inference kernels'
//      layouts, one launch per step: KRSC 16-bit rows [n1 + n2][9 * Cin] (k = (ky*3 + kx)*Cin + ci), the fragment-major image
//      [ceil(rows / 16)][K / 32][4][16][8] of ssdk_weight_frag_bytes (zero rows past the last channel), and the fp32 biases
//      (ssds/modeling/layers/headconv.py: the head convolutions of the training step run on conv3x3_short / halo / smallmap) ----
namespace ssdk {
struct PackParams {
  const float *w1, *w2, *b1, *b2;
  unsigned short *krsc, *frag;
  float* bias;
  int n1, n2, cin, rows_pad, dt;
};
template <int DT>
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const PackParams p) {
  const u32 K = 9u * (u32)p.cin, kch = K / 8u;
  const u32 rows = (u32)(p.n1 + p.n2);
  const u32 total = (u32)p.rows_pad * kch;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const u32 row = i / kch, k0 = (i - row * kch) * 8u;
    const u32 tap = k0 / (u32)p.cin, ci0 = k0 - tap * (u32)p.cin;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < rows) {
      const float* src = row < (u32)p.n1 ? p.w1 + ((size_t)row * p.cin + ci0) * 9 + tap
                                        : p.w2 + ((size_t)(row - (u32)p.n1) * p.cin + ci0) * 9 + tap;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pack2_16<DT>(src[(2 * e) * 9], src[(2 * e + 1) * 9]);
      *reinterpret_cast<u32x4*>(p.krsc + (size_t)row * K + k0) = v;
    }
    if (p.frag) *reinterpret_cast<u32x4*>(p.frag + ((((size_t)(row >> 4) * (K / 32u) + (k0 >> 5)) * 4u + ((k0 >> 3) & 3u)) * 16u + (row & 15u)) * 8u) = v;
    if (k0 == 0 && row < rows) p.bias[row] = row < (u32)p.n1 ? (p.b1 ? p.b1[row] : 0.f) : (p.b2 ? p.b2[row - (u32)p.n1] : 0.f);
  }
}
}  // namespace ssdk

extern "C" int ssdk_pack_conv3x3(const float* w1, const float* b1, int n1, const float* w2, const float* b2, int n2, int cin, void* krsc,
                                 void* frag, float* bias, int dtype, void* stream) {
  using namespace ssdk;
  if (!w1 || !krsc || !bias || n1 <= 0 || n2 < 0 || (n2 > 0 && !w2) || cin <= 0 || (cin % 8)) {
    set_error("ssdk_pack_conv3x3: bad arguments (Cin must be a multiple of 8)");
    return SSDK_E_BADARG;
  }
  if (dtype != SSDK_BF16 && dtype != SSDK_F16) {
    set_error("ssdk_pack_conv3x3: 16-bit weights only");
    return SSDK_E_BADARG;
  }
  if (frag && ((9 * cin) % 32)) {
    set_error("ssdk_pack_conv3x3: a fragment-major image needs 9 * Cin to be a multiple of 32");
    return SSDK_E_BADARG;
  }
  PackParams p;
  p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2;
  p.krsc = (unsigned short*)krsc; p.frag = (unsigned short*)frag; p.bias = bias;
  p.n1 = n1; p.n2 = n2; p.cin = cin; p.dt = dtype;
  p.rows_pad = frag ? (n1 + n2 + 15) / 16 * 16 : n1 + n2;
  const size_t total = (size_t)p.rows_pad * (9 * cin / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == SSDK_BF16) hipLaunchKernelGGL(pack_conv3x3_kernel<SSDK_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pack_conv3x3_kernel<SSDK_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("pack_conv3x3_kernel");
}

// ---- the same pair as the weights of its INPUT-GRADIENT convolution: dx = conv3x3(dy, W') with W'[ci][ky][kx][o] = W[o][ci][2 - ky][2 - kx]
//      (stride 1, pad 1), o zero-padded to opad channels (dy is handed over with opad channels: 504 -> 512 puts the small levels
//      on conv_smallmap_kernel); KRSC rows [Cin][9 * opad] + the fragment-major image ----
namespace ssdk {
struct PackDgradParams {
  const float *w1, *w2;
  unsigned short *krsc, *frag;
  int n1, n2, cin, opad, rows_pad;
};
template <int DT>
__global__ __launch_bounds__(256) void pack_conv3x3_dgrad_kernel(const PackDgradParams p) {
  const u32 K = 9u * (u32)p.opad, kch = K / 8u;
  const u32 total = (u32)p.rows_pad * kch;
  const u32 n12 = (u32)(p.n1 + p.n2);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const u32 ci = i / kch, k0 = (i - ci * kch) * 8u;
    const u32 tap = k0 / (u32)p.opad, o0 = k0 - tap * (u32)p.opad;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ci < (u32)p.cin) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const u32 o = o0 + (u32)e;
        f[e] = o < (u32)p.n1 ? p.w1[((size_t)o * p.cin + ci) * 9 + (8u - tap)]
                             : (o < n12 ? p.w2[((size_t)(o - (u32)p.n1) * p.cin + ci) * 9 + (8u - tap)] : 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pack2_16<DT>(f[2 * e], f[2 * e + 1]);
      *reinterpret_cast<u32x4*>(p.krsc + (size_t)ci * K + k0) = v;
    }
    if (p.frag) *reinterpret_cast<u32x4*>(p.frag + ((((size_t)(ci >> 4) * (K / 32u) + (k0 >> 5)) * 4u + ((k0 >> 3) & 3u)) * 16u + (ci & 15u)) * 8u) = v;
  }
}
}  // namespace ssdk

extern "C" int ssdk_pack_conv3x3_dgrad(const float* w1, int n1, const float* w2, int n2, int cin, int opad, void* krsc, void* frag, int dtype,
                                       void* stream) {
  using namespace ssdk;
  if (!w1 || !krsc || n1 <= 0 || n2 < 0 || (n2 > 0 && !w2) || cin <= 0 || opad < n1 + n2 || (opad % 8)) {
    set_error("ssdk_pack_conv3x3_dgrad: bad arguments (opad >= n1 + n2, a multiple of 8)");
    return SSDK_E_BADARG;
  }
  if (dtype != SSDK_BF16 && dtype != SSDK_F16) {
    set_error("ssdk_pack_conv3x3_dgrad: 16-bit weights only");
    return SSDK_E_BADARG;
  }
  if (frag && (opad % 32)) {
    set_error("ssdk_pack_conv3x3_dgrad: a fragment-major image needs opad to be a multiple of 32");
    return SSDK_E_BADARG;
  }
  PackDgradParams p;
  p.w1 = w1; p.w2 = w2;
  p.krsc = (unsigned short*)krsc; p.frag = (unsigned short*)frag;
  p.n1 = n1; p.n2 = n2; p.cin = cin; p.opad = opad;
  p.rows_pad = frag ? (cin + 15) / 16 * 16 : cin;
  const size_t total = (size_t)p.rows_pad * (9 * opad / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == SSDK_BF16) hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel<SSDK_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel<SSDK_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("pack_conv3x3_dgrad_kernel");
}

// ---- [N, c1, HW] | [N, c2, HW] (NCHW, 16 bit) -> [N, HW, cpad] (NHWC), channels c1 + c2 .. cpad - 1 zero: the concatenated output
//      gradient of an SSD level's loc | conf pair in the layout the inference kernels read (headconv._input_gradient; the framework's
//      strided copy_ into a channels_last buffer ran at 0.3 TB/s: 0.31 ms per step for 180 MB).  64 x 64 tiles through LDS. ----
namespace ssdk {
struct CatParams {
  const unsigned short *a, *b;
  unsigned short* out;
  int n, c1, c2, cpad, hw;
  int ptiles, ctiles;
};
__global__ __launch_bounds__(256) void concat_nchw_to_nhwc_kernel(const CatParams p) {
  __shared__ unsigned short tile[64][66];
  const int per = p.ptiles * p.ctiles;
  for (int t = blockIdx.x; t < p.n * per; t += gridDim.x) {
    const int img = t / per, r = t - img * per, ct = r / p.ptiles, pt = r - ct * p.ptiles;
    const int c0 = ct * 64, p0 = pt * 64;
    // read: thread (cl = tid / 4, four 16-pixel pieces per channel row)
    {
      const int cl = threadIdx.x >> 2, q = (threadIdx.x & 3) * 16;
      const int c = c0 + cl;
      const unsigned short* src = c < p.c1 ? p.a + ((size_t)img * p.c1 + c) * p.hw : (c < p.c1 + p.c2 ? p.b + ((size_t)img * p.c2 + (c - p.c1)) * p.hw : nullptr);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = p0 + q + e;
        tile[cl][q + e] = (src && px < p.hw) ? src[px] : (unsigned short)0;
      }
    }
    __syncthreads();
    // write: thread (pl = tid / 4, four 16-channel pieces per pixel)
    {
      const int pl = threadIdx.x >> 2, q = (threadIdx.x & 3) * 16;
      const int px = p0 + pl;
      if (px < p.hw) {
        unsigned short* dst = p.out + ((size_t)img * p.hw + px) * p.cpad + c0 + q;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          if (c0 + q + e < p.cpad)  // (cpad is even)
            *reinterpret_cast<u32*>(dst + e) = (u32)tile[q + e][pl] | ((u32)tile[q + e + 1][pl] << 16);
        }
      }
    }
    __syncthreads();
  }
}
}  // namespace ssdk

extern "C" int ssdk_concat_nchw_to_nhwc(const void* a, int c1, const void* b, int c2, void* out, int cpad, int N, int HW, int dtype,
                                        void* stream) {
  using namespace ssdk;
  if (!a || !out || N < 1 || HW < 1 || c1 < 1 || c2 < 0 || (c2 > 0 && !b) || cpad < c1 + c2 || (cpad & 1) ||
      (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("ssdk_concat_nchw_to_nhwc: bad arguments (16-bit tensors, cpad >= c1 + c2 and even)");
    return SSDK_E_BADARG;
  }
  CatParams p;
  p.a = (const unsigned short*)a; p.b = (const unsigned short*)b; p.out = (unsigned short*)out;
  p.n = N; p.c1 = c1; p.c2 = c2; p.cpad = cpad; p.hw = HW;
  p.ptiles = (HW + 63) / 64;
  p.ctiles = (cpad + 63) / 64;
  long tiles = (long)N * p.ptiles * p.ctiles;
  if (tiles > 2000000000l) {
    set_error("ssdk_concat_nchw_to_nhwc: tensor too large");
    return SSDK_E_BADARG;
  }
  const unsigned grid = (unsigned)(tiles < 256 * 16 ? tiles : 256 * 16);
  hipLaunchKernelGGL(concat_nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("concat_nchw_to_nhwc_kernel");
}
