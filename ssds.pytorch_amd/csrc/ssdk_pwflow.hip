// ssdk_pwflow.hip -- 1x1 convolution (+ folded BN, activation, residual) with a SHORT K (Cin <= 256) as a STREAMING kernel:
// the bottleneck 1x1 layers of the ResNet / RegNet backbones behind the FPN / BiFPN configurations (reference
// nets/resnet.py:41-56 through torchvision's Bottleneck, nets/regnet.py:139-186), e.g. 64 -> 256 on 160x160 maps at batch 32.
//
// Why: with K = 64 ... 256 these layers are HBM-bound -- 64 -> 256 @160x160 reads 105 MB, adds a 419 MB residual and writes
// 419 MB for 27 GFLOP -- and on the tiled implicit-GEMM kernels (conv_gemm_kernel: 128 x BN x 32 tiles staged through LDS,
// one or two k-steps, a transposed 128-row store per tile) they ran at 1.2 - 3.9 TB/s: 37 such launches were 42 % of the
// FPN-ResNet50@640 step (round-3 review).  Nothing needs staging here:
//   * a WAVE owns 16 consecutive pixels at a time (a "group") and ALL output channels of the workgroup's slice: the B operand
//     of v_mfma_f32_16x16x32 is 16 bytes = 8 input channels of one pixel per lane, read straight from global memory (each
//     pixel's channel row is contiguous in NHWC: whole 64-byte segments), the next group's rows are requested before this
//     group's MFMAs;
//   * the weights of the slice (<= 64 KiB) sit in LDS as A fragments, staged once per workgroup (4 waves; several workgroups
//     per CU), every ds_read_b128 a broadcast-free 1 KiB read;
//   * output channels are PERMUTED onto fragment rows so that a lane ends up with 8 CONSECUTIVE channels of its pixel in two
//     accumulator fragments (fragment 2h + e, row 4g + j  <->  channel 32h + 8g + 4e + j): scale / bias / activation / residual
//     and the store work on 16-byte pieces -- four lanes write one 64-byte line segment -- without a transpose through LDS;
//   * residual modes of ssdk_conv: same-resolution or half-resolution (nearest x2, FPN top-down) residual, activation before
//     or after the add (ResNet block tails); stride 1 or 2 (a stride-2 1x1 reads every second pixel of every second row).
// HBM bytes per group: 16 * (Cin + Cout [+ Cout]) * 2; the kernel's roofline is the layer's input + output (+ residual).
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int PW_THREADS = 256;

struct PwParams {
  ConvParams c;
  int slices;        // Cout / (16 * NF)
  unsigned groups;   // ceil(M / 16)
  unsigned gstride;  // groups advanced per trip = gridDim.x * 4
};

// LDS: [NF][KS][64 lanes] u32x4 weights | [NF / 2][4 fg][8] scale | [NF / 2][4 fg][8] bias
template <int KS, int NF>
struct PwLds {
  static constexpr int w = 0;
  static constexpr int sc = NF * KS * 1024;
  static constexpr int bi = sc + (NF / 2) * 4 * 32;
  static constexpr int bytes = bi + (NF / 2) * 4 * 32;
};

template <int DT, int KS, int NF>
__global__ __launch_bounds__(PW_THREADS, 2) void pwflow_kernel(const PwParams pp) {
  using L = PwLds<KS, NF>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvParams& p = pp.c;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int Cin = p.Cin, Cout = p.Cout;
  const u32 slice0 = blockIdx.y * (u32)(16 * NF);  // first output channel of this workgroup's slice

  // ---- stage the slice's weights as A fragments with the channel permutation, and its scale / bias in lane order ----------
  for (u32 i = tid; i < (u32)(NF * KS * 64); i += PW_THREADS) {
    const u32 l = i & 63u, fk = i >> 6, ks = fk % (u32)KS, f = fk / (u32)KS;
    const u32 row = l & 15u, g = row >> 2, j = row & 3u;
    const u32 co = slice0 + 32u * (f >> 1) + 8u * g + 4u * (f & 1u) + j;
    const u32 k0 = ks * 32u + (l >> 4) * 8u;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < (u32)Cout && k0 < (u32)Cin) v = *reinterpret_cast<const u32x4*>((const u16*)p.w + (size_t)co * Cin + k0);
    *reinterpret_cast<u32x4*>(smem + L::w + i * 16) = v;
  }
  for (u32 i = tid; i < (u32)((NF / 2) * 4 * 8); i += PW_THREADS) {  // [h][g][8 channels]
    const u32 co = slice0 + 32u * (i >> 5) + (i & 31u);
    float s = 1.f, b = 0.f;
    if (co < (u32)Cout) {
      if (p.scale) s = p.scale[co];
      b = p.bias[co];
    }
    reinterpret_cast<float*>(smem + L::sc)[i] = s;
    reinterpret_cast<float*>(smem + L::bi)[i] = b;
  }
  __syncthreads();

  const u16* x = (const u16*)p.x;
  const u16* res = (const u16*)p.res;
  u16* y = (u16*)p.y;
  const u32 M = (u32)p.M, HWo = (u32)(p.Ho * p.Wo), Wo = (u32)p.Wo;
  const bool strided = p.stride != 1;
  const bool half_res = (p.res_mode & 1) != 0;
  const int post = p.post;
  const ActSel as = act_sel(p.act);
  const bool clampy = act_is_clamp(p.act);

  // element offset of this lane's input pixel / residual pixel for output pixel m (m >= M: clamped, never stored)
  auto in_off = [&](u32 m) -> size_t {
    if (!strided) return (size_t)m * (u32)Cin;
    const u32 b = m / HWo, r = m - b * HWo, oy = r / Wo, ox = r - oy * Wo;
    return (((size_t)b * (u32)p.H + oy * 2u) * (u32)p.W + ox * 2u) * (u32)Cin;
  };
  auto res_off = [&](u32 m) -> size_t {
    if (!half_res) return (size_t)m * (u32)Cout;
    const u32 b = m / HWo, r = m - b * HWo, oy = r / Wo, ox = r - oy * Wo;
    return (((size_t)b * (u32)(p.Ho >> 1) + (oy >> 1)) * (u32)(p.Wo >> 1) + (ox >> 1)) * (u32)Cout;
  };
  struct XG {
    u32x4 k[KS];
  };
  auto load_x = [&](u32 g) -> XG {
    XG o;
    u32 m = g * 16u + fr;
    m = m < M ? m : M - 1u;
    const u16* px = x + in_off(m) + fg * 8u;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      o.k[ks] = u32x4{0u, 0u, 0u, 0u};
      if ((u32)(ks * 32) + fg * 8u < (u32)Cin) o.k[ks] = *reinterpret_cast<const u32x4*>(px + ks * 32);
    }
    return o;
  };

  u32 g = blockIdx.x * 4u + wave;
  if (g >= pp.groups) return;
  XG xc = load_x(g);
  for (; g < pp.groups; g += pp.gstride) {
    const u32 gn = g + pp.gstride;
    XG xn;
    if (gn < pp.groups) xn = load_x(gn);  // (wave-uniform) the next group's rows travel under this group's work
    else xn = xc;
    const u32 m = g * 16u + fr;
    const bool live = m < M;
    // residual pieces of this lane's pixel: requested now, used in the epilogue
    u32x4 rv[NF / 2];
    if (res) {
      const u16* rp = res + res_off(live ? m : M - 1u) + slice0 + fg * 8u;
#pragma unroll
      for (int h = 0; h < NF / 2; ++h) rv[h] = *reinterpret_cast<const u32x4*>(rp + 32 * h);
    }
    // channel pairs one after the other: the two fragments of pair h + 1 go through the matrix cores while the VALU finishes
    // pair h (scale, bias, activation, residual, 16-byte store) -- two accumulator sets instead of NF, and the weight / constant
    // reads of a pair stay next to their use (left alone, the scheduler hoists all NF * KS fragment reads: 600 B of spills)
    auto pair_mfma = [&](int h, f32x4 (&acc)[2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4 wa = *reinterpret_cast<const u32x4*>(smem + L::w + (((2 * h + e) * KS + ks) * 64 + (int)lane) * 16);
          acc[e] = mfma16<DT>(wa, xc.k[ks], acc[e]);  // D[row = 4fg + r of fragment 2h + e][pixel fr]
        }
      }
    };
    u16* yp = y + (size_t)m * (u32)Cout + slice0 + fg * 8u;
    f32x4 acc_c[2], acc_n[2];
    pair_mfma(0, acc_c);
#pragma unroll
    for (int h = 0; h < NF / 2; ++h) {
      asm volatile("" ::: "memory");
      if (h + 1 < NF / 2) pair_mfma(h + 1, acc_n);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(smem + L::sc + ((h * 4 + (int)fg) * 8) * 4);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(smem + L::sc + ((h * 4 + (int)fg) * 8 + 4) * 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + L::bi + ((h * 4 + (int)fg) * 8) * 4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + L::bi + ((h * 4 + (int)fg) * 8 + 4) * 4);
      __builtin_amdgcn_sched_barrier(0);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc_c[0][r] * s0[r] + b0[r];
        v[4 + r] = acc_c[1][r] * s1[r] + b1[r];
      }
      if (as.mode != 0) {  // (workgroup-uniform) sigmoid / silu
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
          v[r] = as.mode == 1 ? sg : v[r] * sg;
        }
      }
      if (clampy) {  // (workgroup-uniform; a linear layer keeps its NaNs: v_min / v_max would drop them)
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], as.lo), as.hi);
      }
      u32x4 o = {pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]), pack2_16<DT>(v[4], v[5]), pack2_16<DT>(v[6], v[7])};
      if (res) {  // the conv result is rounded to the model dtype first, then the residual is added (torch's tensor add)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const u32 a = o[e], r = rv[h][e];
          const float lo = post_act(bits16_to_f32<DT>(a & 0xffffu) + bits16_to_f32<DT>(r & 0xffffu), post);
          const float hi = post_act(bits16_to_f32<DT>(a >> 16) + bits16_to_f32<DT>(r >> 16), post);
          o[e] = pack2_16<DT>(lo, hi);
        }
      }
      if (live && slice0 + 32u * (u32)h + fg * 8u < (u32)Cout) *reinterpret_cast<u32x4*>(yp + 32 * h) = o;
      acc_c[0] = acc_n[0];
      acc_c[1] = acc_n[1];
      __builtin_amdgcn_sched_barrier(0);
    }
    xc = xn;
  }
}

// 1: not one of this kernel's layers (the caller goes on), 0: launched
int launch_conv_pwflow(const ConvParams& p, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_PWFLOW") ? atoi(getenv("SSDK_PWFLOW")) : 1;
  if (!env || p.k != 1 || p.pad != 0 || (p.stride != 1 && p.stride != 2)) return 1;
  if (p.in_layout != LAYOUT_NHWC || p.out_layout != LAYOUT_NHWC || p.split != p.Cout || p.ksplits != 1) return 1;
  // short K, maps worth streaming.  Cin <= 128 (measured on FPN-ResNet50@640, batch 32, against conv_gemm_kernel: 64 -> 256
  // @160x160 240 -> 163 us, 128 -> 512 @80x80 160 -> 141 us, 64 -> 64 59 -> 50 us; with Cin = 256 the slice's weights take 64 KiB
  // for 128 channels -- two workgroups per CU, eight slices re-reading the input -- and nothing is gained: 256 -> 1024 @40x40
  // 98 -> 100 us, 256 -> 64 @160x160 117 -> 118 us; SSDK_PWFLOW=2 admits them again)
  if ((p.Cin % 32) || p.Cin > (env == 2 ? 256 : 128) || (p.Cout % 64) || p.M < 16384) return 1;
  if ((p.res_mode & 1) && ((p.Ho | p.Wo) & 1)) return 1;
  if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y | (uintptr_t)p.res) & 15) return 1;
  const int ks = p.Cin / 32;
  const int ksi = ks <= 2 ? 2 : (ks <= 4 ? 4 : 8);
  // output channels per workgroup: the slice's weights (<= 64 KiB) must leave room for two workgroups per CU
  int nf = ksi == 8 ? 8 : 16;
  while (nf > 4 && (p.Cout % (16 * nf))) nf >>= 1;
  if (p.Cout % (16 * nf)) return 1;
  PwParams pp;
  pp.c = p;
  pp.slices = p.Cout / (16 * nf);
  pp.groups = (unsigned)((p.M + 15) / 16);
  // enough workgroups for ~3 waves per SIMD, each wave walking several groups so that the staged weights amortise
  unsigned gx = 256u * 3u / (unsigned)pp.slices;
  if (gx < 64u) gx = 64u;
  const unsigned need = (pp.groups + 3u) / 4u;
  if (gx > need) gx = need;
  pp.gstride = gx * 4u;
  const dim3 grid(gx, (unsigned)pp.slices);
#define SSDK_PW(DT, KS_, NF_)                                                                                       \
  do {                                                                                                              \
    constexpr int lds = PwLds<KS_, NF_>::bytes;                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pwflow_kernel<DT, KS_, NF_>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                     \
    hipLaunchKernelGGL((pwflow_kernel<DT, KS_, NF_>), grid, dim3(PW_THREADS), lds, stream, pp);                     \
  } while (0)
#define SSDK_PWK(DT, KS_)                 \
  do {                                    \
    if (nf == 16) {                       \
      if constexpr (KS_ <= 4) SSDK_PW(DT, KS_, 16); \
    } else if (nf == 8) SSDK_PW(DT, KS_, 8);        \
    else SSDK_PW(DT, KS_, 4);             \
  } while (0)
#define SSDK_PWD(DT)                      \
  do {                                    \
    if (ksi == 2) SSDK_PWK(DT, 2);        \
    else if (ksi == 4) SSDK_PWK(DT, 4);   \
    else SSDK_PWK(DT, 8);                 \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_PWD(SSDK_BF16);
  else SSDK_PWD(SSDK_F16);
#undef SSDK_PWD
#undef SSDK_PWK
#undef SSDK_PW
  return 0;
}

}  // namespace ssdk
