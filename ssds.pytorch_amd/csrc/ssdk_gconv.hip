// ssdk_gconv.hip -- grouped 3x3 convolution with 16 channels per group (+ folded BN + activation) on gfx950.
//
// Reference: the RegNetX bottleneck transform (ssds/modeling/nets/regnet.py: `b = Conv2d(w_b, w_b, 3, stride,
// groups = w_b / 16)`), the only convolution of BASELINE config 5's backbone that is neither dense nor depthwise.
// One group is a 16-pixel x 16-channel x (9 taps x 16 channels) contraction = exactly one MFMA column block, so a
// WAVE owns one group and a run of pixels: the group's weights (16 x 144, padded to 5 k-steps of 32) stay in VGPRs
// in fragment layout, every lane loads its own B fragment (8 channels of one tap of one pixel, 16 bytes) straight
// from L2/L1 -- the 4 waves of a workgroup take 4 neighbouring groups of the SAME pixels, i.e. together whole
// 128-byte lines -- and D = W * X^T leaves each lane with 4 consecutive output channels of one pixel (8-byte store).
// No LDS, no barrier.  HBM-bound by design (input read ~once through L2, output once); FLOPs are tiny.
#include "ssdk_conv_common.h"

namespace ssdk {

struct GconvParams {
  const u16* x;
  const u16* w;  // [C][3][3][16]
  const float* scale;
  const float* bias;
  u16* y;
  int N, H, W, C, stride, Ho, Wo, act;
  int groups, mblocks;  // C / 16, ceil(N*Ho*Wo / (16*GC_FRAGS))
  long M;
};

constexpr int GC_FRAGS = 8;  // 16-pixel fragments per wave

template <int DT, int S>
__global__ __launch_bounds__(256) void gconv3x3_g16_kernel(const GconvParams p) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const u32 gblocks = ((u32)p.groups + 3u) / 4u;
  const u32 g = (blockIdx.x % gblocks) * 4u + wave;
  const u32 mb = blockIdx.x / gblocks;
  if (g >= (u32)p.groups) return;  // wave-uniform
  const int C = p.C;

  // weights of the group in registers: wf[s] = W[g*16 + fr][k = s*32 + fg*8 .. +8], k = tap*16 + ci
  u32x4 wf[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int k = s * 32 + (int)fg * 8;
    wf[s] = u32x4{0u, 0u, 0u, 0u};
    if (k < 144) wf[s] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)(g * 16u + fr) * 144 + k));
  }
  const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + g * 16u + fg * 4u);
  const f32x4 bi = *reinterpret_cast<const f32x4*>(p.bias + g * 16u + fg * 4u);
  const ActSel as = act_sel(p.act);
  const bool any_sig = act_is_sig(p.act), any_clamp = act_is_clamp(p.act);
  const u32 hw = (u32)(p.Ho * p.Wo);

#pragma unroll 2
  for (int f = 0; f < GC_FRAGS; ++f) {
    const long m = ((long)mb * GC_FRAGS + f) * 16 + fr;
    const bool live = m < p.M;
    int n = 0, oy = 0, ox = 0;
    if (live) {
      n = (int)(m / hw);
      const u32 r = (u32)(m % hw);
      oy = (int)(r / (u32)p.Wo);
      ox = (int)(r % (u32)p.Wo);
    }
    const u16* xin = p.x + (size_t)n * p.H * p.W * C + g * 16u + (fg & 1u) * 8u;
    u32x4 xf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + (int)(fg >> 1);  // k = s*32 + fg*8 -> tap = k / 16
      const int iy = oy * S - 1 + tap / 3, ix = ox * S - 1 + tap % 3;
      xf[s] = u32x4{0u, 0u, 0u, 0u};
      if (live && tap < 9 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        xf[s] = *reinterpret_cast<const u32x4*>(xin + ((size_t)iy * p.W + ix) * C);
    }
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) e = mfma16<DT>(wf[s], xf[s], e);  // D[channel fg*4+r][pixel fr]
    if (live) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = e[r] * sc[r] + bi[r];
      if (any_sig) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
          v[r] = as.mode == 1 ? sg : v[r] * sg;
        }
      }
      if (any_clamp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], as.lo), as.hi);
      }
      *reinterpret_cast<uint2*>(p.y + (size_t)m * C + g * 16u + fg * 4u) =
          make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
    }
  }
}

// ---- round 3: the same convolution from an LDS halo tile ------------------------------------------------------------------
// gconv3x3_g16_kernel reads every input pixel nine times as 32-byte pieces (16 channels of one tap of one pixel per half
// lane-row): 16 rows x 32 B per wave load is the access shape that tops out near 14 B/clk per CU (tools/micro/wstream.hip),
// and the kernel's time is exactly that: 128 channels on 112x112 at batch 16 = 462 MB of tap reads / 256 CUs / 14 B/clk =
// 54 us of a measured 66.  Here a workgroup stages the halo of a TH x 16 output patch for CBG groups ONCE, with coalesced
// 16-byte loads along the channels (NHWC: 32 CBG contiguous bytes per pixel), and the B fragments come from LDS:
//   * patch: 8 x 16 output pixels x 8 groups (stride 1: 10 x 18 halo pixels x 256 B = 49 KiB) or 4 x 16 x 4 groups (stride
//     2: 9 x 33 x 128 B = 43 KiB): three workgroups per CU; halo rows padded by 16 B (an odd number of 16-byte slots);
//   * wave w takes the groups w, w + 4, ... of the block: weights in registers as before, per 16-pixel fragment five
//     ds_read_b128 + five MFMAs, D = W * X^T, 8-byte NHWC stores straight from the accumulators.
template <int DT, int S>
__global__ __launch_bounds__(256) void gconv3x3_g16_tile_kernel(const GconvParams p, int tiles_x, int tiles_y, int cblocks) {
  constexpr int TH = S == 1 ? 8 : 4, TW = 16;
  constexpr int CBG = S == 1 ? 8 : 4;            // groups per workgroup
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int RS = CBG * 32 + 16;              // LDS bytes per halo pixel
  constexpr int NF = TH * TW / 16;               // 16-pixel fragments (one output row each)
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  u32 b = blockIdx.x;
  const u32 cb = b % (u32)cblocks;
  b /= (u32)cblocks;
  const u32 tx = b % (u32)tiles_x;
  b /= (u32)tiles_x;
  const u32 ty = b % (u32)tiles_y;
  const u32 n = b / (u32)tiles_y;
  const int C = p.C, g0 = (int)cb * CBG;
  const int ng = p.groups - g0 < CBG ? p.groups - g0 : CBG;  // groups of this block
  const int oy0 = (int)ty * TH, ox0 = (int)tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // ---- halo: IH x IW pixels x (16 ng) channels, 16-byte pieces, all of a thread's loads issued before its LDS stores --------
  {
    constexpr int PPP = CBG * 2;                 // 16-byte pieces per pixel
    constexpr int TOTAL = IH * IW * PPP, NPT = (TOTAL + 255) / 256;
    const u16* xin = p.x + (size_t)n * p.H * p.W * C + (size_t)g0 * 16;
    u32x4 v[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = (int)tid + k * 256, px = q / PPP, pc = q % PPP;
      const int iy = iy0 + px / IW, ix = ix0 + px % IW;
      v[k] = u32x4{0u, 0u, 0u, 0u};
      if (q < TOTAL && pc < 2 * ng && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v[k] = *reinterpret_cast<const u32x4*>(xin + ((size_t)iy * p.W + ix) * C + pc * 8);
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = (int)tid + k * 256, px = q / PPP, pc = q % PPP;
      if (q < TOTAL) *reinterpret_cast<u32x4*>(gsm + (size_t)px * RS + pc * 16) = v[k];
    }
  }
  __syncthreads();

  const ActSel as = act_sel(p.act);
  const bool any_sig = act_is_sig(p.act), any_clamp = act_is_clamp(p.act);
  for (int gl = (int)wave; gl < ng; gl += 4) {  // wave-uniform
    const u32 g = (u32)(g0 + gl);
    u32x4 wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int k = s * 32 + (int)fg * 8;
      wf[s] = u32x4{0u, 0u, 0u, 0u};
      if (k < 144) wf[s] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)(g * 16u + fr) * 144 + k));
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + g * 16u + fg * 4u);
    const f32x4 bi = *reinterpret_cast<const f32x4*>(p.bias + g * 16u + fg * 4u);
    // LDS byte offset of (this lane's pixel fr of fragment 0, tap of k-step s, channel half): k = 32 s + 8 fg -> tap 2 s +
    // (fg >> 1), channels 8 (fg & 1) .. + 7 of the group
    u32 xo[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + (int)(fg >> 1);
      const int ty_ = tap < 9 ? tap / 3 : 0, tx_ = tap < 9 ? tap % 3 : 0;  // (k-step 4, fg >= 2: padding -- zero weights)
      xo[s] = (u32)((ty_ * IW + (int)fr * S + tx_) * RS + gl * 32 + (int)(fg & 1u) * 16);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      u32x4 xf[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) xf[s] = *reinterpret_cast<const u32x4*>(gsm + xo[s] + (u32)(f * S * IW * RS));
      f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) e = mfma16<DT>(wf[s], xf[s], e);  // D[channel fg*4+r][pixel fr]
      const int oy = oy0 + f, ox = ox0 + (int)fr;
      if (oy < p.Ho && ox < p.Wo) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = e[r] * sc[r] + bi[r];
        if (any_sig) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
            v[r] = as.mode == 1 ? sg : v[r] * sg;
          }
        }
        if (any_clamp) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], as.lo), as.hi);
        }
        *reinterpret_cast<uint2*>(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * C + g * 16u + fg * 4u) =
            make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
      }
    }
  }
}

int launch_gconv3x3_g16(const ssdk_conv_desc* d, int Ho, int Wo, hipStream_t stream) {
  if (d->k != 3 || d->Cin != d->Cout || d->groups * 16 != d->Cin || !d->scale || d->residual || d->y2 ||
      d->in_layout != LAYOUT_NHWC || d->out_layout != LAYOUT_NHWC) {
    set_error("conv: grouped convolution is built for k=3, 16 channels per group, NHWC in/out, folded BN scale, no "
              "residual (Cin=%d Cout=%d groups=%d)", d->Cin, d->Cout, d->groups);
    return SSDK_E_BADARG;
  }
  GconvParams p;
  p.x = (const u16*)d->x;
  p.w = (const u16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.y = (u16*)d->y;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.C = d->Cin;
  p.stride = d->stride;
  p.Ho = Ho;
  p.Wo = Wo;
  p.act = d->act;
  p.groups = d->groups;
  p.M = (long)d->N * Ho * Wo;
  p.mblocks = (int)((p.M + 16 * GC_FRAGS - 1) / (16 * GC_FRAGS));
  const long grid = (long)p.mblocks * ((p.groups + 3) / 4);
  if (grid >= (1l << 31)) {
    set_error("conv: grouped convolution grid too large");
    return SSDK_E_BADARG;
  }
  static const int env_tile = getenv("SSDK_GCONV_TILE") ? atoi(getenv("SSDK_GCONV_TILE")) : 1;
  if (env_tile) {
    const int th = d->stride == 1 ? 8 : 4, cbg = d->stride == 1 ? 8 : 4;
    const int tiles_y = (Ho + th - 1) / th, tiles_x = (Wo + 15) / 16, cblocks = (p.groups + cbg - 1) / cbg;
    const long tgrid = (long)d->N * tiles_y * tiles_x * cblocks;
    const int ih = (th - 1) * d->stride + 3, iw = 15 * d->stride + 3;
    const size_t lds = (size_t)ih * iw * (cbg * 32 + 16);
    if (tgrid < (1l << 31) && Wo >= 8) {  // (narrower maps waste the 16-wide patch: the direct kernel)
#define SSDK_GCT(DT, S)                                                                                                      \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv3x3_g16_tile_kernel<DT, S>),                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    hipLaunchKernelGGL((gconv3x3_g16_tile_kernel<DT, S>), dim3((unsigned)tgrid), dim3(256), lds, stream, p, tiles_x, tiles_y, cblocks); \
  } while (0)
      if (d->dtype == SSDK_BF16) {
        if (d->stride == 1) SSDK_GCT(SSDK_BF16, 1);
        else SSDK_GCT(SSDK_BF16, 2);
      } else {
        if (d->stride == 1) SSDK_GCT(SSDK_F16, 1);
        else SSDK_GCT(SSDK_F16, 2);
      }
#undef SSDK_GCT
      return check_launch("gconv3x3_g16_tile_kernel");
    }
  }
#define SSDK_GC(DT, S) hipLaunchKernelGGL((gconv3x3_g16_kernel<DT, S>), dim3((unsigned)grid), dim3(256), 0, stream, p)
  if (d->dtype == SSDK_BF16) {
    if (d->stride == 1) SSDK_GC(SSDK_BF16, 1);
    else SSDK_GC(SSDK_BF16, 2);
  } else {
    if (d->stride == 1) SSDK_GC(SSDK_F16, 1);
    else SSDK_GC(SSDK_F16, 2);
  }
#undef SSDK_GC
  return check_launch("gconv3x3_g16_kernel");
}

}  // namespace ssdk
