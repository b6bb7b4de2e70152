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
