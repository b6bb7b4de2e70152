// ssdk_xpair.hip -- one SSD "extra" layer (ssd.py:88-99 via basic_layers.py:40-57: Conv 1x1 + BN + ReLU followed by
// Conv 3x3 / stride 2 / pad 1 + BN + ReLU) on a SMALL map (<= 64 pixels) as ONE launch, the intermediate map in LDS.
//
// Why: on the 8x8 / 4x4 / 2x2 maps of SSD-MobileNetV2@512 these pairs are six dependent launches of 13-26 us each on
// the critical path of the forward pass (113 us of a 1.58 ms step) although each holds microseconds of work: as
// separate implicit GEMMs they are bound by launch gaps, split-K fences and one-wave-per-SIMD k-loops.  Here a
// workgroup (4 waves) owns (image, quarter of the output channels):
//   phase 1  mid[px][cm] = act(bn(x[px][:] . W1[cm][:]))      all of it (each of the 4 workgroups of an image recomputes
//            it: 8 MFLOP), MFMA operands straight from global memory / L2, results to LDS in the model dtype
//   phase 2  y[opx][co] = act(bn(sum_tap,cm mid[ipx(opx, tap)][cm] . W2[co][tap][cm]))   for its 64 output channels:
//            B operand gathered from the LDS map (out-of-map taps read a zero row), A operand (weights) from global
// Both k-loops keep PF k-steps of global loads in flight through register stages (branch-free, so the compiler counts
// them); what a workgroup has to stream is its weights (W1 + a quarter of W2: 280 KB on the 8x8 level), which is the
// bound that remains (~12 B/clk per CU from L2).
#include "ssdk_conv_common.h"

namespace ssdk {

struct XpairParams {
  const u16* x;
  u16* y;
  const u16* w1;
  const float* s1;
  const float* b1;
  const u16* w2;
  const float* s2;
  const float* b2;
  int N, H, W, Cin, Cmid, Cout, Ho, Wo, act1, act2;
};

constexpr int kXpThreads = 256;

__device__ __forceinline__ float xp_act(float v, int act) {
  if (act == SSDK_ACT_RELU) return __builtin_fmaxf(v, 0.f);
  if (act == SSDK_ACT_RELU6) return __builtin_fminf(__builtin_fmaxf(v, 0.f), 6.f);
  return v;
}

// MF1 = pixel fragments of the input map (ceil(H*W / 16)), NPW = mid-channel fragments per wave (Cmid = 64 * NPW),
// NF2 = output-channel fragments of the workgroup's quarter (Cout = 64 * NF2; KSPLIT = 4 / NF2 waves share a fragment and
// split the 32-channel slices), CS1 = Cin / 32.  Both k-loops are unrolled completely and have no branch: only then does
// the compiler count the loads in flight (s_waitcnt vmcnt(n)) instead of draining them at every loop back edge / merge.
// PK: both weight matrices come as fragment-major images (ssdk.h ssdk_weight_frag_bytes: 1 KiB contiguous per wave load)
template <int DT, int MF1, int NPW, int NF2, int CS1, bool PK>
__global__ __launch_bounds__(kXpThreads) void xpair_kernel(const XpairParams p) {
  constexpr int KSPLIT = 4 / NF2;
  constexpr int CSM = 2 * NPW;          // 32-channel slices of the intermediate map
  constexpr int SPW = CSM / KSPLIT;     // slices per wave in phase 2
  constexpr int RB = SPW >= 2 ? 2 : 1;  // weight ring: RB slices x 9 taps ahead
  static_assert(CSM % KSPLIT == 0 && CS1 % 4 == 0, "shape");
  constexpr int PF = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int n = (int)blockIdx.x >> 2, cq = (int)blockIdx.x & 3;
  const int P = p.H * p.W, OP = p.Ho * p.Wo, Cin = p.Cin, Cmid = p.Cmid, Cout = p.Cout;
  const int MS = Cmid * 2 + 16;  // LDS row stride of the intermediate map (bytes), rows 0..P-1, row P = zeros
  unsigned char* red = smem + (size_t)(P + 1) * MS;  // [4 waves][64 lanes] f32x4 partial sums (KSPLIT > 1)

  for (u32 i = tid; i < (u32)(MS / 4); i += kXpThreads) reinterpret_cast<u32*>(smem + (size_t)P * MS)[i] = 0u;

  // ---- phase 1: the 1x1 convolution of the whole map ----------------------------------------------------------------
  {
    f32x4 acc[NPW][MF1];
#pragma unroll
    for (int a = 0; a < NPW; ++a)
#pragma unroll
      for (int m = 0; m < MF1; ++m) acc[a][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u16* xa[MF1];
    const u16* wa[NPW];
#pragma unroll
    for (int m = 0; m < MF1; ++m) {
      int px = m * 16 + (int)fr;
      px = px < P ? px : P - 1;  // surplus columns of the fragment: computed, never stored
      xa[m] = p.x + ((size_t)n * P + px) * Cin + fg * 8;
    }
#pragma unroll
    for (int a = 0; a < NPW; ++a)
      wa[a] = PK ? p.w1 + (size_t)(wave + 4u * (u32)a) * CS1 * 512 + lane * 8 : p.w1 + (size_t)((wave + 4u * (u32)a) * 16u + fr) * Cin + fg * 8;
    u32x4 ra[PF][NPW], rb[PF][MF1];
    auto issue = [&](int slot, int ks) {
#pragma unroll
      for (int a = 0; a < NPW; ++a) ra[slot][a] = *reinterpret_cast<const u32x4*>(wa[a] + ks * (PK ? 512 : 32));
#pragma unroll
      for (int m = 0; m < MF1; ++m) rb[slot][m] = *reinterpret_cast<const u32x4*>(xa[m] + ks * 32);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(s, s);
#pragma unroll
    for (int ks = 0; ks < CS1; ++ks) {
#pragma unroll
      for (int a = 0; a < NPW; ++a)
#pragma unroll
        for (int m = 0; m < MF1; ++m) acc[a][m] = mfma16<DT>(ra[ks % PF][a], rb[ks % PF][m], acc[a][m]);  // D[cm = 4fg + r][px = fr]
      if (ks + PF < CS1) issue(ks % PF, ks + PF);  // (compile-time condition)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NPW; ++a) {
      const int cm = (int)(wave + 4u * (u32)a) * 16 + (int)fg * 4;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s1 + cm), bi = *reinterpret_cast<const f32x4*>(p.b1 + cm);
#pragma unroll
      for (int m = 0; m < MF1; ++m) {
        const int px = m * 16 + (int)fr;
        if (px < P) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = xp_act(fmaf(acc[a][m][r], sc[r], bi[r]), p.act1);
          *reinterpret_cast<uint2*>(smem + (size_t)px * MS + cm * 2) = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
        }
      }
    }
  }

  // ---- phase 2: the 3x3 / stride 2 convolution for this workgroup's quarter of the output channels -------------------
  const int nf = (int)wave % NF2, kpart = (int)wave / NF2;
  const int co_row = cq * (NF2 * 16) + nf * 16 + (int)fr;  // A operand row of this lane
  const int sl_beg = kpart * SPW;                            // this wave's slices [sl_beg, sl_beg + SPW)
  u32 rowoff[9];  // LDS byte offset of the input pixel behind (output pixel fr, tap), the zero row if outside the map
  {
    const int opx = (int)fr < OP ? (int)fr : 0, oy = opx / p.Wo, ox = opx % p.Wo;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = 2 * oy - 1 + t / 3, ix = 2 * ox - 1 + t % 3;
      const bool ok = (int)fr < OP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      rowoff[t] = (u32)((ok ? iy * p.W + ix : P) * MS) + fg * 16u;
    }
  }
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  const u16* w2row = PK ? p.w2 + ((size_t)(cq * NF2 + nf) * 9 * CSM + sl_beg) * 512 + lane * 8
                        : p.w2 + (size_t)co_row * 9 * Cmid + fg * 8 + (size_t)sl_beg * 32;
  u32x4 rw[RB][9];
  auto issue2 = [&](int buf, int t, int s) {  // weights of (tap t, slice sl_beg + s)
    rw[buf][t] = *reinterpret_cast<const u32x4*>(PK ? w2row + (size_t)(t * CSM + s) * 512 : w2row + (size_t)t * Cmid + s * 32);
  };
  // k-step order: slice pairs outside, taps inside, the two slices of a pair innermost -- neighbouring loads then fetch the
  // two 64-byte halves of the same 128-byte weight line (as in conv_smallmap_kernel)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int b2 = 0; b2 < RB; ++b2) issue2(b2, t, b2);  // in flight under the rest of phase 1 of the other waves
  __syncthreads();  // the intermediate map is complete
#pragma unroll
  for (int sp = 0; sp < SPW; sp += RB) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int h = 0; h < RB; ++h) {
        const int s = sp + h;
        const u32x4 bfrag = *reinterpret_cast<const u32x4*>(smem + rowoff[t] + (sl_beg + s) * 64);
        acc2 = mfma16<DT>(rw[h][t], bfrag, acc2);  // D[co = 4fg + r][opx = fr]
        if (s + RB < SPW) issue2(h, t, s + RB);  // (compile-time condition)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if constexpr (KSPLIT > 1) {
    *reinterpret_cast<f32x4*>(red + ((size_t)wave * 64 + lane) * 16) = acc2;
    __syncthreads();
    if (kpart != 0) return;
#pragma unroll
    for (int k = 1; k < KSPLIT; ++k) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((size_t)(k * NF2 + nf) * 64 + lane) * 16);
      acc2 += o;
    }
  }
  if ((int)fr < OP) {
    const int co = cq * (NF2 * 16) + nf * 16 + (int)fg * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + co), bi = *reinterpret_cast<const f32x4*>(p.b2 + co);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = xp_act(fmaf(acc2[r], sc[r], bi[r]), p.act2);
    *reinterpret_cast<uint2*>(p.y + ((size_t)n * OP + fr) * Cout + co) = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
  }
}

template <int DT, int MF1, int NPW, int NF2, int CS1>
static void xp_launch(const XpairParams& p, bool packed, size_t lds, hipStream_t stream) {
  if (packed) hipLaunchKernelGGL((xpair_kernel<DT, MF1, NPW, NF2, CS1, true>), dim3((unsigned)p.N * 4u), dim3(kXpThreads), lds, stream, p);
  else hipLaunchKernelGGL((xpair_kernel<DT, MF1, NPW, NF2, CS1, false>), dim3((unsigned)p.N * 4u), dim3(kXpThreads), lds, stream, p);
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_xpair(const ssdk_xpair_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->scale1 || !d->bias1 || !d->scale2 || !d->bias2) {
    set_error("xpair: null pointer");
    return SSDK_E_BADARG;
  }
  const int P = d->H * d->W;
  const bool acts_ok = (d->act1 == SSDK_ACT_NONE || d->act1 == SSDK_ACT_RELU || d->act1 == SSDK_ACT_RELU6) &&
                       (d->act2 == SSDK_ACT_NONE || d->act2 == SSDK_ACT_RELU || d->act2 == SSDK_ACT_RELU6);
  if ((d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) || d->N < 1 || d->H < 1 || d->W < 1 || P > 64 || (P > 16 && P != 64) ||
      d->Cin < 128 || (d->Cin % 128) || (d->Cmid != 64 && d->Cmid != 128) || (d->Cout != 128 && d->Cout != 256) ||
      (9 * (d->Cmid / 32)) % (4 / (d->Cout / 64)) || !acts_ok) {
    set_error("xpair: unsupported geometry H=%d W=%d Cin=%d Cmid=%d Cout=%d (maps of <= 16 or exactly 64 pixels, Cin %% 128 == 0, "
              "Cmid 64|128, Cout 128|256, none/relu/relu6)", d->H, d->W, d->Cin, d->Cmid, d->Cout);
    return SSDK_E_BADARG;
  }
  XpairParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  static const int env_pk = getenv("SSDK_WFRAG") ? atoi(getenv("SSDK_WFRAG")) : 1;
  const bool packed = d->w1_frag && d->w2_frag && env_pk != 0;
  p.w1 = (const u16*)(packed ? d->w1_frag : d->w1);
  p.s1 = d->scale1;
  p.b1 = d->bias1;
  p.w2 = (const u16*)(packed ? d->w2_frag : d->w2);
  p.s2 = d->scale2;
  p.b2 = d->bias2;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.Cmid = d->Cmid;
  p.Cout = d->Cout;
  p.Ho = (d->H + 2 - 3) / 2 + 1;
  p.Wo = (d->W + 2 - 3) / 2 + 1;
  p.act1 = d->act1;
  p.act2 = d->act2;
  const size_t lds = (size_t)(P + 1) * (d->Cmid * 2 + 16) + 4 * 64 * 16;
  const int mf1 = (P + 15) / 16, npw = d->Cmid / 64, nf2 = d->Cout / 64, cs1 = d->Cin / 32;
  // the instances that exist (every one is a fully unrolled kernel): the three extras of SSD-MobileNetV2@512 and a 128-wide one
#define SSDK_XP(DT)                                                                                     \
  do {                                                                                                  \
    if (mf1 == 4 && npw == 2 && nf2 == 4 && cs1 == 16) xp_launch<DT, 4, 2, 4, 16>(p, packed, lds, stream);      \
    else if (mf1 == 1 && npw == 2 && nf2 == 4 && cs1 == 8) xp_launch<DT, 1, 2, 4, 8>(p, packed, lds, stream);   \
    else if (mf1 == 1 && npw == 1 && nf2 == 2 && cs1 == 8) xp_launch<DT, 1, 1, 2, 8>(p, packed, lds, stream);   \
    else if (mf1 == 1 && npw == 1 && nf2 == 2 && cs1 == 4) xp_launch<DT, 1, 1, 2, 4>(p, packed, lds, stream);   \
    else {                                                                                              \
      set_error("xpair: no instance for %d pixel fragments, Cin=%d, Cmid=%d, Cout=%d", mf1, d->Cin, d->Cmid, d->Cout); \
      return SSDK_E_BADARG;                                                                             \
    }                                                                                                   \
  } while (0)
  if (d->dtype == SSDK_BF16) SSDK_XP(SSDK_BF16);
  else SSDK_XP(SSDK_F16);
#undef SSDK_XP
  return check_launch("xpair_kernel");
}
