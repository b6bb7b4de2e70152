// ssdk_smallmap.hip -- 3x3 / stride 1 / pad 1 convolution on maps of at most 64 pixels (the multibox heads of the 8x8, 4x4,
// 2x2 and 1x1 levels, ssd.py:100-103; any dense layer of that shape), one of the kernels behind ssdk_conv.
//
// Why: with M = N * H * W of a few hundred rows and K = 9 * Cin of a few thousand these layers are all weights.  As
// 128-row implicit-GEMM tiles with split-K (conv_gemm_kernel) or one wave per 16 x 64 tile (conv_wave_kernel) they take
// 20-40 us each -- k-loops of one wave per SIMD plus split-K fences -- for well under a microsecond of MFMA work.
// Here a workgroup owns 64 PIXELS (= 64 / (H*W) whole images) x 64 output channels:
//   * the feature maps of its images are staged once in LDS (64 rows of Cin, + a zero row for the padding);
//   * a wave owns 16 output channels and all four pixel fragments: per k-step (tap, 32-channel slice) ONE weight
//     fragment from global memory (A operand, six k-steps in flight through registers) serves four MFMAs whose B
//     operands are gathered from the LDS map with a per-(pixel, tap) row offset;
//   * a weight is read once per 64 pixels instead of once per 16, there is no split-K, no fence, no barrier after staging.
// What remains is the weight stream itself (64 x 9 x Cin x 2 bytes per workgroup).
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int kSmThreads = 256;
__host__ __device__ constexpr int sm_pad(int stride) { return stride == 1 ? 32 : 16; }  // (host side sizes every instance for 32)

// CS = Cin / 32 is a template parameter: the k-loop is unrolled completely, because a loop back edge makes the compiler
// drain the load counter at the top of every iteration (s_waitcnt vmcnt(0)) however the body is written
// MFR = pixel fragments per workgroup (4: 64 pixels, 8: 128 pixels -- half the weight traffic per pixel, for the level
// whose grid still fills the chip then)
// TAPS = 9, or 1 for 1x1 maps: only the centre tap ever sees data there (the other eight multiply the padding), so K is Cin
// PK: weights from the fragment-major image p.w_frag (ssdk.h: 1 KiB contiguous per k-step and wave) instead of the KRSC tensor
// NA: 16-channel weight fragments per wave (1 | 2).  With 2 (fragment-major weights only) a pixel fragment read from LDS
// serves two MFMAs: the 8x8 level as 128 pixels x 16 channels per wave was bound by its LDS reads (every wave reads every
// pixel fragment: 8 KiB per wave and k-step for 128 cycles of matrix work); 64 pixels x 32 channels per wave reads 4 KiB
// S: stride (1 | 2).  With 2 the staged map is the INPUT map (H x W = 2 Ho x 2 Wo pixels per image) and a lane's sixteen
// pixels sit two input columns / two input rows apart; the LDS image skews every second input row by one row slot
// (row index iy * W + ix + (iy >> 1)) so that the sixteen 16-byte reads of a fragment still fall into sixteen different
// bank groups (row stride = an odd number of 16-byte chunks; without the skew output rows oy and oy + 1 collide).
// KW (round 6): wave GROUPS per workgroup that split K between them (1 | 2): group kg takes the 32-channel slices
// sl = kg (mod KW) of every tap, i.e. every KW-th KiB of the fragment-major image and the LDS columns 64 kg (mod 64 KW) -- a
// pointer offset and a row-offset bias, every other address stays an immediate.  The layer is a chain of dependent latencies
// (L2 weight fragment -> MFMA) at ONE wave per SIMD: the 8x8 level streamed 1.18 MB of weights per workgroup through four such
// chains in 29 us, for 7.7 us of matrix work (Weak 5).  With KW = 2 a SIMD holds two waves, each walking half
// the k-steps; the partial accumulators of group 1 meet group 0's through LDS once at the end (fp32, group 0 + group 1: a fixed
// order).
template <int DT, int CS, int MFR, int TAPS, bool PK, int NA, int S, int KW = 1>
__device__ __forceinline__ void smallmap_body(const ConvParams& p, const u32 bx, const u32 by) {
  static_assert(NA == 1 || PK, "two channel fragments per wave read the fragment-major image");
  static_assert(S == 1 || TAPS == 9, "stride 2 visits all taps");
  static_assert(KW == 1 || (PK && CS % (2 * KW) == 0), "the K split walks the fragment-major image");
  constexpr int CSL = CS / KW;  // slices per wave group
  constexpr int ROWS = 16 * MFR;
  constexpr int TAP0 = TAPS == 9 ? 0 : 4;  // first tap visited
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave_all = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 nwc = (u32)(blockDim.x >> 6) / (u32)KW;  // channel waves per workgroup
  const u32 wave = KW == 1 ? wave_all : wave_all % nwc, kg = KW == 1 ? 0u : wave_all / nwc;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int P = p.Ho * p.Wo, G = ROWS / P, Cin = p.Cin;  // output pixels per image, images per workgroup
  const int PI = p.H * p.W;                                 // input pixels per image (== P for stride 1)
  const int PL = S == 2 ? PI + (p.H >> 1) : PI;             // LDS row slots per image (skewed for stride 2)
  const int ZROW = G * PL;                                  // the zero row
  const int img0 = (int)bx * G;
  // LDS row stride (bytes); row ROWS = zeros.  Stride 1 (round 4): Cin * 2 + 32, i.e. R = stride / 16 = 2 (mod 4) -- the
  // four 16-lane groups a ds_read_b128 is serviced in mix eight pixels of k-piece fg with eight of piece fg + 1, and only
  // R = 2 (mod 4) keeps their sixteen 16-byte slots apart (ssdk_conv3x3s.hip; the former Cin * 2 + 16 measured 48 % conflict cycles)
  const int RS = Cin * 2 + sm_pad(S);
  const u16* x = (const u16*)p.x;

  // ---- the weight stream starts first: it does not depend on the maps -----------------------------------------------
  const int co_row0 = ((int)by * (int)nwc + (int)wave) * 16 * NA;  // 1, 2 or 4 channel waves per workgroup
  int co_a = co_row0 + (int)fr;
  co_a = co_a < p.Cout ? co_a : p.Cout - 1;  // rows past Cout: computed on a valid row, never stored
  const u16* wrow = PK ? (const u16*)p.w_frag + (size_t)(co_row0 < p.Cout ? co_row0 >> 4 : (p.Cout - 1) >> 4) * 9 * Cin * 16 + lane * 8 + kg * 512u
                       : (const u16*)p.w + (size_t)co_a * 9 * Cin + fg * 8;
  static_assert(CS % 2 == 0, "even number of slices");
  // k-loop (static: the row offsets are plain registers).  Weight
  // fragments run 18 k-steps ahead of the MFMAs through 18 register stages; the loop body has no branch, so
  // the compiler counts the loads in flight (s_waitcnt vmcnt(17)) instead of draining them -- a k-step is four MFMAs
  // (~70 cycles), an L2 round trip 2-4k cycles, and with a branch in the body every k-step waited for its own load
  // (1.2k cycles per k-step measured).
  // Order of the k-steps.  PK: taps outside, slices inside = the order of the packed image, consecutive KiB.  KRSC: slice
  // PAIRS outside, taps inside, the two slices of a pair innermost -- the two 64-byte halves of a weight row's 128-byte line
  // are then fetched by neighbouring loads.
  constexpr int RING = KW == 1 ? 2 * TAPS : (TAPS == 9 ? 9 : 2);  // register stages = k-steps in flight (two waves per SIMD: half)
  u32x4 rw[RING][NA];
  auto kmap = [](int st, int& t, int& sl) {  // sl: the wave group's slice number times KW (+ kg: in the pointers)
    if (PK) { t = st / CSL; sl = (st % CSL) * KW; return; }
    if (TAPS == 1) { t = 0; sl = st; return; }
    const int sp = st / RING, in = st % RING;
    t = in >> 1;
    sl = 2 * sp + (in & 1);
  };
  auto issue = [&](int st) {
    int t, sl;
    kmap(st < CSL * TAPS ? st : CSL * TAPS - 1, t, sl);  // past the end: a harmless re-read
#pragma unroll
    for (int a = 0; a < NA; ++a) {  // (fragment a of this wave: the next 16-row group of the image, clamped like the first)
      const size_t ga = (a && co_row0 + 16 * a < p.Cout) ? (size_t)a * 9 * Cin * 16 : 0;
      rw[st % RING][a] = *reinterpret_cast<const u32x4*>(wrow + ga + (size_t)((t + TAP0) * CS + sl) * (PK ? 512 : 32));
    }
  };
#pragma unroll
  for (int st = 0; st < RING; ++st) issue(st);
  // the epilogue's per-channel constants too (as first written they were loaded in the epilogue: one more exposed round trip)
  const int co0 = co_row0 + (int)fg * 4;
  float e_sc[NA][4], e_bi[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * a + r < p.Cout ? co0 + 16 * a + r : p.Cout - 1;
      e_sc[a][r] = p.scale ? p.scale[co] : 1.f;
      e_bi[a][r] = p.bias[co];
    }

  // ---- stage the maps: ROWS rows x Cin, 16-byte pieces, rows of images past N are zero, row ROWS = zeros ---------------
  // Batches of up to 17 independent loads per thread, then their LDS stores.  (As first written -- one load, one store per
  // loop iteration -- the nine iterations of the 4x4 level each waited a full memory round trip: ~9 us of a 19 us kernel.)
  {
    constexpr int CPR = CS * 4;
    const int TOTAL = G * PI * CPR;  // (stride 1: ROWS * CPR)
    constexpr int PER = (ROWS * S * S * CPR + kSmThreads - 1) / kSmThreads, SB = PER < 17 ? PER : 17;  // one or two round trips
    const int nthr = (int)blockDim.x;
    for (int q0 = (int)tid; q0 < TOTAL; q0 += nthr * SB) {
      u32x4 v[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int q = q0 + j * nthr, row = q / CPR, c = q % CPR;
        v[j] = u32x4{0u, 0u, 0u, 0u};
        if (q < TOTAL && img0 + row / PI < p.N) v[j] = *reinterpret_cast<const u32x4*>(x + ((size_t)img0 * PI + row) * Cin + c * 8);
      }
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int q = q0 + j * nthr, row = q / CPR, c = q % CPR;
        int lrow = row;
        if (S == 2) {
          const int g = row / PI, qi = row % PI;
          lrow = g * PL + qi + ((qi / p.W) >> 1);
        }
        if (q < TOTAL) *reinterpret_cast<u32x4*>(smem + (size_t)lrow * RS + c * 16) = v[j];
      }
    }
    if ((int)tid < CPR) *reinterpret_cast<u32x4*>(smem + (size_t)ZROW * RS + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    if (S == 2) {  // the skew slots (one per two input rows) are never read; nothing to clear
    }
  }
  // LDS byte offset of the input pixel behind (output pixel = fragment m, lane fr; tap), the zero row outside the map
  u32 rowoff[MFR][TAPS];
#pragma unroll
  for (int m = 0; m < MFR; ++m) {
    const int px = m * 16 + (int)fr, g = px / P, q = px % P, oy = q / p.Wo, ox = q % p.Wo;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int iy = oy * S - 1 + (t + TAP0) / 3, ix = ox * S - 1 + (t + TAP0) % 3;
      const bool ok = g < G && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;  // (g >= G: a pixel slot past the
      rowoff[m][t] = (u32)((ok ? g * PL + iy * p.W + ix + (S == 2 ? iy >> 1 : 0) : ZROW) * RS) + fg * 16u + kg * 64u;  // workgroup's images: 64 % P != 0)
    }
  }
  __syncthreads();

  f32x4 acc[NA][MFR];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int m = 0; m < MFR; ++m) acc[a][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The pixel operands (B fragments, LDS) run TWO k-steps ahead of the MFMAs through a ring of three register sets.  (As
  // first written every MFMA waited for a ds_read issued one or two instructions before it -- s_waitcnt lgkmcnt(1) in front
  // of each of the four MFMAs of a k-step: ~660 cycles per k-step for 64 cycles of matrix work, the whole kernel ran at the
  // LDS latency of one wave per SIMD.)
  constexpr int NS = CSL * TAPS;  // k-steps of this wave group
  constexpr int D = 2;           // B-fragment prefetch distance
  u32x4 bq[D + 1][MFR];
  auto lds_issue = [&](int slot, int step) {
    int t, sl;
    kmap(step, t, sl);
#pragma unroll
    for (int m = 0; m < MFR; ++m) bq[slot][m] = *reinterpret_cast<const u32x4*>(smem + rowoff[m][t] + sl * 64);
  };
#pragma unroll
  for (int st = 0; st < D; ++st) lds_issue(st, st < NS ? st : NS - 1);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + D < NS) lds_issue((st + D) % (D + 1), st + D);
#pragma unroll
    for (int m = 0; m < MFR; ++m)
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[a][m] = mfma16<DT>(rw[st % RING][a], bq[st % (D + 1)][m], acc[a][m]);  // D[co = 4fg + r][px = fr]
    issue(st + RING);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks every load down to its use, 18 k-steps later)
  }

  if constexpr (KW > 1) {  // the wave groups' partial sums meet in LDS behind the maps: group 0 adds groups 1.. in order
    unsigned char* red = smem + (((size_t)(ZROW + 1) * RS + 15) & ~(size_t)15);
    if (kg > 0) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MFR; ++m)
          *reinterpret_cast<f32x4*>(red + ((((size_t)(kg - 1) * nwc + wave) * NA + a) * MFR + m) * 1024 + lane * 16) = acc[a][m];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KW; ++g2)
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MFR; ++m) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((((size_t)(g2 - 1) * nwc + wave) * NA + a) * MFR + m) * 1024 + lane * 16);
          acc[a][m] += o;
        }
  }
  // ---- epilogue: lane = (pixel fr of fragment m, channels co0 + 4fg .. +3) -----------------------------------------
  const bool nchw = p.out_layout == LAYOUT_NCHW;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * a + r;
      if (co >= p.Cout) continue;
      const float sc = e_sc[a][r], bi = e_bi[a][r];
      const ActSel as = act_sel(co >= p.split ? p.act2 : p.act);
#pragma unroll
      for (int m = 0; m < MFR; ++m) {
        const int px = m * 16 + (int)fr, b = img0 + px / P, q = px % P;
        if (px / P >= G || b >= p.N) continue;
        float v = acc[a][m][r] * sc + bi;
        if (as.mode) {
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v));
          v = as.mode == 1 ? sg : v * sg;
        }
        v = __builtin_fminf(__builtin_fmaxf(v, as.lo), as.hi);
        const u16 h = (u16)(pack2_16<DT>(v, 0.f) & 0xffffu);
        if (!nchw) ((u16*)p.y)[((size_t)b * P + q) * p.Cout + co] = h;
        else if (co < p.split) ((u16*)p.y)[((size_t)b * p.split + co) * P + q] = h;
        else ((u16*)p.y2)[((size_t)b * (p.Cout - p.split) + (co - p.split)) * P + q] = h;
      }
    }
}

template <int DT, int CS, int MFR, int TAPS, bool PK, int NA, int S, int KW = 1>
__global__ __launch_bounds__(kSmThreads * KW) void conv_smallmap_kernel(const ConvParams p) {
  smallmap_body<DT, CS, MFR, TAPS, PK, NA, S, KW>(p, blockIdx.x, blockIdx.y);
}

// Several independent small-map layers in ONE launch (the multibox heads of the 4x4 / 2x2 / 1x1 levels: 128 + 32 + 8
// workgroups whose kernels are chains of dependent latencies -- weights, map, epilogue -- of 8-12 us each with most of the
// chip idle; run together they cost the longest one).  Members: 3x3 / stride 1, fragment-major weights, 64 pixels x 16
// channels per wave (MFR = 4, NA = 1), 256 threads.  The executor groups neighbouring plan ops that qualify and do not
// read each other's outputs (ssdk_run_ops).
struct SmallmapGroup {
  ConvParams p[kSmallmapGroupMax];
  unsigned start[kSmallmapGroupMax + 1];  // first workgroup of member i
  unsigned gx[kSmallmapGroupMax];         // grid.x of member i (image groups)
  int code[kSmallmapGroupMax];            // 2 * log2(Cin / 128) + (1x1 map)
  int n;
};
template <int DT, int KW>
__global__ __launch_bounds__(kSmThreads * KW) void conv_smallmap_group_kernel(const SmallmapGroup g) {
  int m = 0;
  for (int i = 1; i < g.n; ++i)
    if (blockIdx.x >= g.start[i]) m = i;
  const u32 local = blockIdx.x - g.start[m], bx = local % g.gx[m], by = local / g.gx[m];
  const ConvParams& p = g.p[m];
  switch (g.code[m]) {
    case 0: smallmap_body<DT, 4, 4, 9, true, 1, 1, KW>(p, bx, by); break;
    case 1: smallmap_body<DT, 4, 4, 1, true, 1, 1, KW>(p, bx, by); break;
    case 2: smallmap_body<DT, 8, 4, 9, true, 1, 1, KW>(p, bx, by); break;
    case 3: smallmap_body<DT, 8, 4, 1, true, 1, 1, KW>(p, bx, by); break;
    case 4: smallmap_body<DT, 16, 4, 9, true, 1, 1, KW>(p, bx, by); break;
    default: smallmap_body<DT, 16, 4, 1, true, 1, 1, KW>(p, bx, by); break;
  }
}

// Can this layer be a member of a grouped launch?  (what launch_conv_smallmap would run as <CS, 4, TAPS, true, 1, 1>)
static bool smallmap_group_member(const ConvParams& p) {
  static const int env = getenv("SSDK_CONV_SMALLMAP") ? atoi(getenv("SSDK_CONV_SMALLMAP")) : 1;
  static const int env_pk = getenv("SSDK_WFRAG") ? atoi(getenv("SSDK_WFRAG")) : 1;
  const int P = p.Ho * p.Wo;
  if (!env || !env_pk || !p.w_frag || p.k != 3 || p.stride != 1 || p.pad != 1 || p.H != p.Ho || p.W != p.Wo || P > 64 ||
      (p.Cin != 128 && p.Cin != 256 && p.Cin != 512) || p.in_layout != LAYOUT_NHWC || p.res || p.post != SSDK_ACT_NONE || p.Cout < 16)
    return false;
  const int nfr64 = (p.Cout + 63) / 64;
  const bool big = 2 * P <= 128 && (128 % P) == 0 && (long)((p.N + 128 / P - 1) / (128 / P)) * nfr64 >= 256;
  return !big;  // (the 8x8 level at batch 64 fills the chip on its own, with another instance)
}

// 1: not a group this kernel takes (the caller launches the layers one by one), 0: launched
int launch_conv_smallmap_group(const ConvParams* ps, int n, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_CONV_SMALLMAP_GROUP") ? atoi(getenv("SSDK_CONV_SMALLMAP_GROUP")) : 1;
  if (!env || n < 2 || n > kSmallmapGroupMax) return 1;
  SmallmapGroup g;
  g.n = n;
  size_t lds = 0;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    const ConvParams& p = ps[i];
    if (!smallmap_group_member(p)) return 1;
    const int P = p.Ho * p.Wo, G = 64 / P;
    g.p[i] = p;
    g.gx[i] = (unsigned)((p.N + G - 1) / G);
    const unsigned gy = (unsigned)(((p.Cout + 15) / 16 + 3) / 4);
    g.start[i] = total;
    total += g.gx[i] * gy;
    g.code[i] = (p.Cin == 128 ? 0 : p.Cin == 256 ? 2 : 4) + (P == 1 ? 1 : 0);
    const size_t l = (size_t)(64 + 1) * (p.Cin * 2 + 32);
    lds = l > lds ? l : lds;
  }
  g.start[n] = total;
  // K split over two wave groups: measured on the 4x4 / 2x2 / 1x1 heads of SSD-MobileNetV2@512 (26.3 vs 26.5 us for the group:
  // nothing) and it changes the summation order against the members' single launches, which the executor's tests compare bit
  // for bit -- off unless SSDK_CONV_SMALLMAP_GROUP_KW=2
  constexpr int env_kw = 1;  // (round 6: the SSDK_CONV_SMALLMAP_GROUP_KW switch is gone, its A/B is settled)
  const int kw = env_kw == 2 ? 2 : 1;
  lds = ((lds + 15) & ~(size_t)15) + (size_t)(kw - 1) * 4 * 4 * 1024;  // + the partial sums of wave group 1 (4 waves x 4 fragments)
#define SSDK_SMG(DT, KW_)                                                                                                    \
  do {                                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallmap_group_kernel<DT, KW_>),                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                         \
    hipLaunchKernelGGL((conv_smallmap_group_kernel<DT, KW_>), dim3(total), dim3(kSmThreads * KW_), lds, stream, g);          \
  } while (0)
  if (dtype == SSDK_BF16) {
    if (kw == 2) SSDK_SMG(SSDK_BF16, 2);
    else SSDK_SMG(SSDK_BF16, 1);
  } else {
    if (kw == 2) SSDK_SMG(SSDK_F16, 2);
    else SSDK_SMG(SSDK_F16, 1);
  }
#undef SSDK_SMG
  return 0;
}

// 1: not one of this kernel's layers (the caller goes on), 0: launched
int launch_conv_smallmap(const ConvParams& p, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_CONV_SMALLMAP") ? atoi(getenv("SSDK_CONV_SMALLMAP")) : 1;
  constexpr int env_maxp = 64;  // (round 6: the SSDK_CONV_SMALLMAP_MAXP switch is gone, its A/B is settled)
  const int P = p.Ho * p.Wo;
  if (!env || p.k != 3 || p.pad != 1 || P > env_maxp || P > 64 || (p.Cin != 128 && p.Cin != 256 && p.Cin != 512) ||
      p.in_layout != LAYOUT_NHWC || p.res || p.post != SSDK_ACT_NONE || p.Cout < 16)
    return 1;
  static const int env_pk = getenv("SSDK_WFRAG") ? atoi(getenv("SSDK_WFRAG")) : 1;
  const bool packed = p.w_frag != nullptr && env_pk != 0;
  constexpr int env_s2 = 1;  // (round 6: the SSDK_CONV_SMALLMAP_S2 switch is gone, its A/B is settled)
  const bool s2 = p.stride == 2;
  if (s2) {  // the stride-2 instance: even input map, fragment-major weights, the input maps of a workgroup fit the LDS
    if (!env_s2 || !packed || p.H != 2 * p.Ho || p.W != 2 * p.Wo || P == 1) return 1;
    const size_t need = (size_t)((64 / P) * (p.H * p.W + p.H / 2) + 1) * (p.Cin * 2 + 32);
    if (need > 160 * 1024) return 1;
  } else if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo) {
    return 1;
  }
  // Where 128 pixels per workgroup still leave a workgroup per CU (the 8x8 level at batch 64): with fragment-major weights
  // 64 pixels x 32 channels per wave (na = 2: a pixel fragment serves two MFMAs), else 128 pixels x 16 channels per wave
  const int nfr64 = (p.Cout + 63) / 64;
  const bool big = 2 * P <= 128 && (128 % P) == 0 && (long)((p.N + 128 / P - 1) / (128 / P)) * nfr64 >= 256;
  constexpr int env_na = 2;  // (round 6: the SSDK_CONV_SMALLMAP_NA switch is gone, its A/B is settled)
  const int na = ((big && packed && env_na == 2) || s2) ? 2 : 1;
  const int mfr = (big && na == 1) ? 8 : 4;
  const int G = 16 * mfr / P;
  // waves (= 16 na-channel fragments) per workgroup
  const int groups = (p.N + G - 1) / G, nfr = (p.Cout + 16 * na - 1) / (16 * na);
  constexpr int env_nw = 4;  // (round 6: the SSDK_CONV_SMALLMAP_NW switch is gone, its A/B is settled)
  const int nw = env_nw == 1 || env_nw == 2 ? env_nw : 4;  // (fewer waves per workgroup = more workgroups: measured slower on every level)
  const dim3 grid((unsigned)groups, (unsigned)((nfr + nw - 1) / nw));
  size_t lds = s2 ? (size_t)(G * (p.H * p.W + p.H / 2) + 1) * (p.Cin * 2 + 32) : (size_t)(16 * mfr + 1) * (p.Cin * 2 + 32);
  const int cs = p.Cin / 32;
  // K split over two wave groups (smallmap_body, KW): the weight-bound instance of the 8x8 level (na == 2, stride 1, 4 waves).
  // Measured (SSD-MobileNetV2@512 head of the 8x8 level, batch 64, per-op events, one box): 32.6 -> 30.3 us.  Two waves per SIMD
  // were NOT what held this kernel at 0.26 of the MFMA peak; the weight stream is (1.18 MB per workgroup from L2).  Also measured
  // and dropped (round 6, session 13): a 36-deep instead of an 18-deep weight ring -- one wave per SIMD, 462 registers: 52 us on
  // this level, 17.6 vs 16.9 us on the grouped 4x4 / 2x2 / 1x1 heads: more fragments in flight per wave buy nothing, the stream
  // is throughput-bound, not latency-bound.
  static const int env_kw = getenv("SSDK_CONV_SMALLMAP_KW") ? atoi(getenv("SSDK_CONV_SMALLMAP_KW")) : 2;
  const int kw = (env_kw == 2 && na == 2 && !s2 && nw == 4 && P > 1) ? 2 : 1;
  if (kw == 2) lds = ((lds + 15) & ~(size_t)15) + (size_t)nw * na * mfr * 1024;
#define SSDK_SMS(DT, CS_, MFR_, TAPS_, PK_, NA_, S_)                                                                       \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallmap_kernel<DT, CS_, MFR_, TAPS_, PK_, NA_, S_>),   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    hipLaunchKernelGGL((conv_smallmap_kernel<DT, CS_, MFR_, TAPS_, PK_, NA_, S_>), grid, dim3(64 * nw), lds, stream, p);   \
  } while (0)
#define SSDK_SMK2(DT, CS_)                                                                                                 \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallmap_kernel<DT, CS_, 4, 9, true, 2, 1, 2>),          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    hipLaunchKernelGGL((conv_smallmap_kernel<DT, CS_, 4, 9, true, 2, 1, 2>), grid, dim3(64 * nw * 2), lds, stream, p);     \
  } while (0)
#define SSDK_SM0(DT, CS_, MFR_, TAPS_, PK_, NA_) SSDK_SMS(DT, CS_, MFR_, TAPS_, PK_, NA_, 1)
#define SSDK_SM1(DT, CS_, MFR_, TAPS_)                  \
  do {                                                  \
    if (packed) SSDK_SM0(DT, CS_, MFR_, TAPS_, true, 1); \
    else SSDK_SM0(DT, CS_, MFR_, TAPS_, false, 1);      \
  } while (0)
#define SSDK_SM(DT, CS_)                         \
  do {                                           \
    if (s2) SSDK_SMS(DT, CS_, 4, 9, true, 2, 2); \
    else if (P == 1) SSDK_SM1(DT, CS_, 4, 1);    \
    else if (na == 2 && kw == 2) SSDK_SMK2(DT, CS_);    \
    else if (na == 2) SSDK_SM0(DT, CS_, 4, 9, true, 2); \
    else if (mfr == 8) SSDK_SM1(DT, CS_, 8, 9);  \
    else SSDK_SM1(DT, CS_, 4, 9);                \
  } while (0)
  if (dtype == SSDK_BF16) {
    if (cs == 4) SSDK_SM(SSDK_BF16, 4);
    else if (cs == 8) SSDK_SM(SSDK_BF16, 8);
    else SSDK_SM(SSDK_BF16, 16);
  } else {
    if (cs == 4) SSDK_SM(SSDK_F16, 4);
    else if (cs == 8) SSDK_SM(SSDK_F16, 8);
    else SSDK_SM(SSDK_F16, 16);
  }
#undef SSDK_SM
#undef SSDK_SM1
#undef SSDK_SM0
#undef SSDK_SMK2
#undef SSDK_SMS
  return 0;
}

}  // namespace ssdk
