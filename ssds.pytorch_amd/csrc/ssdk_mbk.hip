// ssdk_mbk.hip -- MobileNetV2 inverted-residual block (nets/mobilenet.py:56, 84-89) for the LOW-RESOLUTION, WIDE blocks: a
// 16-pixel-wide map (the 16x16 maps of SSD-MobileNetV2@512: 160 -> 960 -> 160 | 320), 1x1 expand -> 3x3 depthwise -> 1x1
// project as ONE launch with every tensor in registers and the weights streamed from L2 straight into MFMA operands.
//
// Why (round 5).  On ssdk_mbconv.hip these blocks ran at 8 % of the MFMA peak (52 / 51 / 75 us for 10 / 10 / 15 GFLOP): an
// 8x8 tile per workgroup, ONE workgroup per CU (150 KB of LDS), per 64-channel chunk three phases between two workgroup
// barriers in which every one of the eight waves re-reads the chunk's whole weight image from LDS for 17 - 27 MFMAs, plus
// 1.56x halo work in the expand GEMM.  A 16x16 map at batch 64 is only 16 384 pixels -- 16 per SIMD of the chip -- so whatever
// the tiling, every CU streams (nearly) all 614 KB of the block's weights; what can be chosen is that nothing else moves:
//
//   * a work item is (image, PAIR of output rows): the map is exactly one MFMA fragment wide, so a row of 16 pixels is one
//     fragment, the horizontal taps are DPP lane shifts whose zero fill at lanes 0 / 15 IS the zero padding, and the vertical
//     taps are the other rows of the item.  Two output rows need four expanded rows (2x expand work, no halo columns at all);
//   * the NW waves of an item split the HIDDEN channels (as in ssdk_mbsplit.hip): wave w owns NCHW chunks of 16 and needs
//     nobody else's data until the very end -- no barrier in the main loop;
//   * a wave's weights are its own: they come from a fragment-major image in global memory (host-built, 1 KiB contiguous per
//     wave load: the access shape that streams best from L2) straight into the A operands of its MFMAs, used for
//     4 (expand) / 2 (project) fragments each and never staged in LDS;
//   * the block input of the item (4 rows x 16 px x Cin) sits in LDS as B fragments, filled by LDS-DMA, and is the only LDS
//     traffic of the main loop besides the depthwise taps; the per-channel constants (BN biases, taps) are part of the image,
//     laid out as the kernel reads them, and arrive by LDS-DMA too (the first version staged them with 2 160 strided 8-byte
//     loads per workgroup: 14 k of its 67 k cycles);
//   * the partial projections of the NW waves meet once, at the end: five fragments per round through a 20 KB exchange
//     buffer, summed in wave order (bit-reproducible), BN + residual + store by the wave that owns the fragment;
//   * measured and dropped: two items per workgroup whose waves (i, w) read the same fragments at the same time (loop 27.0 k
//     vs 27.3 k cycles: no L1 sharing to speak of, and the 8-wave exchange barriers cost 5 k cycles more); six or three slices
//     instead of four (32.6 / 29.7 vs 27.5 us).  With one item per workgroup and two workgroups per CU the loop runs at
//     1.2 MB of weights per CU / 27 k cycles = 45 B/clk per CU: the L2 -> CU stream of 256 CUs reading the same 614 KB.
//
// The same structure takes the other low-resolution blocks (template parameters S, NS; MbkItem below):
//   * stride 2 on a 32-wide map (96 -> 576 -> 160, 32x32 -> 16x16): the two input fragments of a row are its EVEN and ODD
//     columns, so output pixel fr finds its centre tap in its own lane of the even fragment, its right tap in its own lane of
//     the odd one and its left tap one lane down in the odd one; five input rows behind two output rows;
//   * 32-wide maps at stride 1 (64 -> 384 -> 64 | 96, 96 -> 576 -> 96 @32x32): two strips per row; a strip's first / last lane
//     takes its horizontal neighbour from the other strip's last / first lane with one extra DPP move (rotate into `old`,
//     shift with bound_ctrl off).
//
// Hazards (cdna_hip_programming.md 5.7): the ReLU6 packing is an inline-asm v_pk_mul_f32 ... clamp that reads MFMA results.
// hipcc does not pad the MFMA -> VALU-read wait states for an asm consumer (in ssdk_mbflow.hip / ssdk_mbsplit.hip the next
// chunk's MFMAs sit in between; here the pack follows its own chunk's last MFMA -- the first version read accumulators that
// had not been written yet: 10 - 40 % wrong outputs everywhere).  mbk_mfma_guard() puts the wait states there explicitly,
// fenced by sched_barriers; a shorter guard sits between the last asm clamp and the projection MFMAs that read its result.
//
// Numerics: those of ssdk_mbflow.hip / ssdk_mbsplit.hip (expand BN scale folded into the weights by the host, bias = the
// accumulator the MFMA starts from, both internal tensors fp16 in units of six with the VALU's [0, 1] clamp as ReLU6), the
// projection's fp32 sum formed as NW partial sums added in wave order.  Cout > 16 * NFO (160 -> 960 -> 320) runs as Cout /
// (16 NFO) column halves: each half repeats expand + depthwise (a third of the block's MFMA work) and owns its slice of the
// projection -- no second accumulator set, no exchange between halves.
#include "ssdk_common.h"
#include "ssdk_flow_common.h"
#include "ssdk_scan.h"  // lds_u8 / glb_u8 address-space typedefs

namespace ssdk {

struct MbkParams {
  const u16* x;
  u16* y;
  const unsigned char* img;  // the block's image (MbkGeo below; include/ssdk.h ssdk_mbconv_desc.w_image)
  int N, H, W, Ho, Wo, Cin, Chid, Cout, residual;
  int pairs, halves;         // row pairs per image = (Ho + 1) / 2, column halves = Cout / (16 NFO)
  unsigned items;            // halves * N * pairs
  unsigned long long* dbg;   // SSDK_MB_DBG=1: cycle stamps of wave 0 of workgroup 0
  unsigned wave_mask, pair_mask;  // DBG instances only (SSDK_MBK_MASKS=<waves>,<pairs>): which waves' / chunk pairs' projections count
};

// Image geometry (bytes).  [weights: halves x NW slices x NP pairs x (2 KS + NFO) KiB][constants: NW slices x MISC_KB KiB]
// [projection BN: halves x 2 KiB]
template <int KS, int NW, int NCHW, int NFO>
struct MbkGeo {
  static constexpr int NP = (NCHW + 1) / 2;
  static constexpr int PAIR_KB = 2 * KS + NFO;
  static constexpr int MISC_KB = (NCHW * 384 + 1023) / 1024;  // per slice: be [NCHW][4] f32x4 | wd [NCHW][9][4] 8 B | bd [NCHW][4] 8 B
  static constexpr int m_be = 0, m_wd = NCHW * 64, m_bd = NCHW * (64 + 288);
  static constexpr int SPB_KB = 2;                             // per half: [NFO][4 fg][sp * 6 f32x4 | bp f32x4]
  static_assert(NFO * 128 <= SPB_KB * 1024, "projection BN block");
  __host__ __device__ static size_t weights(int halves) { return (size_t)halves * NW * NP * PAIR_KB * 1024; }
  __host__ __device__ static size_t misc_off(int halves) { return weights(halves); }
  __host__ __device__ static size_t spb_off(int halves) { return misc_off(halves) + (size_t)NW * MISC_KB * 1024; }
  __host__ __device__ static size_t bytes(int halves) { return spb_off(halves) + (size_t)halves * SPB_KB * 1024; }
};

// Item geometry.  S: stride of the depthwise convolution; NS: 16-pixel strips per OUTPUT row (output width = 16 NS, input
// width = 16 NS S).  An item is two output rows: NR input rows of NXS fragments each.
//   S = 1: input fragment xs = columns 16 xs .. 16 xs + 15; the taps of strip j are the neighbouring lanes, the strip's first
//          / last lane takes the neighbouring strip's last / first lane (fl_from_left2 / fl_from_right2) or the zero padding;
//   S = 2: the two input fragments of output strip j are the EVEN (xs = 2j) and the ODD (2j + 1) columns of its 32 input
//          columns: output pixel fr has its centre tap in its own lane of the even fragment, its right tap in its own lane
//          of the odd one and its left tap one lane down in the odd one (ssdk_mbflow.hip's parity split).
template <int S, int NS>
struct MbkItem {
  static constexpr int NXS = NS * S;
  static constexpr int NR = S + 3;  // input rows behind two output rows: 4 | 5
};

// LDS image (bytes)
template <int S, int NS, int KS, int NW, int NCHW, int NFO>
struct MbkLds {
  using G = MbkGeo<KS, NW, NCHW, NFO>;
  using I = MbkItem<S, NS>;
  static constexpr int FT = 2 * NS * NFO;                       // output fragments of an item
  static constexpr int XF = FT % 5 == 0 ? 5 : (FT % 8 == 0 ? 8 : 4);  // fragments per exchange round
  static_assert(FT % XF == 0, "exchange rounds");
  static constexpr int xt = 0;                                  // [NR][NXS][KS][64 lanes] u32x4: B fragments of the input rows
  static constexpr int misc = xt + I::NR * I::NXS * KS * 1024;  // [NW][MISC_KB KiB]: the slices' constants, image layout
  static constexpr int spb = misc + NW * G::MISC_KB * 1024;     // [SPB_KB KiB]
  static constexpr int xch = spb + G::SPB_KB * 1024;            // [NW][XF][64 lanes] f32x4
  static constexpr int bytes = xch + NW * XF * 1024;
};

// wait states between an MFMA and an inline-asm VALU instruction that reads its result (8-pass XDL: 12 states; padded),
// and between an inline-asm VALU result and the MFMA that reads it as an operand -- hipcc pads neither (see the header)
__device__ __forceinline__ void mbk_mfma_guard() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mbk_valu_guard() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 3" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// neighbour pixel across a strip boundary: lanes 1 .. 15 take cur's left neighbour, lane 0 takes the last lane of the strip
// to the left (row_ror:1 of prev puts prev[15] into lane 0; row_shr:1 of cur with bound_ctrl off leaves lane 0 alone);
// mirrored for the right neighbour (row_ror:15 = rotate left by one puts next[0] into lane 15)
__device__ __forceinline__ u32 fl_from_left2(u32 cur, u32 prev) {
  const int old = __builtin_amdgcn_update_dpp(0, (int)prev, 0x121, 0xf, 0xf, true);
  return (u32)__builtin_amdgcn_update_dpp(old, (int)cur, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 fl_from_right2(u32 cur, u32 next) {
  const int old = __builtin_amdgcn_update_dpp(0, (int)next, 0x12f, 0xf, 0xf, true);
  return (u32)__builtin_amdgcn_update_dpp(old, (int)cur, 0x101, 0xf, 0xf, false);
}

template <int DT, int S, int NS, int KS, int NW, int NCHW, int NFO, bool DBG = false>
__global__ __launch_bounds__(64 * NW, 2) void mbk_kernel(const MbkParams p) {
  using L = MbkLds<S, NS, KS, NW, NCHW, NFO>;
  using G = MbkGeo<KS, NW, NCHW, NFO>;
  using I = MbkItem<S, NS>;
  constexpr int NR = I::NR, NXS = I::NXS, NE = NR * NXS;  // input rows, fragments per row, expanded fragments per chunk
  constexpr int XF = L::XF, FT = L::FT, NP = G::NP, PAIR_KB = G::PAIR_KB;
  constexpr int NO = 2 * NS;                                  // output fragments per channel fragment: (row o, strip j) = o NS + j
  constexpr bool ODD = (NCHW & 1) != 0;       // the last pair holds ONE chunk (its second half is zero in the image)
  constexpr int ROUNDS = FT / XF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));  // this wave's slice of the hidden channels
  const int Cin = p.Cin, Cout = p.Cout, H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;

  // ---- the workgroup's item: (image, pair of output rows, column half), half-minor ---------------------------------------
  const u32 item = blockIdx.x;
  const int half = (int)(item % (u32)p.halves);
  const u32 rem = item / (u32)p.halves;
  const int n = (int)(rem / (u32)p.pairs), rp = (int)(rem % (u32)p.pairs);
  const int oy0 = 2 * rp;
  const int iy0 = S * oy0 - 1;  // input row of the item's row 0
  const int co_base = half * 16 * NFO;
  const u16* ximg = p.x + (size_t)n * H * W * Cin;
  const bool stamp = p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) p.dbg[0] = __builtin_readcyclecounter();

  // ---- this wave's weight stream: one pointer per lane, fragments 1 KiB apart -----------------------------------------
  const u32x4* wimg = reinterpret_cast<const u32x4*>(p.img + ((size_t)(half * NW + wv) * NP) * ((size_t)PAIR_KB * 1024)) + lane;
  u32x4 wa[KS], wb[KS];  // expand A fragments of the pair's two chunks (an ODD slice's missing chunk: zeros, never used)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    wa[ks] = wimg[(size_t)ks * 64];
    wb[ks] = wimg[(size_t)(KS + ks) * 64];
  }

  // ---- LDS-DMA: block input of the item as B fragments (row r <- input row iy0 + r, clamped into the image: a row outside
  // it is multiplied by 0 when it is packed, so it only has to be finite), the slices' constants, the projection BN ---------
  for (int e = wv; e < NE; e += NW) {  // wave-uniform: fragment e = (row r, fragment xs of the row)
    const int r = e / NXS, xs = e % NXS;
    int iy = iy0 + r;
    iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int col = S == 1 ? 16 * xs + (int)fr : 2 * (16 * (xs >> 1) + (int)fr) + (xs & 1);
    const u16* src = ximg + ((size_t)iy * W + col) * Cin + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + ks * 32), (lds_u8*)(smem + L::xt + (e * KS + ks) * 1024), 16, 0, 0);
  }
  {
    const unsigned char* src = p.img + G::misc_off(p.halves) + (size_t)wv * (G::MISC_KB * 1024) + lane * 16;
#pragma unroll
    for (int k = 0; k < G::MISC_KB; ++k)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + k * 1024), (lds_u8*)(smem + L::misc + (wv * G::MISC_KB + k) * 1024), 16, 0, 0);
  }
  if (wv == 0) {
    const unsigned char* src = p.img + G::spb_off(p.halves) + (size_t)half * (G::SPB_KB * 1024) + lane * 16;
#pragma unroll
    for (int k = 0; k < G::SPB_KB; ++k)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + k * 1024), (lds_u8*)(smem + L::spb + k * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA has landed (and its first weight fragments)
  __syncthreads();
  if (stamp) p.dbg[1] = __builtin_readcyclecounter();

  fl_f2 hi[NR];  // 1/6, or 0 for a row outside the image (the zero padding of the EXPANDED tensor)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float k = ((unsigned)(iy0 + r) < (unsigned)H) ? kFlSixth : 0.f;
    hi[r] = fl_f2{k, k};
  }

  f32x4 yacc[NO][NFO];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int f = 0; f < NFO; ++f) yacc[o][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* xt = smem + L::xt + lane * 16;
  const unsigned char* mc = smem + L::misc + wv * (G::MISC_KB * 1024);  // this slice's constants

  // one chunk: expand its NE fragments from the BN bias, pack (ReLU6 in units of six), depthwise -> the item's NO output
  // fragments' channels 16c + 4fg .. +3 as two packed words each
  auto chunk = [&](int c, const u32x4 (&w)[KS], u32 (&dout)[NO][2]) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(mc + G::m_be + (c * 4 + (int)fg) * 16);
    f32x4 e[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = bv;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 xf[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) xf[i] = *reinterpret_cast<const u32x4*>(xt + (i * KS + ks) * 1024);
#pragma unroll
      for (int i = 0; i < NE; ++i) e[i] = fl_mfma<DT>(w[ks], xf[i], e[i]);  // D[hc = 16c + 4fg + q][px = fr]
    }
    uint2 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint2*>(mc + G::m_wd + ((c * 9 + t) * 4 + (int)fg) * 8);
    const uint2 bdi = *reinterpret_cast<const uint2*>(mc + G::m_bd + (c * 4 + (int)fg) * 8);
    mbk_mfma_guard();  // the accumulators are read by inline asm next
    u32 ew[NR][NXS][2];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int xs = 0; xs < NXS; ++xs) {
        ew[r][xs][0] = fl_unit_pack(e[r * NXS + xs][0], e[r * NXS + xs][1], hi[r]);
        ew[r][xs][1] = fl_unit_pack(e[r * NXS + xs][2], e[r * NXS + xs][3], hi[r]);
      }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        fl_h2 s[2] = {fl_as_h2(bdi.x), fl_as_h2(bdi.y)};  // the depthwise bias is the value the sum starts from
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int r = S * o + ky;
          const uint2 w0 = wt[ky * 3], w1 = wt[ky * 3 + 1], w2 = wt[ky * 3 + 2];
#pragma unroll
          for (int wd = 0; wd < 2; ++wd) {
            u32 tl, tc, tr;  // left / centre / right tap of output pixel (o, 16 j + fr) in input row r
            if constexpr (S == 1) {
              tc = ew[r][j][wd];
              tl = j == 0 ? fl_from_left(tc) : fl_from_left2(tc, ew[r][j > 0 ? j - 1 : 0][wd]);
              tr = j == NS - 1 ? fl_from_right(tc) : fl_from_right2(tc, ew[r][j < NS - 1 ? j + 1 : j][wd]);
            } else {
              const u32 odd = ew[r][2 * j + 1][wd];
              tc = ew[r][2 * j][wd];
              tl = j == 0 ? fl_from_left(odd) : fl_from_left2(odd, ew[r][j > 0 ? 2 * j - 1 : 1][wd]);
              tr = odd;
            }
            const u32 ww0 = wd ? w0.y : w0.x, ww1 = wd ? w1.y : w1.x, ww2 = wd ? w2.y : w2.x;
            s[wd] = __builtin_elementwise_fma(fl_as_h2(tl), fl_as_h2(ww0), s[wd]);
            s[wd] = __builtin_elementwise_fma(fl_as_h2(tc), fl_as_h2(ww1), s[wd]);
            // the output row is complete with the last tap: ReLU6 = the clamp of the FMA (units of six)
            if (ky == 2) s[wd] = fl_fma_clamp01(fl_as_h2(tr), fl_as_h2(ww2), s[wd]);
            else s[wd] = __builtin_elementwise_fma(fl_as_h2(tr), fl_as_h2(ww2), s[wd]);
          }
        }
        dout[o * NS + j][0] = fl_as_u32(s[0]);
        dout[o * NS + j][1] = fl_as_u32(s[1]);
      }
  };

  // ---- main loop over this wave's chunk pairs: no barrier, no other wave's data.  Every weight fragment is requested about
  // one chunk phase (~1.5 k cycles) before the MFMA that reads it: half of the pair's projection fragments at its top and
  // half behind its first chunk, the next pair's first chunk behind this pair's first, the next pair's second chunk behind
  // this pair's second.  (The first version requested half of the projection fragments and the next chunk right in front of
  // the projection MFMAs: two exposed L2 round trips per pair, 4.1 k cycles per pair for 1.9 k of MFMA time per SIMD.) ------
  constexpr int FH = NFO / 2;
  for (int t = 0; t < NP; ++t) {
    const u32x4* wp = wimg + (size_t)t * (PAIR_KB * 64);
    const bool last = t == NP - 1;
    const u32x4* wn = wimg + (size_t)(last ? t : t + 1) * (PAIR_KB * 64);  // (past the end: the last pair again, harmless)
    u32 d0[NO][2], d1[NO][2];
    u32x4 wf0[FH], wf1[NFO - FH];
#pragma unroll
    for (int f = 0; f < FH; ++f) wf0[f] = wp[(size_t)(2 * KS + f) * 64];
    chunk(2 * t, wa, d0);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 na[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) na[ks] = wn[(size_t)ks * 64];
#pragma unroll
    for (int f = FH; f < NFO; ++f) wf1[f - FH] = wp[(size_t)(2 * KS + f) * 64];
    if (!(ODD && last)) {
      chunk(2 * t + 1, wb, d1);
    } else {
#pragma unroll
      for (int o = 0; o < NO; ++o) d1[o][0] = d1[o][1] = 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 nb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) nb[ks] = wn[(size_t)(KS + ks) * 64];
    // projection k-step t: the pair's 32 hidden channels ARE the B operand (k-step element j <-> chunk 2t + j/4, channel
    // 4fg + j%4: the permutation the image applies to the projection weights)
    u32x4 db[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) db[o] = u32x4{d0[o][0], d0[o][1], d1[o][0], d1[o][1]};
    if constexpr (DBG) {
      if (!((p.pair_mask >> t) & 1u) || !((p.wave_mask >> wv) & 1u)) {
#pragma unroll
        for (int o = 0; o < NO; ++o) db[o] = u32x4{0u, 0u, 0u, 0u};
      }
    }
    mbk_valu_guard();  // db comes out of inline asm (the clamping FMA)
#pragma unroll
    for (int f = 0; f < NFO; ++f)
#pragma unroll
      for (int o = 0; o < NO; ++o)
        yacc[o][f] = fl_mfma<SSDK_F16>(f < FH ? wf0[f < FH ? f : 0] : wf1[f >= FH ? f - FH : 0], db[o], yacc[o][f]);  // D[co = 16f + 4fg + q][px = fr]
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wa[ks] = na[ks];
      wb[ks] = nb[ks];
    }
  }
  if (stamp) p.dbg[2] = __builtin_readcyclecounter();

  // ---- exchange: XF fragments per round (fragment q = (o NS + j) NFO + f).  Every wave leaves its partial sums, one barrier,
  // the owner of a fragment adds them in wave order 0 .. NW-1, applies the projection BN (+ residual) and stores; one more
  // barrier frees the buffer ---------------------------------------------------------------------------------------------------
  unsigned char* xb = smem + L::xch;
  const unsigned char* spb = smem + L::spb;
  // the residual values of the fragments this wave will finalize, requested before the first round (they were one exposed
  // memory round trip per round)
  constexpr int GPW = (XF + NW - 1) / NW;  // fragments a wave finalizes per round
  uint2 resv[ROUNDS][GPW];
#pragma unroll
  for (int R = 0; R < ROUNDS; ++R)
#pragma unroll
    for (int gg = 0; gg < GPW; ++gg) {
      const int g = wv + gg * NW, q = R * XF + g, oj = q / NFO, f = q % NFO;
      const int oy = oy0 + oj / NS, ox = 16 * (oj % NS) + (int)fr, co = co_base + f * 16 + (int)fg * 4;
      resv[R][gg] = make_uint2(0u, 0u);
      if (S == 1 && p.residual && g < XF && oy < Ho && co < Cout)
        resv[R][gg] = *reinterpret_cast<const uint2*>(ximg + ((size_t)oy * W + ox) * Cin + co);
    }
#pragma unroll
  for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
    for (int g = 0; g < XF; ++g) {
      const int q = R * XF + g, oj = q / NFO, f = q % NFO;  // compile-time after unrolling
      *reinterpret_cast<f32x4*>(xb + ((wv * XF + g) * 64 + (int)lane) * 16) = yacc[oj][f];
    }
    __syncthreads();
#pragma unroll
    for (int gg = 0; gg < GPW; ++gg) {
      const int g = wv + gg * NW;  // wave-uniform
      if (g < XF) {
        const int q = R * XF + g, oj = q / NFO, f = q % NFO;
        f32x4 y = *reinterpret_cast<const f32x4*>(xb + ((0 * XF + g) * 64 + (int)lane) * 16);
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
          const f32x4 t2 = *reinterpret_cast<const f32x4*>(xb + ((w2 * XF + g) * 64 + (int)lane) * 16);
          y = y + t2;
        }
        const int oy = oy0 + oj / NS, ox = 16 * (oj % NS) + (int)fr;
        const int co = co_base + f * 16 + (int)fg * 4;
        if (oy < Ho && co < Cout) {
          const f32x4 spv = *reinterpret_cast<const f32x4*>(spb + (f * 4 + (int)fg) * 32);
          const f32x4 bpv = *reinterpret_cast<const f32x4*>(spb + (f * 4 + (int)fg) * 32 + 16);
          u32 h01 = fl_pack2<DT>(fmaf(y[0], spv[0], bpv[0]), fmaf(y[1], spv[1], bpv[1]));
          u32 h23 = fl_pack2<DT>(fmaf(y[2], spv[2], bpv[2]), fmaf(y[3], spv[3], bpv[3]));
          if (S == 1 && p.residual) {  // rounded to the model dtype first, then x is added (torch's tensor add)
            const uint2 xr = resv[R][gg];
            h01 = fl_pack2<DT>(fl_from16<DT>(h01 & 0xffffu) + fl_from16<DT>(xr.x & 0xffffu), fl_from16<DT>(h01 >> 16) + fl_from16<DT>(xr.x >> 16));
            h23 = fl_pack2<DT>(fl_from16<DT>(h23 & 0xffffu) + fl_from16<DT>(xr.y & 0xffffu), fl_from16<DT>(h23 >> 16) + fl_from16<DT>(xr.y >> 16));
          }
          u16* yrow = p.y + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
          *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h01, h23);
        }
      }
    }
    if (R + 1 < ROUNDS) __syncthreads();
  }
  if (stamp) p.dbg[3] = __builtin_readcyclecounter();
}

// ---- host side -------------------------------------------------------------------------------------------------------------
template <int DT, int S, int NS, int KS, int NW, int NCHW, int NFO, bool DBG = false>
static void mbk_launch(const MbkParams& p, hipStream_t stream) {
  constexpr int lds = MbkLds<S, NS, KS, NW, NCHW, NFO>::bytes;
  static_assert(lds <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbk_kernel<DT, S, NS, KS, NW, NCHW, NFO, DBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((mbk_kernel<DT, S, NS, KS, NW, NCHW, NFO, DBG>), dim3(p.items), dim3(64 * NW), lds, stream, p);
}

// The instantiations: (S, NS, KS, NCHW, NFO), all with NW = 4 slices -- the blocks of MobileNetV2 from the 32x32 maps down:
//   1 1 5 15 10   160 -> 960 -> 160 | 320 @16x16        1 2 2  6 4    64 -> 384 -> 64 @32x32      1 2 3 9 6   96 -> 576 -> 96 @32x32
//   2 1 3  9 10    96 -> 576 -> 160 @32x32 -> 16x16     1 2 2  6 6    64 -> 384 -> 96 @32x32
//   2 2 1  3  4    32 -> 192 -> 64 @64x64 -> 32x32      1 4 1  3 2    32 -> 192 -> 32 @64x64
constexpr int kMbkNW = 4;
struct MbkInst {
  int s, ns, ks, nchw, nfo;
};
static const MbkInst kMbkInst[] = {{1, 1, 5, 15, 10}, {2, 1, 3, 9, 10}, {1, 2, 2, 6, 4}, {1, 2, 2, 6, 6}, {1, 2, 3, 9, 6},
                                   {2, 2, 1, 3, 4}, {1, 4, 1, 3, 2}};

// -> index into kMbkInst, or -1.  Wo: output width.
static int mbk_find(int stride, int Wo, int Cin, int Chid, int Cout, int nw, int* halves) {
  if (nw != kMbkNW || Cin % 32 || Chid % 16 || Cout % 16 || Wo % 16) return -1;
  const int ks = Cin / 32, nch = Chid / 16, nchw = (nch + nw - 1) / nw, ns = Wo / 16;
  for (int i = 0; i < (int)(sizeof(kMbkInst) / sizeof(kMbkInst[0])); ++i) {
    const MbkInst& k = kMbkInst[i];
    if (k.s == stride && k.ns == ns && k.ks == ks && k.nchw == nchw && Cout % (16 * k.nfo) == 0) {
      if (halves) *halves = Cout / (16 * k.nfo);
      return i;
    }
  }
  return -1;
}

static size_t mbk_bytes(int inst, int halves) {
  switch (inst) {
    case 0: return MbkGeo<5, kMbkNW, 15, 10>::bytes(halves);
    case 1: return MbkGeo<3, kMbkNW, 9, 10>::bytes(halves);
    case 2: return MbkGeo<2, kMbkNW, 6, 4>::bytes(halves);
    case 3: return MbkGeo<2, kMbkNW, 6, 6>::bytes(halves);
    case 4: return MbkGeo<3, kMbkNW, 9, 6>::bytes(halves);
    case 5: return MbkGeo<1, kMbkNW, 3, 4>::bytes(halves);
    case 6: return MbkGeo<1, kMbkNW, 3, 2>::bytes(halves);
  }
  return 0;
}

template <int DT>
static void mbk_dispatch(const MbkParams& p, int inst, hipStream_t stream) {
  const bool dbg = p.wave_mask != ~0u || p.pair_mask != ~0u;
  switch (inst) {
    case 0: dbg ? mbk_launch<DT, 1, 1, 5, kMbkNW, 15, 10, true>(p, stream) : mbk_launch<DT, 1, 1, 5, kMbkNW, 15, 10>(p, stream); break;
    case 1: mbk_launch<DT, 2, 1, 3, kMbkNW, 9, 10>(p, stream); break;
    case 2: mbk_launch<DT, 1, 2, 2, kMbkNW, 6, 4>(p, stream); break;
    case 3: mbk_launch<DT, 1, 2, 2, kMbkNW, 6, 6>(p, stream); break;
    case 4: mbk_launch<DT, 1, 2, 3, kMbkNW, 9, 6>(p, stream); break;
    case 5: mbk_launch<DT, 2, 2, 1, kMbkNW, 3, 4>(p, stream); break;
    case 6: mbk_launch<DT, 1, 4, 1, kMbkNW, 3, 2>(p, stream); break;
  }
}

// Returns 1 when the block is not one of this kernel's (the caller then runs ssdk_mbconv.hip's), 0 after a launch.
int launch_mbk(const ssdk_mbconv_desc* d, hipStream_t stream) {
  static const int env = getenv("SSDK_MBK") ? atoi(getenv("SSDK_MBK")) : 1;  // bit mask over the instances (1 = all)
  const int variant = d->variant;  // 0 auto, 3 this kernel wherever it exists (tests), other non-zero values: never
  if ((!env && variant != 3) || (variant != 0 && variant != 3)) return 1;
  if (d->stem || !d->w_image || d->image_nw < 1 || (d->stride != 1 && d->stride != 2)) return 1;
  const int Ho = (d->H + 2 - 3) / d->stride + 1, Wo = (d->W + 2 - 3) / d->stride + 1;
  if (d->W != Wo * d->stride) return 1;  // (the strips tile the row exactly: 16 | 32 columns, an even width for stride 2)
  int halves = 0;
  const int inst = mbk_find(d->stride, Wo, d->Cin, d->Chid, d->Cout, d->image_nw, &halves);
  if (inst < 0 || (size_t)d->w_image_bytes < mbk_bytes(inst, halves) || ((uintptr_t)d->w_image & 15)) return 1;
  if (env != 1 && variant != 3 && !((env >> (inst + 1)) & 1)) return 1;  // (A/B: SSDK_MBK = 2 << instance, or-ed)
  const int pairs = (Ho + 1) / 2;
  const long items = (long)halves * d->N * pairs;
  constexpr int env_min = 256;  // (round 6: the SSDK_MBK_MIN switch is gone, its A/B is settled)
  if (items < env_min && variant != 3) return 1;  // a handful of items cannot fill the chip: the tiled kernel's 8x8 tiles can
  MbkParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  p.img = (const unsigned char*)d->w_image;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.Ho = Ho;
  p.Wo = Wo;
  p.Cin = d->Cin;
  p.Chid = d->Chid;
  p.Cout = d->Cout;
  p.residual = d->residual;
  p.pairs = pairs;
  p.halves = halves;
  p.items = (unsigned)items;
  p.dbg = nullptr;
  p.wave_mask = p.pair_mask = ~0u;
  if (const char* m = getenv("SSDK_MBK_MASKS")) {  // (debug, read per call) "<wave mask>,<pair mask>", hexadecimal
    unsigned a = ~0u, b = ~0u;
    if (sscanf(m, "%x,%x", &a, &b) == 2) {
      p.wave_mask = a;
      p.pair_mask = b;
    }
  }
  static const int env_dbg = getenv("SSDK_MB_DBG") ? atoi(getenv("SSDK_MB_DBG")) : 0;
  static unsigned long long* dbg_dev = nullptr;
  if (env_dbg) {
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(unsigned long long));
    p.dbg = dbg_dev;
  }
  if (d->dtype == SSDK_BF16) mbk_dispatch<SSDK_BF16>(p, inst, stream);
  else mbk_dispatch<SSDK_F16>(p, inst, stream);
  if (env_dbg) {  // debug only: synchronises
    unsigned long long h[4];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[mbk dbg] Cin=%d Chid=%d Cout=%d s=%d W=%d items=%ld : setup %llu loop %llu exchange %llu\n", d->Cin, d->Chid,
            d->Cout, d->stride, d->W, items, h[1] - h[0], h[2] - h[1], h[3] - h[2]);
  }
  return 0;
}

}  // namespace ssdk

// Bytes of the image of a block on a map whose OUTPUT is `Wo` pixels wide, or 0 when no instance of the kernel takes it.
extern "C" size_t ssdk_mbk_image_bytes(int Cin, int Chid, int Cout, int stride, int Wo, int nw, int* nfo) {
  int halves = 0;
  const int inst = ssdk::mbk_find(stride, Wo, Cin, Chid, Cout, nw, &halves);
  if (inst < 0) return 0;
  if (nfo) *nfo = ssdk::kMbkInst[inst].nfo;  // column fragments per half: the image's projection part is built for it
  return ssdk::mbk_bytes(inst, halves);
}
