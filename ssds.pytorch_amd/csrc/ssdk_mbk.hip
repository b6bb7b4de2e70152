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
//   * ITEMS = 2: a workgroup of 2 NW waves runs TWO items whose waves (i, w), i = 0 | 1, read the same weight fragments at
//     the same time -- the two column halves of one row pair (Cout = 320: they also share the input tile) or two row pairs
//     (Cout = 160).  Measured with one item per workgroup and two workgroups per CU: the loop time is exactly weight bytes per
//     CU / 28 B/clk (all 256 CUs stream the same 614 KB from L2); with the pair in lock step the second wave's request finds
//     the line in the CU's L1.
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
  int N, H, Cin, Chid, Cout, residual;
  int pairs, halves;         // row pairs per image = (H + 1) / 2, column halves = Cout / (16 NFO)
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

// LDS image (bytes)
template <int KS, int NW, int NCHW, int NFO, int ITEMS>
struct MbkLds {
  using G = MbkGeo<KS, NW, NCHW, NFO>;
  static constexpr int NR = 4;                                  // expanded rows per item
  static constexpr int XF = 5;                                  // fragments per exchange round
  static constexpr int xt = 0;                                  // [ITEMS][NR][KS][64 lanes] u32x4: B fragments of the input rows
  static constexpr int misc = xt + ITEMS * NR * KS * 1024;      // [NW][MISC_KB KiB]: the slices' constants, image layout
  static constexpr int spb = misc + NW * G::MISC_KB * 1024;     // [ITEMS][SPB_KB KiB]
  static constexpr int xch = spb + ITEMS * G::SPB_KB * 1024;    // [ITEMS][NW][XF][64 lanes] f32x4
  static constexpr int bytes = xch + ITEMS * NW * XF * 1024;
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

template <int DT, int KS, int NW, int NCHW, int NFO, int ITEMS, bool DBG = false>
__global__ __launch_bounds__(64 * NW * ITEMS, 2) void mbk_kernel(const MbkParams p) {
  using L = MbkLds<KS, NW, NCHW, NFO, ITEMS>;
  using G = MbkGeo<KS, NW, NCHW, NFO>;
  constexpr int NR = L::NR, XF = L::XF, NP = G::NP, PAIR_KB = G::PAIR_KB;
  constexpr bool ODD = (NCHW & 1) != 0;       // the last pair holds ONE chunk (its second half is zero in the image)
  static_assert((2 * NFO) % XF == 0, "exchange rounds");
  constexpr int ROUNDS = 2 * NFO / XF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int wvg = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int it = ITEMS == 1 ? 0 : wvg / NW;   // this wave's item of the workgroup
  const int wv = ITEMS == 1 ? wvg : wvg % NW; // ... and its slice of the hidden channels
  const int Cin = p.Cin, Cout = p.Cout, H = p.H;
  constexpr int W = 16;

  // ---- this wave's item: (image, pair of output rows, column half).  Items are numbered half-minor, so the two items of a
  // workgroup are the two halves of one row pair (Cout = 320) or two consecutive row pairs (Cout = 160) ----------------------
  const u32 item = blockIdx.x * ITEMS + (u32)it;  // (the host launches items / ITEMS workgroups: items is a multiple of ITEMS)
  const int half = (int)(item % (u32)p.halves);
  const u32 rem = item / (u32)p.halves;
  const int n = (int)(rem / (u32)p.pairs), rp = (int)(rem % (u32)p.pairs);
  const int oy0 = 2 * rp;
  const int co_base = half * 16 * NFO;
  const u16* ximg = p.x + (size_t)n * H * W * Cin;
  const bool stamp = p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) p.dbg[0] = __builtin_readcyclecounter();

  // ---- this wave's weight stream: one pointer per lane, fragments 1 KiB apart -----------------------------------------
  const u32x4* wimg = reinterpret_cast<const u32x4*>(p.img + ((size_t)(half * NW + wv) * NP) * ((size_t)PAIR_KB * 1024)) + lane;
  u32x4 wa[KS];  // expand A fragments of the NEXT chunk to be expanded (prefetched one chunk ahead)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wa[ks] = wimg[(size_t)ks * 64];

  // ---- LDS-DMA: block input of the item as B fragments (row r of the item <- input row oy0 - 1 + r, clamped into the image:
  // a row outside it is multiplied by 0 when it is packed, so it only has to be finite), the slices' constants (item 0's
  // waves), the item's projection BN (slice 0's wave) ------------------------------------------------------------------------
  unsigned char* xt_item = smem + L::xt + it * (NR * KS * 1024);
  for (int r = wv; r < NR; r += NW) {  // wave-uniform
    int iy = oy0 - 1 + r;
    iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const u16* src = ximg + ((size_t)iy * W + fr) * Cin + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + ks * 32), (lds_u8*)(xt_item + (r * KS + ks) * 1024), 16, 0, 0);
  }
  if (it == 0) {
    const unsigned char* src = p.img + G::misc_off(p.halves) + (size_t)wv * (G::MISC_KB * 1024) + lane * 16;
#pragma unroll
    for (int k = 0; k < G::MISC_KB; ++k)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + k * 1024), (lds_u8*)(smem + L::misc + (wv * G::MISC_KB + k) * 1024), 16, 0, 0);
  }
  if (wv == 0) {
    const unsigned char* src = p.img + G::spb_off(p.halves) + (size_t)half * (G::SPB_KB * 1024) + lane * 16;
#pragma unroll
    for (int k = 0; k < G::SPB_KB; ++k)
      __builtin_amdgcn_global_load_lds((glb_u8*)(src + k * 1024), (lds_u8*)(smem + L::spb + (it * G::SPB_KB + k) * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA has landed (and its first weight fragments)
  __syncthreads();
  if (stamp) p.dbg[1] = __builtin_readcyclecounter();

  fl_f2 hi[NR];  // 1/6, or 0 for a row outside the image (the zero padding of the EXPANDED tensor)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float k = ((unsigned)(oy0 - 1 + r) < (unsigned)H) ? kFlSixth : 0.f;
    hi[r] = fl_f2{k, k};
  }

  f32x4 yacc[2][NFO];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int f = 0; f < NFO; ++f) yacc[o][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* xt = xt_item + lane * 16;
  const unsigned char* mc = smem + L::misc + wv * (G::MISC_KB * 1024);  // this slice's constants

  // one chunk: expand its four rows from the BN bias, pack (ReLU6 in units of six), depthwise -> the two output rows'
  // channels 16c + 4fg .. +3 as two packed words per row
  auto chunk = [&](int c, const u32x4 (&w)[KS], u32 (&dout)[2][2]) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(mc + G::m_be + (c * 4 + (int)fg) * 16);
    f32x4 e[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) e[r] = bv;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 xf[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) xf[r] = *reinterpret_cast<const u32x4*>(xt + (r * KS + ks) * 1024);
#pragma unroll
      for (int r = 0; r < NR; ++r) e[r] = fl_mfma<DT>(w[ks], xf[r], e[r]);  // D[hc = 16c + 4fg + q][px = fr]
    }
    uint2 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint2*>(mc + G::m_wd + ((c * 9 + t) * 4 + (int)fg) * 8);
    const uint2 bdi = *reinterpret_cast<const uint2*>(mc + G::m_bd + (c * 4 + (int)fg) * 8);
    mbk_mfma_guard();  // the accumulators are read by inline asm next
    u32 ew[NR][2];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      ew[r][0] = fl_unit_pack(e[r][0], e[r][1], hi[r]);
      ew[r][1] = fl_unit_pack(e[r][2], e[r][3], hi[r]);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      fl_h2 s0 = fl_as_h2(bdi.x), s1 = fl_as_h2(bdi.y);  // the depthwise bias is the value the sum starts from
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const u32 c0 = ew[o + ky][0], c1 = ew[o + ky][1];
        const uint2 w0 = wt[ky * 3], w1 = wt[ky * 3 + 1], w2 = wt[ky * 3 + 2];
        s0 = __builtin_elementwise_fma(fl_as_h2(fl_from_left(c0)), fl_as_h2(w0.x), s0);
        s1 = __builtin_elementwise_fma(fl_as_h2(fl_from_left(c1)), fl_as_h2(w0.y), s1);
        s0 = __builtin_elementwise_fma(fl_as_h2(c0), fl_as_h2(w1.x), s0);
        s1 = __builtin_elementwise_fma(fl_as_h2(c1), fl_as_h2(w1.y), s1);
        if (ky == 2) {  // the output row is complete with this tap: ReLU6 = the clamp of the FMA (units of six)
          s0 = fl_fma_clamp01(fl_as_h2(fl_from_right(c0)), fl_as_h2(w2.x), s0);
          s1 = fl_fma_clamp01(fl_as_h2(fl_from_right(c1)), fl_as_h2(w2.y), s1);
        } else {
          s0 = __builtin_elementwise_fma(fl_as_h2(fl_from_right(c0)), fl_as_h2(w2.x), s0);
          s1 = __builtin_elementwise_fma(fl_as_h2(fl_from_right(c1)), fl_as_h2(w2.y), s1);
        }
      }
      dout[o][0] = fl_as_u32(s0);
      dout[o][1] = fl_as_u32(s1);
    }
  };

  // ---- main loop over this wave's chunk pairs: no barrier, no other wave's data.  Every weight fragment is requested about
  // one chunk phase (~1.5 k cycles) before the MFMA that reads it: the pair's ten projection fragments at its top, the next
  // pair's first chunk behind this pair's first, the next pair's second chunk behind this pair's second.  (The first version
  // requested half of the projection fragments and the next chunk right in front of the projection MFMAs: two exposed L2
  // round trips per pair, 4.1 k cycles per pair for 1.9 k of MFMA time per SIMD.) -------------------------------------------
  u32x4 wb[KS];  // expand A fragments of the pair's second chunk (an ODD slice's missing chunk: zeros, never used)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wb[ks] = wimg[(size_t)(KS + ks) * 64];
  for (int t = 0; t < NP; ++t) {
    const u32x4* wp = wimg + (size_t)t * (PAIR_KB * 64);
    const bool last = t == NP - 1;
    const u32x4* wn = wimg + (size_t)(last ? t : t + 1) * (PAIR_KB * 64);  // (past the end: the last pair again, harmless)
    u32 d0[2][2], d1[2][2];
    u32x4 wf[NFO];
#pragma unroll
    for (int f = 0; f < NFO; ++f) wf[f] = wp[(size_t)(2 * KS + f) * 64];
    chunk(2 * t, wa, d0);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 na[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) na[ks] = wn[(size_t)ks * 64];
    if (!(ODD && last)) {
      chunk(2 * t + 1, wb, d1);
    } else {
#pragma unroll
      for (int o = 0; o < 2; ++o) d1[o][0] = d1[o][1] = 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 nb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) nb[ks] = wn[(size_t)(KS + ks) * 64];
    // projection k-step t: the pair's 32 hidden channels ARE the B operand (k-step element j <-> chunk 2t + j/4, channel
    // 4fg + j%4: the permutation the image applies to the projection weights)
    u32x4 db[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) db[o] = u32x4{d0[o][0], d0[o][1], d1[o][0], d1[o][1]};
    if constexpr (DBG) {
      if (!((p.pair_mask >> t) & 1u) || !((p.wave_mask >> wv) & 1u)) {
#pragma unroll
        for (int o = 0; o < 2; ++o) db[o] = u32x4{0u, 0u, 0u, 0u};
      }
    }
    mbk_valu_guard();  // db comes out of inline asm (the clamping FMA)
#pragma unroll
    for (int f = 0; f < NFO; ++f)
#pragma unroll
      for (int o = 0; o < 2; ++o) yacc[o][f] = fl_mfma<SSDK_F16>(wf[f], db[o], yacc[o][f]);  // D[co = 16f + 4fg + q][px = fr]
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wa[ks] = na[ks];
      wb[ks] = nb[ks];
    }
  }
  if (stamp) p.dbg[2] = __builtin_readcyclecounter();

  // ---- exchange: XF fragments per round.  Every wave leaves its partial sums, one barrier, the owner of a fragment adds
  // them in wave order 0 .. NW-1, applies the projection BN (+ residual) and stores; one more barrier frees the buffer ----
  unsigned char* xb = smem + L::xch + it * (NW * XF * 1024);
  const unsigned char* spb = smem + L::spb + it * (G::SPB_KB * 1024);
  // the residual values of the fragments this wave will finalize, requested before the first round (they were one exposed
  // memory round trip per round)
  constexpr int GPW = (XF + NW - 1) / NW;  // fragments a wave finalizes per round
  uint2 resv[ROUNDS][GPW];
#pragma unroll
  for (int R = 0; R < ROUNDS; ++R)
#pragma unroll
    for (int gg = 0; gg < GPW; ++gg) {
      const int g = wv + gg * NW, q = R * XF + g, o = q / NFO, f = q % NFO;
      const int oy = oy0 + o, co = co_base + f * 16 + (int)fg * 4;
      resv[R][gg] = make_uint2(0u, 0u);
      if (p.residual && g < XF && oy < H && co < Cout) resv[R][gg] = *reinterpret_cast<const uint2*>(ximg + ((size_t)oy * W + fr) * Cin + co);
    }
#pragma unroll
  for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
    for (int g = 0; g < XF; ++g) {
      const int q = R * XF + g, o = q / NFO, f = q % NFO;  // compile-time after unrolling
      *reinterpret_cast<f32x4*>(xb + ((wv * XF + g) * 64 + (int)lane) * 16) = yacc[o][f];
    }
    __syncthreads();
#pragma unroll
    for (int gg = 0; gg < GPW; ++gg) {
      const int g = wv + gg * NW;  // wave-uniform
      if (g < XF) {
        const int q = R * XF + g, o = q / NFO, f = q % NFO;
        f32x4 y = *reinterpret_cast<const f32x4*>(xb + ((0 * XF + g) * 64 + (int)lane) * 16);
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
          const f32x4 t2 = *reinterpret_cast<const f32x4*>(xb + ((w2 * XF + g) * 64 + (int)lane) * 16);
          y = y + t2;
        }
        const int oy = oy0 + o;
        const int co = co_base + f * 16 + (int)fg * 4;
        if (oy < H && co < Cout) {
          const f32x4 spv = *reinterpret_cast<const f32x4*>(spb + (f * 4 + (int)fg) * 32);
          const f32x4 bpv = *reinterpret_cast<const f32x4*>(spb + (f * 4 + (int)fg) * 32 + 16);
          u32 h01 = fl_pack2<DT>(fmaf(y[0], spv[0], bpv[0]), fmaf(y[1], spv[1], bpv[1]));
          u32 h23 = fl_pack2<DT>(fmaf(y[2], spv[2], bpv[2]), fmaf(y[3], spv[3], bpv[3]));
          if (p.residual) {  // rounded to the model dtype first, then x is added (torch's tensor add)
            const uint2 xr = resv[R][gg];
            h01 = fl_pack2<DT>(fl_from16<DT>(h01 & 0xffffu) + fl_from16<DT>(xr.x & 0xffffu), fl_from16<DT>(h01 >> 16) + fl_from16<DT>(xr.x >> 16));
            h23 = fl_pack2<DT>(fl_from16<DT>(h23 & 0xffffu) + fl_from16<DT>(xr.y & 0xffffu), fl_from16<DT>(h23 >> 16) + fl_from16<DT>(xr.y >> 16));
          }
          u16* yrow = p.y + (((size_t)n * H + oy) * W + fr) * Cout;
          *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h01, h23);
        }
      }
    }
    if (R + 1 < ROUNDS) __syncthreads();
  }
  if (stamp) p.dbg[3] = __builtin_readcyclecounter();
}

// ---- host side -------------------------------------------------------------------------------------------------------------
template <int DT, int KS, int NW, int NCHW, int NFO, int ITEMS, bool DBG = false>
static void mbk_launch(const MbkParams& p, hipStream_t stream) {
  constexpr int lds = MbkLds<KS, NW, NCHW, NFO, ITEMS>::bytes;
  static_assert(lds <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbk_kernel<DT, KS, NW, NCHW, NFO, ITEMS, DBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((mbk_kernel<DT, KS, NW, NCHW, NFO, ITEMS, DBG>), dim3(p.items / ITEMS), dim3(64 * NW * ITEMS), lds, stream, p);
}

// the instantiations: (KS, NW, NCHW, NFO); Chid <= 16 * NW * NCHW
static bool mbk_instance(int ks, int nw, int nchw, int nfo) {
  return ks == 5 && nfo == 10 && ((nw == 4 && nchw == 15) || (nw == 6 && nchw == 10) || (nw == 3 && nchw == 20));
}

template <int DT>
static bool mbk_dispatch(const MbkParams& p, int ks, int nw, int nchw, int nfo, int items_per_wg, hipStream_t stream) {
  if (!mbk_instance(ks, nw, nchw, nfo)) return false;
  const bool two = items_per_wg == 2 && p.items % 2u == 0;
  if (nw == 4 && (p.wave_mask != ~0u || p.pair_mask != ~0u)) mbk_launch<DT, 5, 4, 15, 10, 1, true>(p, stream);  // (debug)
  else if (nw == 4) two ? mbk_launch<DT, 5, 4, 15, 10, 2>(p, stream) : mbk_launch<DT, 5, 4, 15, 10, 1>(p, stream);
  else if (nw == 6) mbk_launch<DT, 5, 6, 10, 10, 1>(p, stream);
  else two ? mbk_launch<DT, 5, 3, 20, 10, 2>(p, stream) : mbk_launch<DT, 5, 3, 20, 10, 1>(p, stream);
  return true;
}

// Bytes of the image of a block, or 0 when no instance of the kernel takes it.  nw: the number of hidden-channel slices
// (waves per item) the image is built for: 4 | 6 | 3.
static size_t mbk_image_bytes(int Cin, int Chid, int Cout, int nw, int* nchw_out, int* halves_out) {
  if (Cin % 32 || Chid % 16 || Cout % 160 || Cout < 160 || nw < 1) return 0;
  const int ks = Cin / 32, nch = Chid / 16, nchw = (nch + nw - 1) / nw, nfo = 10, halves = Cout / 160;
  if (!mbk_instance(ks, nw, nchw, nfo)) return 0;
  if (nchw_out) *nchw_out = nchw;
  if (halves_out) *halves_out = halves;
  if (nw == 4) return MbkGeo<5, 4, 15, 10>::bytes(halves);
  if (nw == 6) return MbkGeo<5, 6, 10, 10>::bytes(halves);
  return MbkGeo<5, 3, 20, 10>::bytes(halves);
}

// Returns 1 when the block is not one of this kernel's (the caller then runs ssdk_mbconv.hip's), 0 after a launch.
int launch_mbk(const ssdk_mbconv_desc* d, hipStream_t stream) {
  static const int env = getenv("SSDK_MBK") ? atoi(getenv("SSDK_MBK")) : 1;
  const int variant = d->variant;  // 0 auto, 3 this kernel wherever it exists (tests), other non-zero values: never
  if ((!env && variant != 3) || (variant != 0 && variant != 3)) return 1;
  if (d->stem || d->stride != 1 || d->W != 16 || !d->w_image || d->image_nw < 1) return 1;
  int nchw = 0, halves = 0;
  const size_t need = mbk_image_bytes(d->Cin, d->Chid, d->Cout, d->image_nw, &nchw, &halves);
  if (need == 0 || (size_t)d->w_image_bytes < need || ((uintptr_t)d->w_image & 15)) return 1;
  const int pairs = (d->H + 1) / 2;
  const long items = (long)halves * d->N * pairs;
  static const int env_min = getenv("SSDK_MBK_MIN") ? atoi(getenv("SSDK_MBK_MIN")) : 256;
  if (items < env_min && variant != 3) return 1;  // a handful of items cannot fill the chip: the tiled kernel's 8x8 tiles can
  MbkParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  p.img = (const unsigned char*)d->w_image;
  p.N = d->N;
  p.H = d->H;
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
  // items per workgroup (SSDK_MBK_ITEMS = 1 | 2; read per call: tests switch it): 2 = the waves of two items read the same
  // weight fragments in lock step (see the header); it needs an even number of items
  const char* ei = getenv("SSDK_MBK_ITEMS");
  const int items_per_wg = (ei && *ei) ? atoi(ei) : 1;  // (measured: 27.0 k vs 27.3 k cycles in the loop, 12 k vs 7 k in the exchange: one item per workgroup)
  static const int env_dbg = getenv("SSDK_MB_DBG") ? atoi(getenv("SSDK_MB_DBG")) : 0;
  static unsigned long long* dbg_dev = nullptr;
  if (env_dbg) {
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(unsigned long long));
    p.dbg = dbg_dev;
  }
  const int ks = d->Cin / 32;
  const bool ok = d->dtype == SSDK_BF16 ? mbk_dispatch<SSDK_BF16>(p, ks, d->image_nw, nchw, 10, items_per_wg, stream)
                                        : mbk_dispatch<SSDK_F16>(p, ks, d->image_nw, nchw, 10, items_per_wg, stream);
  if (ok && env_dbg) {  // debug only: synchronises
    unsigned long long h[4];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[mbk dbg] Cin=%d Chid=%d Cout=%d nw=%d items=%ld x%d : setup %llu loop %llu exchange %llu\n", d->Cin, d->Chid,
            d->Cout, d->image_nw, items, items_per_wg, h[1] - h[0], h[2] - h[1], h[3] - h[2]);
  }
  return ok ? 0 : 1;
}

}  // namespace ssdk

extern "C" size_t ssdk_mbk_image_bytes(int Cin, int Chid, int Cout, int nw) {
  return ssdk::mbk_image_bytes(Cin, Chid, Cout, nw, nullptr, nullptr);
}
