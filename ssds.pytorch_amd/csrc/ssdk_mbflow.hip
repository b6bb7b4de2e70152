// ssdk_mbflow.hip -- MobileNetV2 inverted-residual block (torchvision InvertedResidual behind nets/mobilenet.py:56,
// 84-89: 1x1 expand + BN + ReLU6 -> 3x3 depthwise (stride 1|2) + BN + ReLU6 -> 1x1 project + BN [+ x]) for the
// HIGH-RESOLUTION blocks (Cin <= 32, hidden <= 192, Cout <= 64), with the expanded tensor kept in REGISTERS.
//
// Why a second kernel: ssdk_mbconv.hip moves the 6x-expanded tile through LDS twice (expand MFMA -> sE -> depthwise ->
// sD -> project MFMA) between workgroup barriers; on the 128^2 / 256^2 maps those round trips ARE the run time (SQ
// counters: LDS pipe busy 55-67 %, waves parked at barriers 42-54 %).  Here a WAVE owns a 16-pixel-wide column strip
// of the map and walks down its rows; nothing is shared between waves, there is no barrier after the weights are
// staged, and the only LDS traffic is broadcast reads of weights:
//   * expand:  D[hc][px] = We[hc][k] * x[k][px] on the matrix cores.  The MFMA output layout puts one PIXEL in a lane
//     (lane = fg*16 + fr: pixel fr of the strip, channels 16c + 4fg .. +3 of chunk c), which is exactly what a
//     depthwise convolution wants: the horizontal taps are DPP row shifts (v_mov_b32 row_shr:1 / row_shl:1 inside the
//     16-lane rows), the vertical taps are the next rows the wave computes.
//   * depthwise: every expanded row is folded, as it appears, into the packed-fp16 accumulators of the (up to three)
//     output rows it contributes to; an output row is complete when its last input row has passed.
//   * project: the finished depthwise row already has the B-operand layout of the f16 MFMA (one pixel per lane, eight
//     hidden channels per lane and k-step) up to a permutation of k, which is applied to the projection weights once,
//     when they are staged:  k-step t, lane group fg, element j  <->  hidden channel (2t + j/4)*16 + 4fg + j%4.
// x is read straight from global memory as the expand GEMM's B operand (16 bytes = 8 input channels of one pixel per
// lane, one row ahead); the strips overlap by two pixels (halo), segments of rows by two rows.
//
// Numerics follow ssdk_mbconv.hip: E = fp16(clamp(bn(expand), 0, 6)) (round toward zero; the BN scale is folded into the
// expand weights -- by the host before they are rounded, else at staging time -- and the BN bias is the accumulator the
// MFMA starts from), depthwise in packed fp16 in the order bias, (ky, kx), D = clamp(acc, 0, 6), projection on the f16 MFMA
// with fp32 accumulation, output rounded to the model dtype before the residual is added.  (Round 4: both internal tensors
// are kept in units of six, ReLU6 = the [0, 1] clamp modifier of the instruction that produces them: ssdk_flow_common.h.)
// Stride 2 (round 4, "parity split"): which input pixel sits in which lane is the kernel's choice -- it is only the address
// the lane loads its B operand from.  The wave's two strips are the EVEN and the ODD input columns of one range: lane j of
// strip 0 holds column 2(ox0 + j), lane j of strip 1 column 2(ox0 + j) - 1, so output pixel ox0 + j finds its centre tap in
// its own lane of strip 0, its left tap in its own lane of strip 1 and its right tap one lane up in strip 1: ONE lane
// shift per packed word and 15 outputs per wave and row in one accumulator set.  (Contiguous columns -- round 2/3 -- left
// the outputs in every second lane: 7 per strip, and merging two strips into one accumulator set cost 8 DPP selects per
// word pair, a quarter of the row loop's VALU instructions.)
#include <stdio.h>
#include <atomic>
#include <type_traits>

#include "ssdk_common.h"
#include "ssdk_flow_common.h"

namespace ssdk {

constexpr int kFlowThreads = 256;

// LDS image of the weights (bytes), all 16-byte aligned; NCH = hidden chunks of 16, T = projection k-steps, NFO = Cout / 16
template <int NCH, int NFO>
struct FlowLds {
  static constexpr int T = (NCH + 1) / 2;
  static constexpr int we = 0;                             // [NCH][64 lanes] u32x4
  static constexpr int sb = we + NCH * 1024;               // [NCH][4 fg][se f32x4 | be f32x4]
  static constexpr int wd = sb + NCH * 4 * 32;             // [NCH][9 taps][4 fg] 8 bytes
  static constexpr int bd = wd + NCH * 9 * 4 * 8;          // [NCH][4 fg] 8 bytes
  static constexpr int wp = (bd + NCH * 4 * 8 + 15) & ~15; // [NFO][T][64 lanes] u32x4
  static constexpr int spb = wp + NFO * T * 1024;          // [NFO][4 fg][sp f32x4 | bp f32x4]
  static constexpr int bytes = spb + NFO * 4 * 32;
};

template <int DT, int S, int NCH, int NFO, int NS, bool STEM = false, bool YE = false, bool PAIR = false, bool XD = false>
__global__ __launch_bounds__(kFlowThreads, ((STEM && !YE) || PAIR) ? 3 : 2) void mbflow_kernel(const FlowParams p) {
  static_assert(!PAIR || (S == 1 && !STEM), "row pairs: stride-1 blocks");
  static_assert(!XD || (STEM && !YE), "dword image loads: the stem's interior instance");
  using L = FlowLds<NCH, NFO>;
  constexpr int T = L::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int Cin = p.Cin, Chid = p.Chid, Cout = p.Cout;

  // ---- stage the weights once per workgroup in the layouts the lanes read them in ----------------------------------
  for (u32 i = tid; i < (u32)(NCH * 64); i += kFlowThreads) {  // expand weights as A fragments: row hc = 16c + fr, k = 8fg ..
    const u32 c = i >> 6, l = i & 63u, hc = c * 16 + (l & 15u), k0 = (l >> 4) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if constexpr (STEM) {  // p.we: [Chid][ky 3][kx padded to 8][ci padded to 4]; this kernel's k = ci*9 + ky*3 + kx
      if (hc < (u32)Chid) {
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
          u32 ci, ky, kx;
          bool kv;
          if constexpr (XD) {  // the slot order of the dword image loads (load_x below): lane group g holds the triples 2g, 2g + 1 (+ 8)
            const u32 g = k0 >> 3;
            u32 t;
            if (j < 3u) {  // (kx 1, kx 2) = the pixel's own aligned dword, kx 0 = the high half of its left neighbour's
              t = 2u * g;
              kx = j == 2u ? 0u : j + 1u;
              kv = true;
            } else if (j < 6u) {
              t = 2u * g + 1u;
              kx = j == 3u ? 0u : j - 3u;
              kv = true;
            } else {  // triple 8: its dword in group 0, its kx = 0 tap in the high half of group 1's last dword
              t = 8u;
              kx = g == 0u ? j - 5u : 0u;
              kv = g == 0u || (g == 1u && j == 7u);
            }
            ci = t / 3u;
            ky = t % 3u;
            kv = kv && ci < (u32)p.Cimg;
          } else {
            const u32 k = k0 + j;
            ci = k / 9u;
            ky = (k % 9u) / 3u;
            kx = k % 3u;
            kv = k < 27u && ci < (u32)p.Cimg;
          }
          const u32 h = kv ? (u32)p.we[(size_t)hc * 96 + (ky * 8 + kx) * 4 + ci] : 0u;
          v[j >> 1] |= h << ((j & 1u) * 16u);
        }
      }
    } else {
      if (hc < (u32)Chid && k0 < (u32)Cin) v = *reinterpret_cast<const u32x4*>(p.we + (size_t)hc * Cin + k0);
    }
    // The expand BatchNorm rides on the matrix cores: its scale is folded into the staged weights here (a scale of
    // exactly 1 -- the host folds it before rounding the weights, fused_conv.MbPack -- leaves them bit for bit), its
    // bias is the accumulator the MFMA starts from.  The row loop then has no BN arithmetic at all.
    if (hc < (u32)Chid) {
      const float sc = p.se[hc];
      if (sc != 1.0f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32 w2 = v[q];
          v[q] = fl_to16<DT>(fl_from16<DT>(w2 & 0xffffu) * sc) | (fl_to16<DT>(fl_from16<DT>(w2 >> 16) * sc) << 16);
        }
      }
    }
    *reinterpret_cast<u32x4*>(smem + L::we + i * 16) = v;
  }
  for (u32 i = tid; i < (u32)(NCH * 4); i += kFlowThreads) {  // expand BN, depthwise bias: 4 channels 16c + 4g ..
    const u32 hc = (i >> 2) * 16 + (i & 3u) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    uint2 d = make_uint2(0u, 0u);
    if (hc < (u32)Chid) {
      s = *reinterpret_cast<const f32x4*>(p.se + hc);
      b = *reinterpret_cast<const f32x4*>(p.be + hc);
      d = *reinterpret_cast<const uint2*>(p.bd + hc);
      d = make_uint2(fl_sixth_h2(d.x), fl_sixth_h2(d.y));  // (units of six, ssdk_flow_common.h)
    }
    *reinterpret_cast<f32x4*>(smem + L::sb + i * 32) = s;
    *reinterpret_cast<f32x4*>(smem + L::sb + i * 32 + 16) = b;
    *reinterpret_cast<uint2*>(smem + L::bd + i * 8) = d;
  }
  for (u32 i = tid; i < (u32)(NCH * 9 * 4); i += kFlowThreads) {  // depthwise taps [c][tap][g]
    const u32 g = i & 3u, tap = (i >> 2) % 9u, c = (i >> 2) / 9u, hc = c * 16 + g * 4;
    uint2 d = make_uint2(0u, 0u);
    if (hc < (u32)Chid) d = *reinterpret_cast<const uint2*>(p.wd + (size_t)tap * Chid + hc);
    *reinterpret_cast<uint2*>(smem + L::wd + i * 8) = d;
  }
  for (u32 i = tid; i < (u32)(NFO * T * 64 * 2); i += kFlowThreads) {  // projection weights, k permuted (see the header): 8 bytes per item
    const u32 half = i & 1u, l = (i >> 1) & 63u, ft = i >> 7, t = ft % (u32)T, f = ft / (u32)T;
    const u32 co = f * 16 + (l & 15u), hc = (2 * t + half) * 16 + (l >> 4) * 4;
    uint2 d = make_uint2(0u, 0u);
    if (co < (u32)Cout && hc < (u32)Chid) d = *reinterpret_cast<const uint2*>(p.wp + (size_t)co * Chid + hc);
    *reinterpret_cast<uint2*>(smem + L::wp + (ft * 64 + l) * 16 + half * 8) = d;
  }
  for (u32 i = tid; i < (u32)(NFO * 4); i += kFlowThreads) {  // projection BN: 4 output channels 16f + 4g ..
    const u32 co = (i >> 2) * 16 + (i & 3u) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (co < (u32)Cout) {
      s = *reinterpret_cast<const f32x4*>(p.sp + co) * 6.0f;  // (the depthwise output arrives in units of six)
      b = *reinterpret_cast<const f32x4*>(p.bp + co);
    }
    *reinterpret_cast<f32x4*>(smem + L::spb + i * 32) = s;
    *reinterpret_cast<f32x4*>(smem + L::spb + i * 32 + 16) = b;
  }
  __syncthreads();

  // ---- this wave's work item: (image, row segment, group of NS neighbouring strips) ---------------------------------
  // NS strips per wave share every weight read (the LDS pipe was the co-bottleneck with one strip) and give the wave two
  // independent MFMA -> BN -> DPP -> FMA chains to interleave.
  // (readfirstlane: the compiler cannot see that tid >> 6 is wave-uniform; with it the item's coordinates, row pointers
  //  and row predicates live in SGPRs and the address arithmetic leaves the VALU)
  const u32 item = blockIdx.x * (kFlowThreads / 64) + (u32)__builtin_amdgcn_readfirstlane((int)wave);
  const int groups = (p.strips + NS - 1) / NS;
  const int nseg = STEM ? __builtin_popcountll(p.seg_mask) : p.segs;
  const u32 per_img = (u32)(groups * nseg);
  if (item >= (u32)p.N * per_img) return;
  const int n = (int)(item / per_img), rem = (int)(item % per_img);
  int seg = rem / groups;
  const int grp = rem % groups;
  if constexpr (STEM) {  // the seg-th set bit of the mask
    unsigned long long m = p.seg_mask;
    for (int i = 0; i < seg; ++i) m &= m - 1;
    seg = __builtin_ctzll(m);
  }
  constexpr int OW = S == 1 ? 14 : 7;             // output pixels per strip (stride 2 with two strips: 15 per PAIR, below)
  // stride 2 with two strips per wave: even / odd input columns, ONE accumulator set (see the header)
  constexpr bool MERGE = S == 2 && NS == 2;
  constexpr int NA = MERGE ? 1 : NS;              // accumulator / output sets per wave
  const int oy0 = seg * p.rs, oy1 = (oy0 + p.rs < p.Ho ? oy0 + p.rs : p.Ho) - 1;  // output rows [oy0, oy1]
  int ix[NS], oxl[NA];
  bool col_ok[NS], out_lane[NA];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int strip = grp * NS + s;
    const int ox0 = strip * OW;
    ix[s] = MERGE ? 2 * (grp * 15 + (int)fr) - s : ox0 * S - 1 + (int)fr;  // this lane's input column in strip s
    col_ok[s] = strip < p.strips && (unsigned)ix[s] < (unsigned)p.W;
    if constexpr (!MERGE) {
      // output pixel of this lane (if any): stride 1: lanes 1..14, stride 2: odd lanes 1..13
      oxl[s] = S == 1 ? ox0 + (int)fr - 1 : ox0 + ((int)fr - 1) / 2;
      out_lane[s] = strip < p.strips && (S == 1 ? (fr >= 1u && fr <= 14u) : ((fr & 1u) && fr <= 13u)) && oxl[s] < p.Wo;
    }
  }
  if constexpr (MERGE) {  // lane j <= 14: output grp * 15 + j (lane 15 has no right tap)
    oxl[0] = grp * 15 + (int)fr;
    out_lane[0] = fr <= 14u && oxl[0] < p.Wo;
  }

  const u16* ximg = STEM ? p.x + (size_t)n * p.Cimg * p.Himg * p.Wimg : p.x + (size_t)n * p.H * p.W * Cin;
  const bool k_ok = fg * 8u < (u32)Cin;
  // STEM: the B operand is an im2col row of the image, k = ci*9 + ky*3 + kx (lane group fg holds k = 8fg .. 8fg+7):
  // eight 2-byte loads per lane at offsets that do not depend on the row or the strip (uniform base + per-lane offset)
  // Per-lane byte offsets of the eight values, fixed for the whole kernel, with the column validity folded in: a value
  // that falls left / right of the image (or a k >= 27) carries an offset far outside the buffer, and a buffer load
  // out of range returns 0 without touching memory -- the steady-state loop needs no masks and no branches.
  constexpr u32 kOOR = 0x40000000u;
  u32 xoff[STEM ? NS : 1][STEM ? 8 : 1], xoff0[(STEM && !YE) ? NS : 1][(STEM && !YE) ? 8 : 1], kyM[STEM ? 3 : 1];
  const int pstr = STEM ? (p.layout == 1 ? 1 : p.Cimg) : 0, cstr = STEM ? (p.layout == 1 ? p.Himg * p.Wimg : 1) : 0;
  if constexpr (STEM) {
    kyM[0] = kyM[1] = kyM[2] = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32 k = fg * 8u + (u32)j, ci = k / 9u, ky = (k % 9u) / 3u, kx = k % 3u;
      const bool kv = k < 27u && ci < (u32)p.Cimg;
      const u32 vo = (u32)(2 * (int)fr * pstr + (int)ci * cstr + ((int)ky * p.Wimg + (int)kx) * pstr) * 2u;
      kyM[0] |= (kv && ky == 0u) ? (1u << j) : 0u;
      kyM[1] |= (kv && ky == 1u) ? (1u << j) : 0u;
      kyM[2] |= (kv && ky == 2u) ? (1u << j) : 0u;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        xoff[s][j] = (kv && col_ok[s] && (unsigned)(2 * ix[s] - 1 + (int)kx) < (unsigned)p.Wimg) ? vo : kOOR;
        if constexpr (!YE) xoff0[s][j] = ky == 0u ? kOOR : xoff[s][j];  // stem row 0: its ky = 0 taps lie above the image
      }
    }
  }  // (a k >= 27 carries an out-of-range offset like an out-of-image value: the load returns 0, no mask needed)
  // STEM: the buffer is this image, opened `xmargin` bytes EARLY so that the (row, strip) offset in the SGPR is never
  // negative (row 0 starts one image row and three pixels before the image); nothing in front of the image is ever
  // read: the values that would lie there carry out-of-range offsets.
  // XD (round 5): the image rows are read as ALIGNED DWORDS (NCHW image, even width).  Of a 3x3 / stride-2 patch the taps
  // kx = 1, 2 of a (channel, kernel row) are one aligned dword (columns 2 ix, 2 ix + 1) and kx = 0 (column 2 ix - 1) is the
  // high half of the LEFT NEIGHBOUR pixel's dword -- one lane down (DPP).  Nine (ci, ky) triples in 32 k-slots: lane group
  // g holds triples 2g and 2g + 1 as dwords [pair A | (A.kx0, B.kx0) | pair B | x], x = the pair of triple 8 in group 0 and
  // triple 8's kx = 0 (the high half of the left neighbour's dword, loaded directly) in group 1: THREE dword loads per lane,
  // row and strip (+ two for the left edge lane of the wave's first strip) instead of eight 2-byte loads -- the kernel's
  // time follows the number of load instructions (measured with timing-only builds: 137 / 115 / 104 us for 8 / 5 / 3).
  // Byte offsets: soff = (row 2 iy - 1, column of the strip's lane 0) + xmargin - 4, per lane (2 fr + ci plane + ky row) * 2
  // + 4 for its own pixel, + 0 for its left neighbour (so that nothing is ever negative); invalid values carry kOOR.
  u32 xd_off[XD ? NS : 1][3], xd_off0[XD ? NS : 1][2], xd_edge[2], xd_edge0[2];
  if constexpr (XD) {
    const u32 tA = 2u * fg, tB = 2u * fg + 1u;
    auto voff = [&](u32 t, int shift, int s_) -> u32 {  // triple t of pixel ix[s_] - shift
      const u32 ci = t / 3u, ky = t % 3u;
      const int px = ix[s_] - shift;
      const bool ok = ci < (u32)p.Cimg && grp * NS + s_ < p.strips && (unsigned)px < (unsigned)p.W;
      return ok ? (u32)((2 * (int)fr + (int)ci * p.Himg * p.Wimg + (int)ky * p.Wimg) * 2 + 4 * (1 - shift)) : kOOR;
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      xd_off[s][0] = voff(tA, 0, s);
      xd_off[s][1] = voff(tB, 0, s);
      xd_off[s][2] = fg == 0u ? voff(8u, 0, s) : (fg == 1u ? voff(8u, 1, s) : kOOR);
      xd_off0[s][0] = tA % 3u == 0u ? kOOR : xd_off[s][0];  // stem row 0: the ky = 0 taps lie above the image
      xd_off0[s][1] = tB % 3u == 0u ? kOOR : xd_off[s][1];
    }
    xd_edge[0] = fr == 0u ? voff(tA, 1, 0) : kOOR;  // the wave's first strip: its lane 0 has no lane to its left
    xd_edge[1] = fr == 0u ? voff(tB, 1, 0) : kOOR;
    xd_edge0[0] = tA % 3u == 0u ? kOOR : xd_edge[0];
    xd_edge0[1] = tB % 3u == 0u ? kOOR : xd_edge[1];
  }
  const int xmargin = STEM ? (p.Wimg + 8) * pstr * 2 : 0;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const unsigned char*>(ximg) - xmargin), 0, STEM ? p.Cimg * p.Himg * p.Wimg * 2 + xmargin : 0, 0x00020000);
  // The loaded words of one strip row, NOT yet packed (STEM: eight 16-bit values in eight registers): packing them
  // here would put the s_waitcnt right behind the loads, and the row is fetched one row AHEAD of its use.
  constexpr int XW = STEM ? 8 : 4;
  struct XRow {
    u32 w[XW];
  };
  auto load_x = [&](int iy, int s) -> XRow {  // B operand of the expand GEMM: 8 input channels of pixel (iy, ix[s])
    XRow out;
#pragma unroll
    for (int q = 0; q < XW; ++q) out.w[q] = 0u;
    if constexpr (STEM) {
      const int strip = grp * NS + s;
      // element offset of (image row 2iy-1, image column 2*ix-1 of the strip's lane 0, channel 0)
      const int base = ((2 * iy - 1) * p.Wimg + (2 * (strip * 14 - 1) - 1)) * pstr;
      if constexpr (XD) {
        const bool row_in = (unsigned)iy < (unsigned)p.H, top = iy == 0;  // wave-uniform
        const int soff = row_in ? (base + 1) * 2 + xmargin - 4 : (int)kOOR;  // (column 2 ix of the strip's lane 0; pstr == 1)
        out.w[0] = (u32)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(top ? xd_off0[s][0] : xd_off[s][0]), soff, 0);
        out.w[1] = (u32)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(top ? xd_off0[s][1] : xd_off[s][1]), soff, 0);
        out.w[2] = (u32)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)xd_off[s][2], soff, 0);
        if (s == 0) {
          out.w[3] = (u32)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(top ? xd_edge0[0] : xd_edge[0]), soff, 0);
          out.w[4] = (u32)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(top ? xd_edge0[1] : xd_edge[1]), soff, 0);
        }
        return out;
      } else if constexpr (YE) {  // items that touch the top / bottom of the image: rows outside it are masked per value
        u32 rowmask = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          rowmask |= ((unsigned)iy < (unsigned)p.H && (unsigned)(2 * iy - 1 + ky) < (unsigned)p.Himg) ? kyM[ky] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const u32 off = (((rowmask >> j) & 1u) && xoff[s][j] != kOOR) ? xoff[s][j] + (u32)(base * 2 + xmargin) : kOOR;
          out.w[j] = (u32)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, (int)off, 0, 0);
        }
      } else {
        // per-lane offset in a VGPR (fixed) + the row / strip offset in an SGPR, no branch: a row outside the block's grid
        // puts the whole row out of range through the SGPR, stem row 0 swaps in the offsets without the ky = 0 taps
        const bool row_in = (unsigned)iy < (unsigned)p.H, top = iy == 0;  // wave-uniform
        const int soff = row_in ? base * 2 + xmargin : (int)kOOR;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          out.w[j] = (u32)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, (int)(top ? xoff0[s][j] : xoff[s][j]), soff, 0);
      }
      return out;
    } else {
      if (k_ok && col_ok[s] && (unsigned)iy < (unsigned)p.H) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ximg + ((size_t)iy * p.W + ix[s]) * Cin + fg * 8);
        out.w[0] = v[0];
        out.w[1] = v[1];
        out.w[2] = v[2];
        out.w[3] = v[3];
      }
      return out;
    }
  };

  // (wave-uniform: every lane of the stamped wave writes the same word)
  const bool dbgw = p.dbg != nullptr && blockIdx.x == gridDim.x / 2 && __builtin_amdgcn_readfirstlane((int)wave) == 0;
  int dbg_k = 0;
  fl_h2 accA[NA][NCH * 2], accB[NA][NCH * 2], accC[NA][NCH * 2];
  fl_h2 accD[PAIR ? NA : 1][PAIR ? NCH * 2 : 1];  // row pairs keep four output rows open

  // One input row.  FIN / MID / INI: the row is the last (ky = 2) / middle (ky = 1) / first (ky = 0) row of the output
  // row accumulated in fin / mid / ini; the FIN row is completed, projected and stored as output row `oy_fin`.
  auto row = [&](const XRow (&xraw)[NS], int iy, auto FINc, auto MIDc, auto INIc, fl_h2 (&fin)[NA][NCH * 2],
                 fl_h2 (&mid)[NA][NCH * 2], fl_h2 (&ini)[NA][NCH * 2], int oy_fin) {
    constexpr bool FIN = decltype(FINc)::value, MID = decltype(MIDc)::value, INI = decltype(INIc)::value;
    if (dbgw && dbg_k < 250) {
      p.dbg[dbg_k++] = __builtin_readcyclecounter();
      if constexpr (STEM) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // (the next row's loads stay in flight)
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      p.dbg[dbg_k++] = __builtin_readcyclecounter();
    }
    u32x4 xf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (XD) {
        // the left neighbour pixel's dwords: one lane down; lane 0 takes the edge load (first strip) or lane 13 of the strip to
        // the left (strips are 14 pixels apart: its lane 14 is this strip's lane 0)
        int oa, ob;
        if (s == 0) {
          oa = (int)xraw[0].w[3];
          ob = (int)xraw[0].w[4];
        } else {
          oa = __builtin_amdgcn_update_dpp(0, (int)xraw[s - 1].w[0], 0x123, 0xf, 0xf, true);  // row_ror:3: lane 0 <- lane 13
          ob = __builtin_amdgcn_update_dpp(0, (int)xraw[s - 1].w[1], 0x123, 0xf, 0xf, true);
        }
        const u32 an = (u32)__builtin_amdgcn_update_dpp(oa, (int)xraw[s].w[0], 0x111, 0xf, 0xf, false);  // row_shr:1, lane 0 keeps `old`
        const u32 bn = (u32)__builtin_amdgcn_update_dpp(ob, (int)xraw[s].w[1], 0x111, 0xf, 0xf, false);
        xf[s][0] = xraw[s].w[0];
        xf[s][1] = __builtin_amdgcn_perm(bn, an, 0x07060302u);  // (A.kx0 = high half of an, B.kx0 = high half of bn)
        xf[s][2] = xraw[s].w[1];
        xf[s][3] = fg == 1u ? (xraw[s].w[2] & 0xffff0000u) : xraw[s].w[2];
      } else if constexpr (STEM) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xf[s][q] = xraw[s].w[2 * q] | (xraw[s].w[2 * q + 1] << 16);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) xf[s][q] = xraw[s].w[q];
      }
    }
    fl_f2 hi[NS];  // 1/6, or 0 for a pixel outside the image (the zero padding of the EXPANDED tensor)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float k = (col_ok[s] && (unsigned)iy < (unsigned)p.H) ? kFlSixth : 0.f;
      hi[s] = fl_f2{k, k};
    }
    const bool store_row = FIN && oy_fin >= oy0 && oy_fin <= oy1;               // wave-uniform
    uint2 resv[NA][NFO];
    if (!STEM && !MERGE && FIN && store_row && p.residual) {  // issued early; consumed in the epilogue (stride 1 only)
#pragma unroll
      for (int s = 0; s < NA; ++s)
#pragma unroll
        for (int f = 0; f < NFO; ++f) {
          const int co = f * 16 + (int)fg * 4;
          uint2 r = make_uint2(0u, 0u);
          if (out_lane[s] && co < Cout) r = *reinterpret_cast<const uint2*>(ximg + ((size_t)oy_fin * p.W + oxl[s]) * Cin + co);
          resv[s][f] = r;
        }
    }
    // Software pipeline over the chunks: while the VALU works on chunk c (BN, clamp, DPP shifts, taps) the matrix pipe
    // already runs the expand MFMAs of chunk c+1, and the A fragment of chunk c+2 is on its way from LDS.
    asm volatile("" ::: "memory");  // (the weights are loop invariant: they are to be RE-READ from LDS, broadcast reads,
                                    //  not kept in hundreds of registers across the rows)
    f32x4 e_cur[NS], e_nxt[NS];
    u32x4 wa = *reinterpret_cast<const u32x4*>(smem + L::we + (int)lane * 16);
    {  // the accumulator starts from the BN bias of the lane's four channels (the scale sits in the weights)
      const f32x4 bv0 = *reinterpret_cast<const f32x4*>(smem + L::sb + (int)fg * 32 + 16);
#pragma unroll
      for (int s = 0; s < NS; ++s) e_cur[s] = fl_mfma<DT>(wa, xf[s], bv0);  // D[hc = 16c + 4fg + r][px = fr]
    }
    if (NCH > 1) wa = *reinterpret_cast<const u32x4*>(smem + L::we + (64 + (int)lane) * 16);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      asm volatile("" ::: "memory");
      uint2 wt[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint2*>(smem + L::wd + ((c * 9 + t) * 4 + (int)fg) * 8);
      // the depthwise bias is the value a fresh accumulator starts from (ky = 0 row)
      uint2 bdi = make_uint2(0u, 0u);
      if constexpr (INI) bdi = *reinterpret_cast<const uint2*>(smem + L::bd + (c * 4 + (int)fg) * 8);
      if (c + 1 < NCH) {
        const f32x4 bvn = *reinterpret_cast<const f32x4*>(smem + L::sb + ((c + 1) * 4 + (int)fg) * 32 + 16);
#pragma unroll
        for (int s = 0; s < NS; ++s) e_nxt[s] = fl_mfma<DT>(wa, xf[s], bvn);
        if (c + 2 < NCH) wa = *reinterpret_cast<const u32x4*>(smem + L::we + ((c + 2) * 64 + (int)lane) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      u32 ew[NS][2];  // E = fp16(clamp(bn(expand), 0, 6 | 0)) of the lane's pixel: channels (0, 1) and (2, 3) of its four
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const f32x4 e = e_cur[s];
        ew[s][0] = fl_unit_pack(e[0], e[1], hi[s]);
        ew[s][1] = fl_unit_pack(e[2], e[3], hi[s]);
      }
      auto fold = [&](fl_h2 l0, fl_h2 l1, fl_h2 c0, fl_h2 c1, fl_h2 r0, fl_h2 r1, int a) {
        auto taps = [&](int ky, fl_h2& a0, fl_h2& a1, bool init) {
          const uint2 w0 = wt[ky * 3], w1 = wt[ky * 3 + 1], w2 = wt[ky * 3 + 2];
          fl_h2 s0 = init ? fl_as_h2(bdi.x) : a0, s1 = init ? fl_as_h2(bdi.y) : a1;
          s0 = __builtin_elementwise_fma(l0, fl_as_h2(w0.x), s0);
          s1 = __builtin_elementwise_fma(l1, fl_as_h2(w0.y), s1);
          s0 = __builtin_elementwise_fma(c0, fl_as_h2(w1.x), s0);
          s1 = __builtin_elementwise_fma(c1, fl_as_h2(w1.y), s1);
          if (ky == 2) {  // the output row is complete with this tap: ReLU6 = the clamp of the FMA (units of six)
            s0 = fl_fma_clamp01(r0, fl_as_h2(w2.x), s0);
            s1 = fl_fma_clamp01(r1, fl_as_h2(w2.y), s1);
          } else {
            s0 = __builtin_elementwise_fma(r0, fl_as_h2(w2.x), s0);
            s1 = __builtin_elementwise_fma(r1, fl_as_h2(w2.y), s1);
          }
          // pin the results here: the updates of the rows that finish LATER are only used by the next row's code, and the
          // compiler otherwise sinks them below this row's (conditional) projection -- with every chunk's E values and
          // weights kept alive until then (400 live registers)
          asm volatile("" : "+v"(s0), "+v"(s1));
          a0 = s0;
          a1 = s1;
        };
        if constexpr (INI) taps(0, ini[a][2 * c], ini[a][2 * c + 1], true);
        if constexpr (MID) taps(1, mid[a][2 * c], mid[a][2 * c + 1], false);
        if constexpr (FIN) taps(2, fin[a][2 * c], fin[a][2 * c + 1], false);
      };
      if constexpr (MERGE) {  // left tap: own lane of the odd columns, centre: own lane of the even ones, right: odd, one lane up
        fold(fl_as_h2(ew[1][0]), fl_as_h2(ew[1][1]), fl_as_h2(ew[0][0]), fl_as_h2(ew[0][1]), fl_as_h2(fl_from_right(ew[1][0])),
             fl_as_h2(fl_from_right(ew[1][1])), 0);
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fold(fl_as_h2(fl_from_left(ew[s][0])), fl_as_h2(fl_from_left(ew[s][1])), fl_as_h2(ew[s][0]), fl_as_h2(ew[s][1]),
               fl_as_h2(fl_from_right(ew[s][0])), fl_as_h2(fl_from_right(ew[s][1])), s);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) e_cur[s] = e_nxt[s];
      // chunks are independent: left alone, the scheduler hoists every chunk's weight reads to the top of the row and
      // runs out of registers
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dbgw && dbg_k < 250) p.dbg[dbg_k++] = __builtin_readcyclecounter();
    if constexpr (FIN) {
      if (store_row) {
        // ---- the output row is complete: bias + ReLU6 in place, then it IS the projection's B operand --------------
        asm volatile("" ::: "memory");
        f32x4 yacc[NA][NFO];
#pragma unroll
        for (int s = 0; s < NA; ++s)
#pragma unroll
          for (int f = 0; f < NFO; ++f) yacc[s][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t) {
          u32x4 db[NA];
#pragma unroll
          for (int s = 0; s < NA; ++s) {
            db[s][0] = fl_as_u32(fin[s][4 * t]);
            db[s][1] = fl_as_u32(fin[s][4 * t + 1]);
            if (2 * t + 1 < NCH) {
              db[s][2] = fl_as_u32(fin[s][(4 * t + 2 < NCH * 2) ? 4 * t + 2 : 0]);
              db[s][3] = fl_as_u32(fin[s][(4 * t + 3 < NCH * 2) ? 4 * t + 3 : 0]);
            } else {
              db[s][2] = 0u;
              db[s][3] = 0u;
            }
          }
#pragma unroll
          for (int f = 0; f < NFO; ++f) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(smem + L::wp + ((f * T + t) * 64 + (int)lane) * 16);
#pragma unroll
            for (int s = 0; s < NA; ++s) yacc[s][f] = fl_mfma<SSDK_F16>(wf, db[s], yacc[s][f]);  // D[co = 16f + 4fg + r][px = fr]
          }
        }
#pragma unroll
        for (int s = 0; s < NA; ++s) {
          if (out_lane[s]) {
            u16* yrow = p.y + (((size_t)n * p.Ho + oy_fin) * p.Wo + oxl[s]) * Cout;
#pragma unroll
            for (int f = 0; f < NFO; ++f) {
              const int co = f * 16 + (int)fg * 4;
              if (co < Cout) {
                const f32x4 spv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32);
                const f32x4 bpv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32 + 16);
                // (hardware packed conversions: round to nearest even like the integer sequence of ssdk_mbconv.hip)
                u32 h01 = fl_pack2<DT>(fmaf(yacc[s][f][0], spv[0], bpv[0]), fmaf(yacc[s][f][1], spv[1], bpv[1]));
                u32 h23 = fl_pack2<DT>(fmaf(yacc[s][f][2], spv[2], bpv[2]), fmaf(yacc[s][f][3], spv[3], bpv[3]));
                if (!STEM && !MERGE && p.residual) {  // the block's output is rounded to the model dtype first, then x is added (torch's tensor add)
                  const u32 x01 = resv[s][f].x, x23 = resv[s][f].y;
                  h01 = fl_pack2<DT>(fl_from16<DT>(h01 & 0xffffu) + fl_from16<DT>(x01 & 0xffffu), fl_from16<DT>(h01 >> 16) + fl_from16<DT>(x01 >> 16));
                  h23 = fl_pack2<DT>(fl_from16<DT>(h23 & 0xffffu) + fl_from16<DT>(x23 & 0xffffu), fl_from16<DT>(h23 >> 16) + fl_from16<DT>(x23 >> 16));
                }
                *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h01, h23);
              }
            }
          }
        }
      }
    }
    if (dbgw && dbg_k < 250) p.dbg[dbg_k++] = __builtin_readcyclecounter();
  };


  // TWO input rows per visit of the chunks (PAIR; round 5): every weight read of the row loop -- expand fragment, bias, nine
  // taps, projection fragments -- then serves two rows.  (SQ counters of the 24 -> 144 -> 24 block at 128 x 128: LDS pipe 72 %
  // busy, 31 % of the wave cycles waiting for it -- twelve broadcast reads per chunk and row of one strip.)  Input rows iy,
  // iy + 1: row iy is the last row of output iy - 1 (a0), the middle of iy (a1), the first of iy + 1 (a2); row iy + 1 the
  // last of iy (a1), the middle of iy + 1 (a2), the first of iy + 2 (a3).  Stride 1, every row has all three roles.
  auto row2 = [&](const XRow (&x0)[NS], const XRow (&x1)[NS], int iy, fl_h2 (&a0)[NA][NCH * 2], fl_h2 (&a1)[NA][NCH * 2],
                  fl_h2 (&a2)[NA][NCH * 2], fl_h2 (&a3)[NA][NCH * 2]) {
    if (dbgw && dbg_k < 250) {
      p.dbg[dbg_k++] = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (the next pair's loads stay in flight)
      p.dbg[dbg_k++] = __builtin_readcyclecounter();
    }
    u32x4 xf[2][NS];
    fl_f2 hi[2][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xf[0][s][q] = x0[s].w[q];
        xf[1][s][q] = x1[s].w[q];
      }
      const float k0 = (col_ok[s] && (unsigned)iy < (unsigned)p.H) ? kFlSixth : 0.f;
      const float k1 = (col_ok[s] && (unsigned)(iy + 1) < (unsigned)p.H) ? kFlSixth : 0.f;
      hi[0][s] = fl_f2{k0, k0};
      hi[1][s] = fl_f2{k1, k1};
    }
    const bool st[2] = {iy - 1 >= oy0 && iy - 1 <= oy1, iy >= oy0 && iy <= oy1};  // wave-uniform: output rows iy - 1, iy
    uint2 resv[2][NA][NFO];
    if (p.residual) {
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < NA; ++s)
#pragma unroll
          for (int f = 0; f < NFO; ++f) {
            const int co = f * 16 + (int)fg * 4;
            uint2 r = make_uint2(0u, 0u);
            if (st[o] && out_lane[s] && co < Cout) r = *reinterpret_cast<const uint2*>(ximg + ((size_t)(iy - 1 + o) * p.W + oxl[s]) * Cin + co);
            resv[o][s][f] = r;
          }
    }
    asm volatile("" ::: "memory");
    f32x4 e_cur[2][NS], e_nxt[2][NS];
    u32x4 wa = *reinterpret_cast<const u32x4*>(smem + L::we + (int)lane * 16);
    {
      const f32x4 bv0 = *reinterpret_cast<const f32x4*>(smem + L::sb + (int)fg * 32 + 16);
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < NS; ++s) e_cur[o][s] = fl_mfma<DT>(wa, xf[o][s], bv0);
    }
    if (NCH > 1) wa = *reinterpret_cast<const u32x4*>(smem + L::we + (64 + (int)lane) * 16);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      asm volatile("" ::: "memory");
      uint2 wt[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint2*>(smem + L::wd + ((c * 9 + t) * 4 + (int)fg) * 8);
      const uint2 bdi = *reinterpret_cast<const uint2*>(smem + L::bd + (c * 4 + (int)fg) * 8);
      if (c + 1 < NCH) {
        const f32x4 bvn = *reinterpret_cast<const f32x4*>(smem + L::sb + ((c + 1) * 4 + (int)fg) * 32 + 16);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int s = 0; s < NS; ++s) e_nxt[o][s] = fl_mfma<DT>(wa, xf[o][s], bvn);
        if (c + 2 < NCH) wa = *reinterpret_cast<const u32x4*>(smem + L::we + ((c + 2) * 64 + (int)lane) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      auto taps = [&](int ky, fl_h2 l0, fl_h2 l1, fl_h2 c0, fl_h2 c1, fl_h2 r0, fl_h2 r1, fl_h2& q0, fl_h2& q1) {
        const uint2 w0 = wt[ky * 3], w1 = wt[ky * 3 + 1], w2 = wt[ky * 3 + 2];
        fl_h2 s0 = ky == 0 ? fl_as_h2(bdi.x) : q0, s1 = ky == 0 ? fl_as_h2(bdi.y) : q1;
        s0 = __builtin_elementwise_fma(l0, fl_as_h2(w0.x), s0);
        s1 = __builtin_elementwise_fma(l1, fl_as_h2(w0.y), s1);
        s0 = __builtin_elementwise_fma(c0, fl_as_h2(w1.x), s0);
        s1 = __builtin_elementwise_fma(c1, fl_as_h2(w1.y), s1);
        if (ky == 2) {
          s0 = fl_fma_clamp01(r0, fl_as_h2(w2.x), s0);
          s1 = fl_fma_clamp01(r1, fl_as_h2(w2.y), s1);
        } else {
          s0 = __builtin_elementwise_fma(r0, fl_as_h2(w2.x), s0);
          s1 = __builtin_elementwise_fma(r1, fl_as_h2(w2.y), s1);
        }
        asm volatile("" : "+v"(s0), "+v"(s1));
        q0 = s0;
        q1 = s1;
      };
#pragma unroll
      for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f32x4 e = e_cur[o][s];
          const u32 w0 = fl_unit_pack(e[0], e[1], hi[o][s]), w1 = fl_unit_pack(e[2], e[3], hi[o][s]);
          const fl_h2 l0 = fl_as_h2(fl_from_left(w0)), l1 = fl_as_h2(fl_from_left(w1)), c0 = fl_as_h2(w0), c1 = fl_as_h2(w1);
          const fl_h2 r0 = fl_as_h2(fl_from_right(w0)), r1 = fl_as_h2(fl_from_right(w1));
          if (o == 0) {
            taps(0, l0, l1, c0, c1, r0, r1, a2[s][2 * c], a2[s][2 * c + 1]);
            taps(1, l0, l1, c0, c1, r0, r1, a1[s][2 * c], a1[s][2 * c + 1]);
            taps(2, l0, l1, c0, c1, r0, r1, a0[s][2 * c], a0[s][2 * c + 1]);
          } else {
            taps(0, l0, l1, c0, c1, r0, r1, a3[s][2 * c], a3[s][2 * c + 1]);
            taps(1, l0, l1, c0, c1, r0, r1, a2[s][2 * c], a2[s][2 * c + 1]);
            taps(2, l0, l1, c0, c1, r0, r1, a1[s][2 * c], a1[s][2 * c + 1]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < NS; ++s) e_cur[o][s] = e_nxt[o][s];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dbgw && dbg_k < 250) p.dbg[dbg_k++] = __builtin_readcyclecounter();
    if (st[0] || st[1]) {  // (wave-uniform) both finished rows are projected with one read of every weight fragment
      asm volatile("" ::: "memory");
      f32x4 yacc[2][NA][NFO];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < NA; ++s)
#pragma unroll
          for (int f = 0; f < NFO; ++f) yacc[o][s][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; ++t) {
        u32x4 db[2][NA];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int s = 0; s < NA; ++s) {
            const fl_h2* fin = o == 0 ? a0[s] : a1[s];
            db[o][s][0] = fl_as_u32(fin[4 * t]);
            db[o][s][1] = fl_as_u32(fin[4 * t + 1]);
            if (2 * t + 1 < NCH) {
              db[o][s][2] = fl_as_u32(fin[(4 * t + 2 < NCH * 2) ? 4 * t + 2 : 0]);
              db[o][s][3] = fl_as_u32(fin[(4 * t + 3 < NCH * 2) ? 4 * t + 3 : 0]);
            } else {
              db[o][s][2] = 0u;
              db[o][s][3] = 0u;
            }
          }
#pragma unroll
        for (int f = 0; f < NFO; ++f) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(smem + L::wp + ((f * T + t) * 64 + (int)lane) * 16);
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int s = 0; s < NA; ++s) yacc[o][s][f] = fl_mfma<SSDK_F16>(wf, db[o][s], yacc[o][s][f]);
        }
      }
#pragma unroll
      for (int f = 0; f < NFO; ++f) {
        const int co = f * 16 + (int)fg * 4;
        if (co < Cout) {
          const f32x4 spv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32);
          const f32x4 bpv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32 + 16);
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int s = 0; s < NA; ++s) {
              if (!(st[o] && out_lane[s])) continue;
              u16* yrow = p.y + (((size_t)n * p.Ho + (iy - 1 + o)) * p.Wo + oxl[s]) * Cout;
              u32 h01 = fl_pack2<DT>(fmaf(yacc[o][s][f][0], spv[0], bpv[0]), fmaf(yacc[o][s][f][1], spv[1], bpv[1]));
              u32 h23 = fl_pack2<DT>(fmaf(yacc[o][s][f][2], spv[2], bpv[2]), fmaf(yacc[o][s][f][3], spv[3], bpv[3]));
              if (p.residual) {
                const u32 x01 = resv[o][s][f].x, x23 = resv[o][s][f].y;
                h01 = fl_pack2<DT>(fl_from16<DT>(h01 & 0xffffu) + fl_from16<DT>(x01 & 0xffffu), fl_from16<DT>(h01 >> 16) + fl_from16<DT>(x01 >> 16));
                h23 = fl_pack2<DT>(fl_from16<DT>(h23 & 0xffffu) + fl_from16<DT>(x23 & 0xffffu), fl_from16<DT>(h23 >> 16) + fl_from16<DT>(x23 >> 16));
              }
              *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h01, h23);
            }
        }
      }
    }
    if (dbgw && dbg_k < 250) p.dbg[dbg_k++] = __builtin_readcyclecounter();
  };

  const auto Y = std::true_type{};
  const auto No = std::false_type{};
  auto load_row = [&](XRow (&dst)[NS], int iy) {
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[s] = load_x(iy, s);
  };
  // The row loop is a plain counted loop over groups of 3 (stride 1) / 4 (stride 2) rows, so that the three
  // accumulator sets rotate through fixed registers; the up to 2 / 3 surplus rows at the end of a segment compute
  // into accumulators nobody stores (their output rows lie past oy1).  x is fetched one row ahead.
  XRow xa[NS], xb[NS];
  {
    if constexpr (PAIR) {
      // pairs of input rows (r, r + 1), fetched one pair ahead; the four accumulator sets rotate by two per pair
      const int rend = oy1 + 1;
      XRow xc[NS], xd[NS];
      load_row(xa, oy0 - 1);
      load_row(xb, oy0);
      for (int r = oy0 - 1; r <= rend; r += 4) {
        load_row(xc, r + 2);
        load_row(xd, r + 3);
        row2(xa, xb, r, accA, accB, accC, accD);
        load_row(xa, r + 4);
        load_row(xb, r + 5);
        row2(xc, xd, r + 2, accC, accD, accA, accB);
      }
    } else if constexpr (S == 1) {
      // input rows oy0-1 .. oy1+1; input row r is the first row of output r+1, the middle of r, the last of r-1
      const int rend = oy1 + 1;
      load_row(xa, oy0 - 1);
      for (int r = oy0 - 1; r <= rend; r += 6) {
        load_row(xb, r + 1);
        row(xa, r, Y, Y, Y, accA, accB, accC, r - 1);
        load_row(xa, r + 2);
        row(xb, r + 1, Y, Y, Y, accB, accC, accA, r);
        load_row(xb, r + 3);
        row(xa, r + 2, Y, Y, Y, accC, accA, accB, r + 1);
        load_row(xa, r + 4);
        row(xb, r + 3, Y, Y, Y, accA, accB, accC, r + 2);
        load_row(xb, r + 5);
        row(xa, r + 4, Y, Y, Y, accB, accC, accA, r + 3);
        load_row(xa, r + 6);
        row(xb, r + 5, Y, Y, Y, accC, accA, accB, r + 4);
      }
    } else {
      // input rows 2*oy0-1 .. 2*oy1+1; odd row 2m+1: last row of output m, first of m+1; even row 2m: middle of m
      const int rend = 2 * oy1 + 1;
      load_row(xa, 2 * oy0 - 1);
      for (int r = 2 * oy0 - 1; r <= rend; r += 4) {
        load_row(xb, r + 1);
        row(xa, r, Y, No, Y, accA, accC, accB, (r - 1) / 2);          // odd: finishes A, starts B
        load_row(xa, r + 2);
        row(xb, r + 1, No, Y, No, accC, accB, accC, 0);               // even: middle of B
        load_row(xb, r + 3);
        row(xa, r + 2, Y, No, Y, accB, accC, accA, (r + 1) / 2);      // odd: finishes B, starts A
        load_row(xa, r + 4);
        row(xb, r + 3, No, Y, No, accC, accA, accC, 0);               // even: middle of A
      }
    }
  }
}

constexpr int kFlowNS = 2;  // strips per wave

template <int DT, int S, int NCH, int NFO>
static void flow_launch(const FlowParams& p, unsigned grid, hipStream_t stream) {
  constexpr int lds = FlowLds<NCH, NFO>::bytes;
  if constexpr (S == 1 && NCH == 9) {
    if (p.layout == 101) {  // one strip per wave (launch_mbflow sets the marker: half the accumulators, three waves per SIMD)
      static const int pair = getenv("SSDK_FLOW_PAIR") ? atoi(getenv("SSDK_FLOW_PAIR")) : 1;
      if (pair) {  // two input rows per visit of the chunks: every weight read serves two rows
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DT, S, NCH, NFO, 1, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((mbflow_kernel<DT, S, NCH, NFO, 1, false, false, true>), dim3(grid), dim3(kFlowThreads), lds, stream, p);
        return;
      }
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DT, S, NCH, NFO, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((mbflow_kernel<DT, S, NCH, NFO, 1>), dim3(grid), dim3(kFlowThreads), lds, stream, p);
      return;
    }
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DT, S, NCH, NFO, kFlowNS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((mbflow_kernel<DT, S, NCH, NFO, kFlowNS>), dim3(grid), dim3(kFlowThreads), lds, stream, p);
}

template <int DT, int S>
static bool flow_dispatch(const FlowParams& p, int nch, int nfo, unsigned grid, hipStream_t stream) {
#define SSDK_FLOW(NCH_, NFO_)                                   \
  if (nch == NCH_ && nfo == NFO_) {                             \
    flow_launch<DT, S, NCH_, NFO_>(p, grid, stream);            \
    return true;                                                \
  }
  SSDK_FLOW(6, 2)
  SSDK_FLOW(9, 2)
  SSDK_FLOW(12, 2)
  SSDK_FLOW(12, 4)
#undef SSDK_FLOW
  return false;
}

// Returns 1 when the block is not one of this kernel's (the caller then runs ssdk_mbconv.hip's), 0 after a launch.
int launch_mbflow(const ssdk_mbconv_desc* d, hipStream_t stream) {
  static const int env = getenv("SSDK_MB_FLOW") ? atoi(getenv("SSDK_MB_FLOW")) : 1;
  const int variant = d->variant;  // 0 auto, 1 register-flow wherever it exists, -1 never (ssdk_mbconv_desc)
  if ((!env && variant <= 0) || variant < 0 || variant == 2 || variant == 3) return 1;  // (2 / 3 = ssdk_mbsplit.hip / ssdk_mbk.hip)
  const bool stem = d->stem != 0;
  if (stem) {  // network stem + expand-free first block: 3x3/s2 conv (<= 3 channels -> 32) as the "expand" GEMM, dw stride 1, 32 -> 16
    if (d->Cin > 3 || d->Chid != 32 || d->Cout != 16 || d->stride != 1 || d->residual) return 1;
  } else if (d->Cin > 32 || (d->Cin % 8) || d->Chid > 192 || (d->Chid % 16) || d->Cout > 64 || (d->Cout % 8)) {
    return 1;
  }
  const int nch = d->Chid / 16, nfo = stem ? 1 : (d->Cout <= 32 ? 2 : 4);
  if (!stem && !((nch == 6 || nch == 9 || nch == 12) && (nfo == 2 || nch == 12))) return 1;
  FlowParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  p.we = (const u16*)d->w_expand;
  p.se = d->scale_expand;
  p.be = d->bias_expand;
  p.wd = (const u16*)d->w_dw;
  p.bd = (const u16*)d->bias_dw;
  p.wp = (const u16*)d->w_project;
  p.sp = d->scale_project;
  p.bp = d->bias_project;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.Himg = d->H;
  p.Wimg = d->W;
  p.Cimg = d->Cin;
  p.layout = d->stem;
  if (stem) {  // the block's input grid is the stem convolution's output
    p.H = (d->H + 2 - 3) / 2 + 1;
    p.W = (d->W + 2 - 3) / 2 + 1;
    p.Cin = 32;
  }
  p.Chid = d->Chid;
  p.Cout = d->Cout;
  p.Ho = (p.H + 2 - 3) / d->stride + 1;
  p.Wo = (p.W + 2 - 3) / d->stride + 1;
  p.residual = d->residual;
  p.seg_mask = 0;
  p.dbg = nullptr;
  static const int env_dbg = getenv("SSDK_MB_DBG") ? atoi(getenv("SSDK_MB_DBG")) : 0;
  static unsigned long long* dbg_dev = nullptr;
  if (env_dbg) {
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 256 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_dev, 0, 256 * sizeof(unsigned long long), stream);
    p.dbg = dbg_dev;
  }
  auto dbg_print = [&]() {  // debug only: synchronises
    if (!env_dbg) return;
    unsigned long long hh[256];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(hh, dbg_dev, sizeof(hh), hipMemcpyDeviceToHost);
    fprintf(stderr, "[mbflow dbg] Cin=%d Chid=%d Cout=%d s=%d stem=%d rs=%d: per row (wait x | chunks | project+store | to next row):", d->Cin,
            d->Chid, d->Cout, d->stride, (int)stem, p.rs);
    for (int i = 0; i + 4 < 250 && hh[i + 4]; i += 4)
      fprintf(stderr, " [%llu %llu %llu %llu]", hh[i + 1] - hh[i], hh[i + 2] - hh[i + 1], hh[i + 3] - hh[i + 2], hh[i + 4] - hh[i + 3]);
    fprintf(stderr, "\n");
  };
  const int ow = d->stride == 1 ? 14 : 7;
  p.strips = (p.Wo + ow - 1) / ow;
  if (d->stride == 2) p.strips = 2 * ((p.Wo + 14) / 15);  // parity split: a pair of strips = 15 outputs (even / odd input columns)
  // rows per segment: long segments amortise the two halo rows, short ones give the chip enough waves (>= ~3 per SIMD)
  constexpr int env_rs = 0;  // (round 6: the SSDK_MB_FLOW_RS switch is gone, its A/B is settled)
  // strips per wave: two share every weight read, but the 144-channel stride-1 block then holds 216 registers (3 x 2 x 18
  // accumulators): two waves per SIMD.  With ONE strip it holds 132 -- three waves per SIMD -- and runs 125 -> 114 us (round 4,
  // same box; forcing 128 registers for a fourth wave spills and gives the gain back: 121 us).  The stem block (NCH = 2) does not
  // gain from one strip (137 -> 140 us).  SSDK_MB_FLOW_NS1: bit 0 the 144-channel block (default on), bit 1 the stem block.
  constexpr int env_ns1 = 1;  // (round 6: the SSDK_MB_FLOW_NS1 switch is gone, its A/B is settled)
  const bool ns1 = ((env_ns1 & 1) && !stem && d->stride == 1 && nch == 9) || ((env_ns1 & 2) && stem);
  const int flow_ns = ns1 ? 1 : kFlowNS;
  if (ns1 && !stem) p.layout = 101;
  const int groups = (p.strips + flow_ns - 1) / flow_ns;
  // (measured: ~5000 items of 16-32 rows beat ~2500 of 32-64 rows; below 16 rows the two halo rows of a segment cost
  //  too much, and a map that then yields fewer than 2048 items -- 64x64 at batch 64 -- stays with the LDS-tiled kernel)
  int rs = 64;
  while (rs > 16 && (long)d->N * groups * ((p.Ho + rs - 1) / rs) < 4096) rs >>= 1;
  if ((long)d->N * groups * ((p.Ho + rs - 1) / rs) < 2048 && env_rs <= 0 && variant <= 0) return 1;
  if (variant > 0) {  // forced (tests): short segments so that small maps still exercise several segments
    while (rs > 8 && (long)d->N * groups * ((p.Ho + rs - 1) / rs) < 2048) rs >>= 1;
  }
  if (env_rs > 0) rs = env_rs;
  if (rs > p.Ho) rs = p.Ho;
  p.rs = rs;
  p.segs = (p.Ho + rs - 1) / rs;
  const long items = (long)d->N * groups * p.segs;
  const unsigned grid = (unsigned)((items + 3) / 4);
  bool ok;
  if (stem) {
    // The interior instance copes with everything above the image and with rows outside the block's grid; only a real
    // stem row whose LAST tap row falls below the image (odd image heights) needs the masking instance.
    if (p.segs > 64) return 1;
    unsigned long long edge = 0;
    for (int sgi = 0; sgi < p.segs; ++sgi) {
      const int o0 = sgi * p.rs, o1 = (o0 + p.rs < p.Ho ? o0 + p.rs : p.Ho) - 1;
      const int last_row = o1 + 1 < p.H - 1 ? o1 + 1 : p.H - 1;  // last real stem row the segment reads
      if (2 * last_row + 1 >= p.Himg) edge |= 1ull << sgi;          // its ky = 2 tap lies below the image (odd heights)
    }
    const unsigned long long all = p.segs == 64 ? ~0ull : ((1ull << p.segs) - 1ull);
    constexpr int lds = FlowLds<2, 1>::bytes;
    auto go = [&](auto DTc, auto YEc, unsigned long long mask) {
      constexpr int DTv = decltype(DTc)::value;
      constexpr bool YEv = decltype(YEc)::value;
      if (!mask) return;
      FlowParams q = p;
      q.seg_mask = mask;
      const long it = (long)d->N * groups * __builtin_popcountll(mask);
      const unsigned g = (unsigned)((it + 3) / 4);
      if (ns1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DTv, 1, 2, 1, 1, true, YEv>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((mbflow_kernel<DTv, 1, 2, 1, 1, true, YEv>), dim3(g), dim3(kFlowThreads), lds, stream, q);
        return;
      }
      if constexpr (!YEv) {
        // NCHW image of even width, 4-byte aligned: the interior instance reads it as aligned dwords (three loads per lane,
        // row and strip instead of eight; SSDK_STEM_DWORD=0: the 2-byte gather)
        static const int env_xd = getenv("SSDK_STEM_DWORD") ? atoi(getenv("SSDK_STEM_DWORD")) : 1;
        if (env_xd && q.layout == 1 && (q.Wimg & 1) == 0 && (reinterpret_cast<uintptr_t>(q.x) & 3u) == 0) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DTv, 1, 2, 1, kFlowNS, true, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          hipLaunchKernelGGL((mbflow_kernel<DTv, 1, 2, 1, kFlowNS, true, false, false, true>), dim3(g), dim3(kFlowThreads), lds, stream, q);
          return;
        }
      }
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbflow_kernel<DTv, 1, 2, 1, kFlowNS, true, YEv>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((mbflow_kernel<DTv, 1, 2, 1, kFlowNS, true, YEv>), dim3(g), dim3(kFlowThreads), lds, stream, q);
    };
    if (d->dtype == SSDK_BF16) {
      go(std::integral_constant<int, SSDK_BF16>{}, std::false_type{}, all & ~edge);
      go(std::integral_constant<int, SSDK_BF16>{}, std::true_type{}, edge);
    } else {
      go(std::integral_constant<int, SSDK_F16>{}, std::false_type{}, all & ~edge);
      go(std::integral_constant<int, SSDK_F16>{}, std::true_type{}, edge);
    }
    dbg_print();
    return 0;
  }
  if (d->dtype == SSDK_BF16) ok = d->stride == 1 ? flow_dispatch<SSDK_BF16, 1>(p, nch, nfo, grid, stream) : flow_dispatch<SSDK_BF16, 2>(p, nch, nfo, grid, stream);
  else ok = d->stride == 1 ? flow_dispatch<SSDK_F16, 1>(p, nch, nfo, grid, stream) : flow_dispatch<SSDK_F16, 2>(p, nch, nfo, grid, stream);
  if (ok) dbg_print();
  return ok ? 0 : 1;
}

}  // namespace ssdk
