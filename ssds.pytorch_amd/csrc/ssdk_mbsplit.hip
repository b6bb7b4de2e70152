// ssdk_mbsplit.hip -- MobileNetV2 inverted-residual block (nets/mobilenet.py:56, 84-89) for the MID-RESOLUTION blocks
// (128^2 ... 32^2 maps at batch 64), with the expanded tensor in registers like ssdk_mbflow.hip and the HIDDEN CHANNELS OF
// A STRIP PAIR SPLIT OVER THE WAVES OF A WORKGROUP.
//
// Why: ssdk_mbflow.hip gives one wave every hidden channel of its work item (image, row segment, pair of 16-pixel column
// strips); a 64x64 map at batch 64 then has 768 items of 18 rows x 12 chunks -- fewer waves than the chip has SIMDs,
// each a long serial chain -- so those blocks ran on the LDS-tiled kernel (ssdk_mbconv.hip), whose expand -> LDS ->
// depthwise -> LDS -> project phases between workgroup barriers reach ~15 % of the VALU / MFMA time they need (61 us for
// a block whose packed-FMA work is 11 us).  Here the NW waves of a workgroup share ONE item and each owns NCHW of its
// NW * NCHW hidden chunks of 16 channels: expand MFMA -> BN bias / ReLU6 -> depthwise by DPP lane shifts -> its share of
// the projection, all in registers exactly as in ssdk_mbflow.hip.  What the split adds is one exchange per OUTPUT ROW:
// every wave leaves its partial projection sums (fp32, fragment order) in LDS, one workgroup barrier, and wave w reduces
// the fragments it owns in wave order 0 .. NW-1 (bit-reproducible), applies the projection BN (+ residual) and stores.
// The exchange moves Cout/Chid of what the tiled kernel moved through LDS (the 6x-expanded tensor, twice).
//
// Numerics: those of ssdk_mbflow.hip except that the projection's fp32 sum over the hidden channels is formed as NW
// partial sums added in wave order.
#include <type_traits>

#include "ssdk_common.h"
#include "ssdk_flow_common.h"

namespace ssdk {

// LDS image (bytes), all 16-byte aligned.  NCH = NW * NCHW hidden chunks, KS = expand k-steps, TW = projection k-steps
// per wave (its NCHW chunks in pairs), F = output fragments per row (accumulator sets x NFO), XB = exchange buffers.
template <int NCHW, int NW, int KS, int NFO, int NA, int XB>
struct SplitLds {
  static constexpr int NCH = NCHW * NW;
  static constexpr int TW = (NCHW + 1) / 2;
  static constexpr int F = NA * NFO;
  static constexpr int we = 0;                                  // [NCH][KS][64 lanes] u32x4
  static constexpr int be = we + NCH * KS * 1024;               // [NCH][4 fg] f32x4
  static constexpr int wd = be + NCH * 4 * 16;                  // [NCH][9 taps][4 fg] 8 bytes
  static constexpr int bd = wd + NCH * 9 * 4 * 8;               // [NCH][4 fg] 8 bytes
  static constexpr int wp = (bd + NCH * 4 * 8 + 15) & ~15;      // [NFO][NW * TW][64 lanes] u32x4
  static constexpr int spb = wp + NFO * NW * TW * 1024;         // [NFO][4 fg][sp f32x4 | bp f32x4]
  static constexpr int xch = spb + NFO * 4 * 32;                // [XB][NW][F][64 lanes] f32x4
  static constexpr int bytes = xch + XB * NW * F * 1024;
};

template <int DT, int S, int KS, int NCHW, int NW, int NFO, int XB, bool TREG>
__global__ __launch_bounds__(64 * NW, 2) void mbsplit_kernel(const FlowParams p) {
  constexpr int NS = 2;                          // strips per item
  constexpr bool MERGE = S == 2;                 // stride 2: even / odd input columns, one accumulator set (ssdk_mbflow.hip header)
  constexpr int NA = MERGE ? 1 : NS;
  using L = SplitLds<NCHW, NW, KS, NFO, NA, XB>;
  constexpr int NCH = L::NCH, TW = L::TW, F = L::F;
  constexpr int NT = 64 * NW;
  constexpr int FPW = (F + NW - 1) / NW;         // fragments a wave finalizes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));  // this wave's chunk group
  const int cb = wv * NCHW;                                         // its first hidden chunk
  const int Cin = p.Cin, Chid = p.Chid, Cout = p.Cout;

  // ---- stage the weights of ALL chunks once per workgroup (layouts of ssdk_mbflow.hip) ------------------------------
  for (u32 i = tid; i < (u32)(NCH * KS * 64); i += NT) {  // expand weights as A fragments [chunk][ks]: row 16c + fr, k = 32ks + 8fg ..
    const u32 l = i & 63u, ck = i >> 6, ks = ck % (u32)KS, c = ck / (u32)KS;
    const u32 hc = c * 16 + (l & 15u), k0 = ks * 32 + (l >> 4) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (hc < (u32)Chid && k0 < (u32)Cin) {
      v = *reinterpret_cast<const u32x4*>(p.we + (size_t)hc * Cin + k0);
      const float sc = p.se[hc];  // the host folds the BN scale before rounding (scale == 1); else folded here
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
  for (u32 i = tid; i < (u32)(NCH * 4); i += NT) {  // expand BN bias, depthwise bias: channels 16c + 4g ..
    const u32 hc = (i >> 2) * 16 + (i & 3u) * 4;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    uint2 d = make_uint2(0u, 0u);
    if (hc < (u32)Chid) {
      b = *reinterpret_cast<const f32x4*>(p.be + hc);
      d = *reinterpret_cast<const uint2*>(p.bd + hc);
      d = make_uint2(fl_sixth_h2(d.x), fl_sixth_h2(d.y));  // (units of six, ssdk_flow_common.h)
    }
    *reinterpret_cast<f32x4*>(smem + L::be + i * 16) = b;
    *reinterpret_cast<uint2*>(smem + L::bd + i * 8) = d;
  }
  for (u32 i = tid; i < (u32)(NCH * 9 * 4); i += NT) {  // depthwise taps [c][tap][g]
    const u32 g = i & 3u, tap = (i >> 2) % 9u, c = (i >> 2) / 9u, hc = c * 16 + g * 4;
    uint2 d = make_uint2(0u, 0u);
    if (hc < (u32)Chid) d = *reinterpret_cast<const uint2*>(p.wd + (size_t)tap * Chid + hc);
    *reinterpret_cast<uint2*>(smem + L::wd + i * 8) = d;
  }
  // projection weights [f][w * TW + t]: k-step t of wave w pairs ITS chunks 2t, 2t+1; lane group fg, element j <->
  // hidden channel (w * NCHW + 2t + j/4) * 16 + 4fg + j%4 (zero where 2t + j/4 >= NCHW or beyond Chid)
  for (u32 i = tid; i < (u32)(NFO * NW * TW * 64 * 2); i += NT) {
    const u32 half = i & 1u, l = (i >> 1) & 63u, ft = i >> 7, tg = ft % (u32)(NW * TW), f = ft / (u32)(NW * TW);
    const u32 w = tg / (u32)TW, t = tg % (u32)TW, lc = 2 * t + half;
    const u32 co = f * 16 + (l & 15u), hc = (w * NCHW + lc) * 16 + (l >> 4) * 4;
    uint2 d = make_uint2(0u, 0u);
    if (lc < (u32)NCHW && co < (u32)Cout && hc < (u32)Chid) d = *reinterpret_cast<const uint2*>(p.wp + (size_t)co * Chid + hc);
    *reinterpret_cast<uint2*>(smem + L::wp + (ft * 64 + l) * 16 + half * 8) = d;
  }
  for (u32 i = tid; i < (u32)(NFO * 4); i += NT) {  // projection BN: output channels 16f + 4g ..
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

  // ---- the workgroup's item: (image, row segment, pair of strips); every wave of the workgroup shares it -------------
  const u32 item = blockIdx.x;
  const int groups = (p.strips + NS - 1) / NS;
  const u32 per_img = (u32)(groups * p.segs);
  const int n = (int)(item / per_img), rem = (int)(item % per_img);
  const int seg = rem / groups, grp = rem % groups;
  constexpr int OW = S == 1 ? 14 : 7;
  const int oy0 = seg * p.rs, oy1 = (oy0 + p.rs < p.Ho ? oy0 + p.rs : p.Ho) - 1;  // output rows [oy0, oy1]
  int ix[NS], oxl[NA];
  bool col_ok[NS], out_lane[NA];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int strip = grp * NS + s;
    const int ox0 = strip * OW;
    ix[s] = MERGE ? 2 * (grp * 15 + (int)fr) - s : ox0 * S - 1 + (int)fr;
    col_ok[s] = strip < p.strips && (unsigned)ix[s] < (unsigned)p.W;
    if constexpr (!MERGE) {
      oxl[s] = ox0 + (int)fr - 1;
      out_lane[s] = strip < p.strips && fr >= 1u && fr <= 14u && oxl[s] < p.Wo;
    }
  }
  if constexpr (MERGE) {  // lane j <= 14: output grp * 15 + j (lane 15 has no right tap)
    oxl[0] = grp * 15 + (int)fr;
    out_lane[0] = fr <= 14u && oxl[0] < p.Wo;
  }

  const u16* ximg = p.x + (size_t)n * p.H * p.W * Cin;
  struct XRow {
    u32x4 k[KS];
  };
  auto load_x = [&](int iy, int s) -> XRow {  // B operand of the expand GEMM: 8 input channels of pixel (iy, ix[s]) per k-step
    XRow out;
    const bool ok = col_ok[s] && (unsigned)iy < (unsigned)p.H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      out.k[ks] = u32x4{0u, 0u, 0u, 0u};
      if (ok && (u32)(ks * 32) + fg * 8u < (u32)Cin)
        out.k[ks] = *reinterpret_cast<const u32x4*>(ximg + ((size_t)iy * p.W + ix[s]) * Cin + ks * 32 + fg * 8);
    }
    return out;
  };

  fl_h2 accA[NA][NCHW * 2], accB[NA][NCHW * 2], accC[NA][NCHW * 2];
  int xrow = 0;  // output rows exchanged so far (exchange buffer parity)
  // TREG: a wave owns only NCHW chunks, so its depthwise taps and biases fit in registers for the whole item (NCHW * 24
  // registers) instead of being re-read from LDS for every row: 9 ds_read_b64 per chunk and row (which the compiler pairs
  // into ds_read2_b64, half the LDS rate) were ~3/4 of the kernel's LDS cycles, and the LDS pipe is shared by 12 waves
  uint2 wtr[TREG ? NCHW : 1][TREG ? 9 : 1], bdr[TREG ? NCHW : 1];
  f32x4 bvr[TREG ? NCHW : 1];
  if constexpr (TREG) {
#pragma unroll
    for (int c = 0; c < NCHW; ++c) {
#pragma unroll
      for (int t = 0; t < 9; ++t) wtr[c][t] = *reinterpret_cast<const uint2*>(smem + L::wd + (((cb + c) * 9 + t) * 4 + (int)fg) * 8);
      bdr[c] = *reinterpret_cast<const uint2*>(smem + L::bd + ((cb + c) * 4 + (int)fg) * 8);
      bvr[c] = *reinterpret_cast<const f32x4*>(smem + L::be + ((cb + c) * 4 + (int)fg) * 16);
    }
  }

  auto row = [&](const XRow (&xraw)[NS], int iy, auto FINc, auto MIDc, auto INIc, fl_h2 (&fin)[NA][NCHW * 2],
                 fl_h2 (&mid)[NA][NCHW * 2], fl_h2 (&ini)[NA][NCHW * 2], int oy_fin) {
    constexpr bool FIN = decltype(FINc)::value, MID = decltype(MIDc)::value, INI = decltype(INIc)::value;
    fl_f2 hi[NS];  // 1/6, or 0 for a pixel outside the image (the zero padding of the EXPANDED tensor)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float k = (col_ok[s] && (unsigned)iy < (unsigned)p.H) ? kFlSixth : 0.f;
      hi[s] = fl_f2{k, k};
    }
    const bool store_row = FIN && oy_fin >= oy0 && oy_fin <= oy1;  // uniform over the workgroup
    // residual of the fragments this wave finalizes: issued early, consumed after the exchange (stride 1 only)
    uint2 resv[FPW];
    if (!MERGE && FIN && store_row && p.residual) {
#pragma unroll
      for (int q = 0; q < FPW; ++q) {
        const int fi = wv + q * NW, a = fi / NFO, f = fi % NFO;
        const int co = f * 16 + (int)fg * 4;
        uint2 r = make_uint2(0u, 0u);
        const bool ol = (NA == 1 || a == 0) ? out_lane[0] : out_lane[NA - 1];
        const int ox = (NA == 1 || a == 0) ? oxl[0] : oxl[NA - 1];
        if (fi < F && ol && co < Cout) r = *reinterpret_cast<const uint2*>(ximg + ((size_t)oy_fin * p.W + ox) * Cin + co);
        resv[q] = r;
      }
    }
    asm volatile("" ::: "memory");  // weights are RE-READ from LDS every row (broadcast reads), not kept in registers
    auto expand = [&](int c, f32x4 (&e)[NS]) {  // chunk c of this wave: D[hc = 16(cb+c) + 4fg + r][px = fr], from the BN bias
      f32x4 bv;
      if constexpr (TREG) bv = bvr[c];
      else bv = *reinterpret_cast<const f32x4*>(smem + L::be + ((cb + c) * 4 + (int)fg) * 16);
#pragma unroll
      for (int s = 0; s < NS; ++s) e[s] = bv;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 wa = *reinterpret_cast<const u32x4*>(smem + L::we + (((cb + c) * KS + ks) * 64 + (int)lane) * 16);
#pragma unroll
        for (int s = 0; s < NS; ++s) e[s] = fl_mfma<DT>(wa, xraw[s].k[ks], e[s]);
      }
    };
    f32x4 e_cur[NS], e_nxt[NS];
    expand(0, e_cur);
#pragma unroll
    for (int c = 0; c < NCHW; ++c) {
      asm volatile("" ::: "memory");
      uint2 wt[9];
      uint2 bdi = make_uint2(0u, 0u);  // the depthwise bias is the value a fresh accumulator starts from (ky = 0 row)
      if constexpr (TREG) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = wtr[c][t];
        bdi = bdr[c];
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const uint2*>(smem + L::wd + (((cb + c) * 9 + t) * 4 + (int)fg) * 8);
        if constexpr (INI) bdi = *reinterpret_cast<const uint2*>(smem + L::bd + ((cb + c) * 4 + (int)fg) * 8);
      }
      if (c + 1 < NCHW) expand(c + 1, e_nxt);  // the matrix pipe runs chunk c+1 while the VALU works on chunk c
      __builtin_amdgcn_sched_barrier(0);
      u32 ew[NS][2];
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
          asm volatile("" : "+v"(s0), "+v"(s1));  // (pinned: see ssdk_mbflow.hip -- MachineSink below the projection)
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
      __builtin_amdgcn_sched_barrier(0);  // chunks are independent: keep the scheduler from hoisting every chunk's reads
    }
    if constexpr (FIN) {
      if (store_row) {
        // ---- this wave's chunks of the output row: ReLU6, then they ARE the B operand of its projection k-steps --------
        asm volatile("" ::: "memory");
        f32x4 yacc[NA][NFO];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int f = 0; f < NFO; ++f) yacc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          u32x4 db[NA];
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            db[a][0] = fl_as_u32(fin[a][4 * t]);
            db[a][1] = fl_as_u32(fin[a][4 * t + 1]);
            if (2 * t + 1 < NCHW) {
              db[a][2] = fl_as_u32(fin[a][(4 * t + 2 < NCHW * 2) ? 4 * t + 2 : 0]);
              db[a][3] = fl_as_u32(fin[a][(4 * t + 3 < NCHW * 2) ? 4 * t + 3 : 0]);
            } else {
              db[a][2] = 0u;
              db[a][3] = 0u;
            }
          }
#pragma unroll
          for (int f = 0; f < NFO; ++f) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(smem + L::wp + (((f * NW + wv) * TW + t) * 64 + (int)lane) * 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) yacc[a][f] = fl_mfma<SSDK_F16>(wf, db[a], yacc[a][f]);  // D[co = 16f + 4fg + r][px = fr]
          }
        }
        // ---- exchange: partial sums of every fragment to LDS, barrier, the owner of a fragment adds them in wave order ----
        unsigned char* xb = smem + L::xch + (XB > 1 ? (xrow & 1) * (NW * F * 1024) : 0);
        if constexpr (XB == 1) __syncthreads();  // the previous row's readers are done with the buffer
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int f = 0; f < NFO; ++f)
            *reinterpret_cast<f32x4*>(xb + ((wv * F + a * NFO + f) * 64 + (int)lane) * 16) = yacc[a][f];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FPW; ++q) {
          const int fi = wv + q * NW;  // wave-uniform
          if (fi < F) {
            const int a = fi / NFO, f = fi % NFO;
            f32x4 y = *reinterpret_cast<const f32x4*>(xb + ((0 * F + fi) * 64 + (int)lane) * 16);
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(xb + ((w2 * F + fi) * 64 + (int)lane) * 16);
              y = y + t;
            }
            const int co = f * 16 + (int)fg * 4;
            const bool ol = NA == 1 ? out_lane[0] : (a == 0 ? out_lane[0] : out_lane[NA - 1]);
            const int ox = NA == 1 ? oxl[0] : (a == 0 ? oxl[0] : oxl[NA - 1]);
            if (ol && co < Cout) {
              const f32x4 spv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32);
              const f32x4 bpv = *reinterpret_cast<const f32x4*>(smem + L::spb + (f * 4 + (int)fg) * 32 + 16);
              u32 h01 = fl_pack2<DT>(fmaf(y[0], spv[0], bpv[0]), fmaf(y[1], spv[1], bpv[1]));
              u32 h23 = fl_pack2<DT>(fmaf(y[2], spv[2], bpv[2]), fmaf(y[3], spv[3], bpv[3]));
              if (!MERGE && p.residual) {  // rounded to the model dtype first, then x is added (torch's tensor add)
                const u32 x01 = resv[q].x, x23 = resv[q].y;
                h01 = fl_pack2<DT>(fl_from16<DT>(h01 & 0xffffu) + fl_from16<DT>(x01 & 0xffffu), fl_from16<DT>(h01 >> 16) + fl_from16<DT>(x01 >> 16));
                h23 = fl_pack2<DT>(fl_from16<DT>(h23 & 0xffffu) + fl_from16<DT>(x23 & 0xffffu), fl_from16<DT>(h23 >> 16) + fl_from16<DT>(x23 >> 16));
              }
              u16* yrow = p.y + (((size_t)n * p.Ho + oy_fin) * p.Wo + ox) * Cout;
              *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h01, h23);
            }
          }
        }
        ++xrow;
      }
    }
  };

  const auto Y = std::true_type{};
  const auto No = std::false_type{};
  auto load_row = [&](XRow (&dst)[NS], int iy) {
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[s] = load_x(iy, s);
  };
  // Row loop as in ssdk_mbflow.hip: three accumulator sets rotate through fixed registers (groups of 6 / 4 rows), x one
  // row ahead; surplus rows past oy1 compute into accumulators nobody stores.
  XRow xa[NS], xb2[NS];
  if constexpr (S == 1) {
    const int rend = oy1 + 1;
    load_row(xa, oy0 - 1);
    for (int r = oy0 - 1; r <= rend; r += 6) {
      load_row(xb2, r + 1);
      row(xa, r, Y, Y, Y, accA, accB, accC, r - 1);
      load_row(xa, r + 2);
      row(xb2, r + 1, Y, Y, Y, accB, accC, accA, r);
      load_row(xb2, r + 3);
      row(xa, r + 2, Y, Y, Y, accC, accA, accB, r + 1);
      load_row(xa, r + 4);
      row(xb2, r + 3, Y, Y, Y, accA, accB, accC, r + 2);
      load_row(xb2, r + 5);
      row(xa, r + 4, Y, Y, Y, accB, accC, accA, r + 3);
      load_row(xa, r + 6);
      row(xb2, r + 5, Y, Y, Y, accC, accA, accB, r + 4);
    }
  } else {
    const int rend = 2 * oy1 + 1;
    load_row(xa, 2 * oy0 - 1);
    for (int r = 2 * oy0 - 1; r <= rend; r += 4) {
      load_row(xb2, r + 1);
      row(xa, r, Y, No, Y, accA, accC, accB, (r - 1) / 2);          // odd: finishes A, starts B
      load_row(xa, r + 2);
      row(xb2, r + 1, No, Y, No, accC, accB, accC, 0);              // even: middle of B
      load_row(xb2, r + 3);
      row(xa, r + 2, Y, No, Y, accB, accC, accA, (r + 1) / 2);      // odd: finishes B, starts A
      load_row(xa, r + 4);
      row(xb2, r + 3, No, Y, No, accC, accA, accC, 0);              // even: middle of A
    }
  }
}

template <int DT, int S, int KS, int NCHW, int NW, int NFO, int XB, bool TREG>
static void split_launch(const FlowParams& p, unsigned grid, hipStream_t stream) {
  constexpr int lds = SplitLds<NCHW, NW, KS, NFO, S == 2 ? 1 : 2, XB>::bytes;
  static_assert(lds <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbsplit_kernel<DT, S, KS, NCHW, NW, NFO, XB, TREG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((mbsplit_kernel<DT, S, KS, NCHW, NW, NFO, XB, TREG>), dim3(grid), dim3(64 * NW), lds, stream, p);
}

template <int DT>
static bool split_dispatch(const FlowParams& p, int stride, int ks, int nch, int nfo, int xb, unsigned grid, hipStream_t stream) {
#define SSDK_SPLIT(S_, KS_, NCHW_, NW_, NFO_)                                                   \
  if (stride == S_ && ks == KS_ && nch == NCHW_ * NW_ && nfo == NFO_) {                           \
    if (xb == 1) split_launch<DT, S_, KS_, NCHW_, NW_, NFO_, 1, false>(p, grid, stream);          \
    else if (xb == 2) split_launch<DT, S_, KS_, NCHW_, NW_, NFO_, 2, false>(p, grid, stream);     \
    else split_launch<DT, S_, KS_, NCHW_, NW_, NFO_, 1, true>(p, grid, stream);                   \
    return true;                                                                                  \
  }
  SSDK_SPLIT(2, 1, 3, 3, 2)   // 24 -> 144 -> 32, stride 2 (128^2 -> 64^2)
  SSDK_SPLIT(1, 1, 3, 4, 2)   // 32 -> 192 -> 32 (64^2)
  SSDK_SPLIT(2, 1, 3, 4, 4)   // 32 -> 192 -> 64, stride 2 (64^2 -> 32^2)
#undef SSDK_SPLIT
  return false;
}

// Returns 1 when the block is not one of this kernel's (the caller then runs ssdk_mbconv.hip's), 0 after a launch.
int launch_mbsplit(const ssdk_mbconv_desc* d, hipStream_t stream) {
  static const int env = getenv("SSDK_MB_SPLIT") ? atoi(getenv("SSDK_MB_SPLIT")) : 1;
  const int variant = d->variant;  // 0 auto, 2 this kernel wherever it exists (tests), other non-zero values: never
  if ((!env && variant != 2) || (variant != 0 && variant != 2)) return 1;
  if (d->stem || (d->Cin % 8) || (d->Chid % 16) || (d->Cout % 8)) return 1;
  const int ks = (d->Cin + 31) / 32, nch = d->Chid / 16, nfo = d->Cout <= 32 ? 2 : (d->Cout <= 64 ? 4 : (d->Cout + 15) / 16);
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
  p.layout = 0;
  p.Chid = d->Chid;
  p.Cout = d->Cout;
  p.Ho = (p.H + 2 - 3) / d->stride + 1;
  p.Wo = (p.W + 2 - 3) / d->stride + 1;
  p.residual = d->residual;
  p.seg_mask = 0;
  p.dbg = nullptr;
  const int ow = d->stride == 1 ? 14 : 7;
  p.strips = (p.Wo + ow - 1) / ow;
  if (d->stride == 2) p.strips = 2 * ((p.Wo + 14) / 15);  // parity split: a pair of strips = 15 outputs (ssdk_mbflow.hip header)
  const int groups = (p.strips + 1) / 2;
  // rows per segment: the longest of 32 / 16 / 8 that still gives the chip ~3 workgroups per CU
  constexpr int env_rs = 0;  // (round 6: the SSDK_MB_SPLIT_RS switch is gone, its A/B is settled)
  constexpr int env_xb = 1;  // (round 6: the SSDK_MB_SPLIT_XB switch is gone, its A/B is settled)  // 1 | 2 exchange buffers, 3: one buffer + taps in registers
  // Measured on SSD-MobileNetV2@512, batch 64 (profiles/r04_split_ab.txt): 16-row segments beat 8 (two halo rows per
  // segment: 71 vs 64 us on the 128^2 block) and 32 (too few workgroups: 82 us); three workgroups per CU (one exchange
  // buffer, two barriers per row) beat two (double-buffered exchange: 74 us) and beat two with the taps in registers (72 us)
  // -- resident waves matter more than LDS reads here; and a map that yields fewer than ~700 workgroups of 16 rows (the
  // 64^2 -> 32^2 stride-2 block: 384) stays on the tiled kernel (38 vs 47-52 us).
  int rs = 16;
  constexpr int env_min = 700;  // (round 6: the SSDK_MB_SPLIT_MIN switch is gone, its A/B is settled)
  if (variant == 2)  // forced (tests): short segments so that small maps still exercise several segments
    while (rs > 4 && (long)d->N * groups * ((p.Ho + rs - 1) / rs) < 64) rs >>= 1;
  if (env_rs > 0) rs = env_rs;
  if (rs > p.Ho) rs = p.Ho;
  p.rs = rs;
  p.segs = (p.Ho + rs - 1) / rs;
  const long items = (long)d->N * groups * p.segs;
  if (items < env_min && variant != 2) return 1;
  const unsigned grid = (unsigned)items;
  const bool ok = d->dtype == SSDK_BF16 ? split_dispatch<SSDK_BF16>(p, d->stride, ks, nch, nfo, env_xb, grid, stream)
                                        : split_dispatch<SSDK_F16>(p, d->stride, ks, nch, nfo, env_xb, grid, stream);
  return ok ? 0 : 1;
}

}  // namespace ssdk
