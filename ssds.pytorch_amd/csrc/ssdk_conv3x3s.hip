// ssdk_conv3x3s.hip -- 3x3 / stride-1 convolution with a SHORT K (Cin <= 128) on the matrix cores: the multibox head of the
// first SSD level (reference ssd.py:100-103 on the 96-channel 32x32 map: K = 864, N = 504, M = 65 536 at batch 64).
//
// Why a second halo kernel: conv3x3_halo_kernel owns all 160 KiB of LDS (two 64-channel halo slabs + a weight ring), i.e. ONE
// workgroup per CU, and with K = 864 its main loop is only half of a tile's time -- address set-up, the first halo round
// trip and above all the epilogue (sigmoid on 480 of 504 columns, 16-bit conversion, stores) run with the matrix cores idle
// (measured per tile: 4.2 k + 2.2 k cycles before, 12.3 k + 3.6 k cycles behind a 25.6 k loop: 24 % of the MFMA peak).
// Here K is short enough for the WHOLE halo of a 128-pixel patch to sit in LDS ((th+2)(tw+2) rows of Cin channels: 42 KiB at
// Cin = 96), so a workgroup keeps its patch and walks over ALL output-channel tiles:
//
//   * workgroup = one 128-pixel patch (th x tw of one image), 4 waves as 1 (M) x 4 (N); per 128-channel tile 8 x 2
//     accumulator fragments of v_mfma_f32_16x16x32 per wave; the halo is staged ONCE for the ceil(Cout / 128) tiles, and
//     nothing else ever writes LDS: there is no barrier behind the staging;
//   * k-step = (tap, 32-channel slice), taps outside: the A fragment of a lane is 16 bytes of the halo row of "its" pixel
//     shifted by the tap -- an address offset; rows are padded to an odd number of 16-byte chunks (conflict-free ds_read_b128);
//   * the B fragments (weights) go straight from the fragment-major image (ssdk.h ssdk_weight_frag_bytes) into operand
//     registers: a wave's 32 channels x 32 k are two coalesced 1 KiB loads, three k-steps ahead through three register
//     sets (9 * Cin / 32 k-steps per tile is a multiple of three: the ring runs on across tiles).  Measured against staging
//     them through a three-deep LDS ring with one barrier per k-step: the same 78 us -- the loop was not what bound it;
//   * what bound it was the epilogue (sigmoid on 480 of 504 columns: ~60 VALU instructions per fragment, 25 us over the
//     launch).  The epilogue of tile t is INTERLEAVED with the main loop of tile t + 1: the finished accumulators move to a
//     second register set and one fragment per k-step is scaled / activated / converted / stored, its instructions issued
//     BETWEEN that step's MFMAs (sched_group_barrier: one MFMA, four VALU, ...): a MFMA holds the matrix pipe for 16 cycles
//     but the issue port for 4.  For that the fragment has to be straight-line code in the MFMAs' basic block: activation
//     classes are template parameters, the per-lane activation select is two plain selects, and stores are raw buffer
//     stores whose out-of-range offset masks a lane (no exec-mask branch).  (First version: one tile per workgroup, two
//     workgroups per CU -- all workgroups start together, so both were in their epilogue at the same time: loop 44 us +
//     epilogue 30 us + staging 10 us ran back to back.  Second: epilogue code behind the 16 MFMAs of its k-step, in program
//     order: no overlap at all, 78 us.  Interleaved: 66-68 us.)
//   * stores go straight from the accumulators: with pixels as the A operand a lane holds four consecutive pixels of one
//     channel = 8 contiguous bytes of an NCHW plane (split loc | conf heads; Wo % 4 == 0).  For NHWC output (tower layers)
//     the operands swap and the lane holds four consecutive channels of one pixel (Cout % 4 == 0).
//   * Cin > 128 (the halo no longer fits twice: FPN / BiFPN tower layers with 256 channels, the 320-channel SSD head): ONE
//     workgroup per CU, i.e. one wave per SIMD and nobody to hide its LDS latency -- the A fragments of the next k-step are
//     read into a second register set before the current step's MFMAs are issued (512 registers are available there).
//     The same trick measured SLOWER with two workgroups per CU (the other workgroup's waves already fill the gap).
//   * a workgroup takes all channel tiles of its patch, or -- where that leaves fewer workgroups than CUs (the 16x16 SSD
//     level: 128 patches) -- 1 / nsplit of them.
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int S3_THREADS = 256, S3_BN = 128, S3_PIX = 128;

struct ShortParams {
  ConvParams c;
  int th, tw, tw_shift;  // patch (th * tw == 128, tw a power of two >= 8)
  int tiles_y, tiles_x, n_tiles;
  int nsplit, tpw;       // workgroups per patch, channel tiles per workgroup
  int hw2, hrows;        // tw + 2, (th + 2) * (tw + 2)
  int rs;                // LDS bytes per halo row
  int groups;            // ceil(Cout / 16): 16-row groups of the fragment-major weight image
  unsigned mg_tx, mg_ty, mg_hw2, mg_ns;
  unsigned y1bytes, y2bytes;  // buffer-descriptor ranges of the two outputs (< 4 GiB each)
};

__device__ __forceinline__ u32 s3_div(u32 n, u32 d, u32 M) { return d == 1u ? n : __umulhi(n, M); }

constexpr int s3_wpc(int cs) { return cs <= 4 ? 2 : 1; }  // workgroups per CU the instance is compiled for

// CS = Cin / 32.  NCHW: split-plane output with none | sigmoid | silu per column (the heads); else NHWC output with
// none | relu | relu6 (tower / body layers).  The activation class is fixed at compile time: the epilogue must be
// straight-line code to be scheduled between the MFMAs.
// WIDE (round 6, NCHW only): the two channel fragments (j = 0, 1) of a pixel fragment are finished TOGETHER and trade halves with
// v_permlane16_swap_b32 -- lane rows 0 / 2 end with 8 consecutive pixels of channel j = 0, rows 1 / 3 with 8 of channel j = 1 -- so
// the 68.8 MB of the first head level leave in 16-byte stores instead of 8-byte ones.  Needs the
// lane's 8 pixels in one map row: Wo % 8 == 0.
template <int DT, int CS, bool NCHW, bool WIDE = false>
__global__ __launch_bounds__(S3_THREADS, s3_wpc(CS)) void conv3x3_short_kernel(const ShortParams sp) {
  static_assert(!WIDE || NCHW, "the paired epilogue is the NCHW one");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvParams& p = sp.c;
  // halo row stride (bytes): an odd number of 16-byte chunks.  (Measured against 256-byte rows with the 16-byte slot XOR-ed by
  // (row & 15) -- conflict-free for the non-contiguous lane groups of ds_read_b128 on paper: 6-10 % SLOWER on every shape; the
  // per-tap address rebuild costs more than the conflicts it removes.)
  // Round 4: the row stride is a run-time value (sp.rs) = CS * 64 + 32: ds_read_b128 is serviced in FOUR groups of 16 lanes
  // that are not lane-contiguous ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md): a group mixes eight pixels of
  // one k-slice piece (fg) with the other eight pixels of the NEXT piece (fg + 1).  With R = stride / 16 the sixteen 16-byte
  // slots {fr R, (fr R + 1)} mod 16 of such a group are all different iff R = 2 (mod 4); the former odd R = 4 CS + 1 put 5 of 16
  // lanes on a slot already taken -- every read took two LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50).
  const int RS = sp.rs;
  constexpr int NS = 9 * CS;        // k-steps per channel tile (a multiple of 3)
  constexpr bool PREF = s3_wpc(CS) == 1;  // one wave per SIMD: A fragments one k-step ahead in a second register set
  static_assert(!PREF || NS % 2 == 0, "the two A-fragment sets alternate across tiles");
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  // wave layout 1 (M) x 4 (N): a wave owns ALL 128 pixels x 32 channels = MI x NJ = 8 x 2 fragments.  (2 x 2 waves of 4 x 4
  // fragments need the same 16 MFMAs per k-step but 4 KiB of weights per wave and k-step, half of them a duplicate of the
  // neighbour's: 32 KiB per CU and k-step pair against ~45 B/clk of L2 bandwidth = 711 cycles for 512 of matrix work.
  // Here: 2 KiB of weights and 8 KiB of LDS reads per wave -- 355 / 512 / 512 cycles of L2 / LDS / MFMA.)
  constexpr int MI = 8, NJ = 2;
  constexpr int NC = NCHW ? 1 : 4;  // epilogue constants per fragment column (NHWC: a lane holds four channels)
  const u32 wn = wave, fr = lane & 15u, fg = lane >> 4;

  const u32 pt = s3_div(blockIdx.x, (u32)sp.nsplit, sp.mg_ns);  // the workgroups of a patch are neighbours (its halo in L2)
  const u32 part = blockIdx.x - pt * (u32)sp.nsplit;
  u32 pq = s3_div(pt, (u32)sp.tiles_x, sp.mg_tx);
  const u32 tx = pt - pq * (u32)sp.tiles_x;
  const u32 b = s3_div(pq, (u32)sp.tiles_y, sp.mg_ty);
  const u32 ty = pq - b * (u32)sp.tiles_y;
  const int y0 = (int)ty * sp.th, x0 = (int)tx * sp.tw;
  const u32 t0 = part * (u32)sp.tpw;
  const u32 t1 = t0 + (u32)sp.tpw < (u32)sp.n_tiles ? t0 + (u32)sp.tpw : (u32)sp.n_tiles;

  // ---- weights: the B fragments of this wave's 32 channels (blocks g = 8 nt + 2 wn + j of the image), straight from
  //      global memory into operand registers, three k-steps ahead (uniform block offset + 32-bit lane offset: the loads take
  //      the scalar-base form)
  const unsigned char* wbase = (const unsigned char*)p.w_frag;
  const u32 wlane = lane * 16u;
  auto woff = [&](u32 nt, int j) -> size_t {  // byte offset of block (g, k-step 0): uniform
    u32 g = nt * 8u + (u32)NJ * wn + (u32)j;
    g = g < (u32)sp.groups ? g : (u32)sp.groups - 1u;  // groups past Cout: any valid block (never stored)
    return (size_t)g * NS * 1024;
  };
  auto ldw = [&](size_t off) { return *reinterpret_cast<const u32x4*>(wbase + off + wlane); };
  u32x4 fbq[3][NJ];  // register set q holds the B fragments of a k-step s with s % 3 == q
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < NJ; ++j) fbq[s][j] = ldw(woff(t0, j) + (size_t)s * 1024);

  // ---- halo: (th+2) x (tw+2) rows of Cin channels, zeros outside the image, batches of independent 16-byte loads --------
  {
    constexpr int CPR = CS * 4;  // 16-byte pieces per row
    const int total = sp.hrows * CPR;
    const u16* xb = (const u16*)p.x + (size_t)b * p.H * p.W * p.Cin;
    constexpr int SB = PREF ? 16 : 10;
    for (int q0 = (int)tid; q0 < total; q0 += S3_THREADS * SB) {
      u32x4 v[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int q = q0 + j * S3_THREADS;
        const u32 row = (u32)q / (u32)CPR, cpiece = (u32)q - row * (u32)CPR;
        const u32 hy = s3_div(row, (u32)sp.hw2, sp.mg_hw2), hx = row - hy * (u32)sp.hw2;
        const int iy = y0 + (int)hy - 1, ix = x0 + (int)hx - 1;
        v[j] = u32x4{0u, 0u, 0u, 0u};
        if (q < total && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v[j] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * p.W + ix) * p.Cin + cpiece * 8);
      }
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int q = q0 + j * S3_THREADS;
        const u32 row = (u32)q / (u32)CPR, cpiece = (u32)q - row * (u32)CPR;
        if (q < total) *reinterpret_cast<u32x4*>(smem + (size_t)row * RS + cpiece * 16) = v[j];
      }
    }
  }

  // ---- fragment roles ----------------------------------------------------------------------------------------------------
  u32 a_ad[MI];  // LDS byte address of pixel fragment i at tap (0, 0), slice 0
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const u32 ml = (u32)i * 16u + fr;
    const u32 y = ml >> sp.tw_shift, x = ml & (u32)(sp.tw - 1);
    a_ad[i] = (y * (u32)sp.hw2 + x) * (u32)RS + fg * 16u;
  }
  // output pixel(s) of this lane in fragment i.  NCHW: accumulator rows = pixels 16 i + 4 fg .. +3 (one map row);
  // NHWC (operands swapped): accumulator column = pixel 16 i + fr
  u32 o_off[MI];  // oy * Wo + ox, or ~0 outside the map
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const u32 ml = (u32)i * 16u + (NCHW ? (WIDE ? (fg >> 1) * 8u : fg * 4u) : fr);
    const int oy = y0 + (int)(ml >> sp.tw_shift), ox = x0 + (int)(ml & (u32)(sp.tw - 1));
    o_off[i] = (oy < p.Ho && ox < p.Wo) ? (u32)(oy * p.Wo + ox) : 0xffffffffu;
  }
  const u32 hw = (u32)(p.Ho * p.Wo);

  f32x4 acc[MI][NJ], prev[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // epilogue constants of the tile whose accumulators sit in `prev` (kept small where two workgroups share a CU: the main
  // loop lives at the 256-register limit -- activation selectors and plane offsets are rebuilt per fragment from the channel
  // number, in the MFMAs' shadow).  NCHW: column j = channel e_n0 + 16 j; NHWC: rows = channels e_n0 + 16 j .. + 3
  float e_sc[NJ][NC], e_bi[NJ][NC];
  u32 e_n0 = 0;
  const ActSel as_a = act_sel(p.act), as_b = act_sel(p.act2);
  // Branch-free: raw buffer stores, one per output tensor (loc | conf planes); a lane that has nothing to store there carries
  // an out-of-range offset and the hardware drops it -- no exec-mask branch, so the whole fragment stays in the basic block of
  // the MFMAs it is scheduled between.
  const __amdgpu_buffer_rsrc_t yr1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, sp.y1bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y2 ? p.y2 : p.y), 0, sp.y2bytes, 0x00020000);
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  auto epi_frag = [&](const f32x4& a, int i, int j) {
    const u32 n = e_n0 + (u32)j * 16u;
    float v[4];
    if constexpr (NCHW) {
      const bool second = (int)n >= p.split;
      const int mode = second ? as_b.mode : as_a.mode;
      // (epilogue4 of ssdk_conv_common.h with the activation select written as two plain selects: as a nested conditional
      //  with a multiply in one arm it compiles to exec-mask branches -- basic-block boundaries the scheduler cannot cross)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = a[r] * e_sc[j][0] + e_bi[j][0];
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
        const float t = v[r] * sg;
        const float u = mode == 1 ? sg : v[r];
        v[r] = mode == 2 ? t : u;
      }
      const uint2 h = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
      const u32 plane = second ? ((b * (u32)(p.Cout - p.split) + (n - (u32)p.split)) * hw) * 2u : ((b * (u32)p.split + n) * hw) * 2u;
      const bool ok = n < (u32)p.Cout && o_off[i] != 0xffffffffu;
      const u32 off = plane + o_off[i] * 2u;
      __builtin_amdgcn_raw_buffer_store_b64(v2u{h.x, h.y}, yr1, (int)(ok && !second ? off : 0xfffffff0u), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(v2u{h.x, h.y}, yr2, (int)(ok && second ? off : 0xfffffff0u), 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = __builtin_fminf(__builtin_fmaxf(a[r] * e_sc[j][r] + e_bi[j][r], as_a.lo), as_a.hi);
      const uint2 h = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
      const bool ok = n < (u32)p.Cout && o_off[i] != 0xffffffffu;  // (Cout % 4 == 0: all four channels or none)
      const u32 off = ((b * hw + o_off[i]) * (u32)p.Cout + n) * 2u;
      __builtin_amdgcn_raw_buffer_store_b64(v2u{h.x, h.y}, yr1, (int)(ok ? off : 0xfffffff0u), 0, 0);
    }
  };
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  auto epi_pair = [&](const f32x4& a0, const f32x4& a1, int i) {  // WIDE: fragments (i, 0) and (i, 1) of the finished tile
    uint2 h[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32 n = e_n0 + (u32)j * 16u;
      const int mode = (int)n >= p.split ? as_b.mode : as_a.mode;
      const f32x4& a = j ? a1 : a0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = a[r] * e_sc[j][0] + e_bi[j][0];
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
        const float t = v[r] * sg;
        const float u = mode == 1 ? sg : v[r];
        v[r] = mode == 2 ? t : u;
      }
      h[j] = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
    }
    // rows (fg) 1 / 3 of the j = 0 words trade places with rows 0 / 2 of the j = 1 words: the first result of a lane in row 0 | 1 |
    // 2 | 3 is (j 0, fg 0) | (j 1, fg 0) | (j 0, fg 2) | (j 1, fg 2), the second (j 0, fg 1) | (j 1, fg 1) | (j 0, fg 3) | (j 1, fg 3)
    const v2u sx = __builtin_amdgcn_permlane16_swap(h[0].x, h[1].x, false, false);
    const v2u sy = __builtin_amdgcn_permlane16_swap(h[0].y, h[1].y, false, false);
    const u32 n = e_n0 + (fg & 1u) * 16u;  // the channel this lane now holds 8 pixels of
    const bool second = (int)n >= p.split;
    const u32 plane = second ? ((b * (u32)(p.Cout - p.split) + (n - (u32)p.split)) * hw) * 2u : ((b * (u32)p.split + n) * hw) * 2u;
    const bool ok = n < (u32)p.Cout && o_off[i] != 0xffffffffu;
    const u32 off = plane + o_off[i] * 2u;
    const v4u o = {sx.x, sy.x, sx.y, sy.y};
    __builtin_amdgcn_raw_buffer_store_b128(o, yr1, (int)(ok && !second ? off : 0xfffffff0u), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(o, yr2, (int)(ok && second ? off : 0xfffffff0u), 0, 0);
  };
  __syncthreads();

  u32x4 fa[PREF ? 2 : 1][MI];
  auto read_a = [&](int set, int ks) {
    const int tap = ks / CS, sl = ks % CS;
    const u32 toff = (u32)((tap / 3) * sp.hw2 + (tap % 3)) * (u32)RS + (u32)sl * 64u;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const u32x4*>(smem + a_ad[i] + toff);
  };
  if (PREF) read_a(0, 0);

  // one channel tile: NS k-steps, completely unrolled (the compiler counts the loads / stores in flight).  EPI: finish the
  // fragments of the previous tile along the way (ceil(16 / NS) per k-step)
  auto run_tile = [&](u32 nt, auto epi_tag) {
    constexpr bool EPI = decltype(epi_tag)::value;
    constexpr int FPS = (16 + NS - 1) / NS;
    const u32 ntn = nt + 1u < t1 ? nt + 1u : nt;  // behind the last tile: harmless re-reads of its own weights
    size_t wc[NJ], wx[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      wc[j] = woff(nt, j);
      wx[j] = woff(ntn, j);
    }
    const u32 nbase = nt * S3_BN + wn * (16u * NJ) + (NCHW ? fr : fg * 4u);
    constexpr int KC = (16 + FPS - 1) / FPS;  // k-step in which this tile's epilogue constants are requested: the previous
                                              // tile's fragments -- which still use e_sc / e_bi -- are finished by then
    static_assert(KC <= NS - 1, "the constants must be requested inside the tile's loop");
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      if (PREF) {
        read_a((ks + 1) & 1, (ks + 1) % NS);  // (the last step of a tile: step 0 of the next, same halo)
        __builtin_amdgcn_sched_barrier(0);    // (pinned in front of this step's MFMAs: the scheduler otherwise sinks the reads
                                              //  to a few MFMAs before their use -- less than the LDS round trip)
      } else {
        read_a(0, ks);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = NCHW ? mfma16<DT>(fa[PREF ? ks & 1 : 0][i], fbq[ks % 3][j], acc[i][j])
                           : mfma16<DT>(fbq[ks % 3][j], fa[PREF ? ks & 1 : 0][i], acc[i][j]);
      if (EPI) {
        if constexpr (WIDE) {  // pairs: fragments 2 q, 2 q + 1 in the k-step that used to finish fragment 2 q
#pragma unroll
          for (int e = ks * FPS; e < (ks + 1) * FPS && e < 16; ++e)
            if ((e & 1) == 0) epi_pair(prev[e / NJ][0], prev[e / NJ][1], e / NJ);
        } else {
#pragma unroll
          for (int e = ks * FPS; e < (ks + 1) * FPS && e < 16; ++e) epi_frag(prev[e / NJ][e % NJ], e / NJ, e % NJ);
        }
      }
      {
        const int k3 = ks + 3;  // the B fragments of step ks + 3 into the set just used
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          fbq[ks % 3][j] = ldw(k3 < NS ? wc[j] + (size_t)k3 * 1024 : wx[j] + (size_t)(k3 - NS) * 1024);
      }
      // issue order inside the k-step: one MFMA, then a few of the epilogue's VALU instructions, 16 times over -- a MFMA
      // holds the matrix pipe for 16 cycles but the issue port for 4, and left to itself the scheduler emits the 16 MFMAs
      // back to back and the epilogue behind them (measured: the "interleaved" epilogue then cost as much as a separate one)
      if (EPI && ks * FPS < 16) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, (WIDE ? 8 : 4) * FPS, 0);  // VALU
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks == KC) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < NC; ++r) {
            const u32 n = nbase + (u32)j * 16u + (u32)r, nn = n < (u32)p.Cout ? n : (u32)p.Cout - 1u;
            const float sv = (p.scale ? p.scale : p.bias)[nn];  // (no branch: a select of the address, then of the value)
            e_sc[j][r] = p.scale ? sv : 1.f;
            e_bi[j][r] = p.bias[nn];
          }
      }
    }
    // hand the accumulators over
    e_n0 = nbase;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        prev[i][j] = acc[i][j];
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  // (the B fragments of steps 0..2 of the first tile are in fbq; step ks loads ks + 3, and every later tile finds its first
  //  three steps loaded by the previous tile's last ones)
  run_tile(t0, std::false_type{});
  for (u32 nt = t0 + 1u; nt < t1; ++nt) run_tile(nt, std::true_type{});
  // the last tile's epilogue has no main loop to hide behind
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if constexpr (WIDE) {
      epi_pair(prev[i][0], prev[i][1], i);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) epi_frag(prev[i][j], i, j);
    }
  }
}

// 1: not one of this kernel's layers (the caller goes on), 0: launched
int launch_conv3x3_short(const ConvParams& p, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_CONV3X3_SHORT") ? atoi(getenv("SSDK_CONV3X3_SHORT")) : 1;  // 2: short K only (Cin <= 128)
  static const int env_pk = getenv("SSDK_WFRAG") ? atoi(getenv("SSDK_WFRAG")) : 1;
  if (!env || !env_pk || !p.w_frag || p.k != 3 || p.stride != 1 || p.pad != 1 || p.H != p.Ho || p.W != p.Wo) return 1;
  if ((p.Cin % 32) || p.Cout < 96 || p.in_layout != LAYOUT_NHWC || p.res || p.post != SSDK_ACT_NONE) return 1;
  const int cs = p.Cin / 32;
  // the instances that exist: Cin <= 128 (two workgroups per CU) and Cin = 256 (one per CU).  Measured against the halo
  // kernel at batch 32 (tools/gemm_probe.py): FPN tower layer 256 -> 256 on 40x40 66 vs 72 us, on 80x80 257 vs 256-280 us
  // (a tie: those maps stay on the halo kernel); a Cin = 320 instance (SSD head L1, 90 k-steps per tile, one wave per SIMD)
  // lost 64 vs 48 us and is not built.
  if (!(cs >= 1 && cs <= 4) && !(env == 1 && cs == 8 && p.Ho * p.Wo <= 2500)) return 1;
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  auto act_ok = [&](int a) {
    return nchw ? (a == SSDK_ACT_NONE || a == SSDK_ACT_SIGMOID || a == SSDK_ACT_SILU) : (a == SSDK_ACT_NONE || a == SSDK_ACT_RELU || a == SSDK_ACT_RELU6);
  };
  if (!act_ok(p.act) || (nchw && !act_ok(p.act2))) return 1;
  if (nchw ? (p.Wo & 3) != 0 : ((p.Cout & 3) != 0 || p.split != p.Cout)) return 1;  // 8-byte stores: four pixels of a row | four channels
  if (p.Ho * p.Wo < S3_PIX || p.Wo < 8) return 1;
  ShortParams sp;
  sp.c = p;
  // patch: th x tw = 128 pixels, tw a power of two; the shape that wastes the fewest tile rows, wide ones preferred (NCHW
  // stores: long runs per channel)
  int best_tw = 0;
  double best_u = 0.0;
  for (int t = 8; t <= 64 && t <= 2 * p.Wo; t *= 2) {
    const int th = S3_PIX / t;
    const double u = (double)p.Ho * p.Wo / ((double)((p.Ho + th - 1) / th) * ((p.Wo + t - 1) / t) * S3_PIX) * (1.0 + 0.01 * t / 64.0);
    if (u > best_u) {
      best_u = u;
      best_tw = t;
    }
  }
  if (best_u < 0.6) return 1;
  sp.tw = best_tw;
  sp.th = S3_PIX / best_tw;
  sp.tw_shift = 0;
  while ((1 << sp.tw_shift) < sp.tw) ++sp.tw_shift;
  sp.tiles_y = (p.Ho + sp.th - 1) / sp.th;
  sp.tiles_x = (p.Wo + sp.tw - 1) / sp.tw;
  sp.n_tiles = (p.Cout + S3_BN - 1) / S3_BN;
  sp.hw2 = sp.tw + 2;
  sp.hrows = (sp.th + 2) * (sp.tw + 2);
  sp.groups = (p.Cout + 15) / 16;
  const long patches = (long)p.N * sp.tiles_y * sp.tiles_x;
  // workgroups per patch: all channel tiles in one workgroup unless that leaves CUs idle
  const int slots = 256 * s3_wpc(cs) / 2;  // (Cin <= 128: two workgroups per CU, but one per CU already fills the matrix pipes)
  int nsplit = 1;
  while (patches * nsplit < slots && nsplit * 2 <= sp.n_tiles) nsplit *= 2;
  if (patches * nsplit < slots || patches * nsplit > 0x7fffffffl) return 1;
  sp.nsplit = nsplit;
  sp.tpw = (sp.n_tiles + nsplit - 1) / nsplit;
  if (sp.tpw < 2) return 1;  // one tile per workgroup: no next tile to hide the epilogue behind -- the halo kernel's case
  constexpr int env_pad = 32;  // (round 6: the SSDK_S3_PAD switch is gone, its A/B is settled)  // (16: the round-3 stride, A/B runs)
  sp.rs = cs * 64 + (env_pad == 16 ? 16 : 32);
  const size_t lds = (size_t)((sp.hrows * sp.rs + 1023) & ~1023);
  if (lds > (size_t)(s3_wpc(cs) == 2 ? 80 : 160) * 1024) return 1;
  auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  sp.mg_tx = magic(sp.tiles_x);
  sp.mg_ty = magic(sp.tiles_y);
  sp.mg_hw2 = magic(sp.hw2);
  sp.mg_ns = magic(sp.nsplit);
  {
    const size_t hw = (size_t)p.Ho * p.Wo;
    const size_t b1 = (size_t)p.N * p.split * hw * 2, b2 = (size_t)p.N * (p.Cout - p.split) * hw * 2;
    if (b1 >= 0xfffffff0ull || b2 >= 0xfffffff0ull) return 1;  // 32-bit buffer offsets
    sp.y1bytes = (unsigned)b1;
    sp.y2bytes = (unsigned)b2;
  }
  const unsigned grid = (unsigned)(patches * nsplit);
  static const int env_wide = getenv("SSDK_S3_WIDE") ? atoi(getenv("SSDK_S3_WIDE")) : 1;
  const bool wide = env_wide && nchw && (p.Wo & 7) == 0 && sp.tw >= 8;
#define SSDK_S3(DT, CS_, NCHW_)                                                                                              \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_short_kernel<DT, CS_, NCHW_>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    hipLaunchKernelGGL((conv3x3_short_kernel<DT, CS_, NCHW_>), dim3(grid), dim3(S3_THREADS), lds, stream, sp);             \
  } while (0)
#define SSDK_S3W(DT, CS_)                                                                                                   \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_short_kernel<DT, CS_, true, true>),                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    hipLaunchKernelGGL((conv3x3_short_kernel<DT, CS_, true, true>), dim3(grid), dim3(S3_THREADS), lds, stream, sp);        \
  } while (0)
#define SSDK_S3A(DT, CS_)                      \
  do {                                         \
    if (nchw && wide) SSDK_S3W(DT, CS_);       \
    else if (nchw) SSDK_S3(DT, CS_, true);     \
    else SSDK_S3(DT, CS_, false);              \
  } while (0)
#define SSDK_S3C(DT)                     \
  do {                                   \
    if (cs == 1) SSDK_S3A(DT, 1);        \
    else if (cs == 2) SSDK_S3A(DT, 2);   \
    else if (cs == 3) SSDK_S3A(DT, 3);   \
    else if (cs == 4) SSDK_S3A(DT, 4);   \
    else SSDK_S3A(DT, 8);                \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_S3C(SSDK_BF16);
  else SSDK_S3C(SSDK_F16);
#undef SSDK_S3C
#undef SSDK_S3A
#undef SSDK_S3W
#undef SSDK_S3
  return 0;
}

}  // namespace ssdk
