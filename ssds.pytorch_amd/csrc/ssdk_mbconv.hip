// ssdk_mbconv.hip -- one MobileNetV2 inverted-residual block as ONE kernel on gfx950.
//
// Reference: the torchvision InvertedResidual blocks the reference's backbone is built from
// (ssds/modeling/nets/mobilenet.py:56, 84-89): 1x1 expand (BN, ReLU6) -> 3x3 depthwise stride s (BN, ReLU6)
// -> 1x1 linear projection (BN) [+ x].  Run layer by layer, the expanded tensor (6x the block input) is
// written to and read from HBM twice and dominates the network's traffic (0.8 GB per layer at 256x256,
// batch 64).  Here a workgroup owns an 8x8 output tile of one image and keeps everything on chip:
//
//   sX   the input tile with its halo ((8s+2)^2 pixels x Cin) in LDS, loaded once
//   per 32-channel chunk of the hidden dimension:
//     P1  expand:   E[p][hc]  = relu6(se * sum_ci X[p][ci] We[hc][ci] + be)     MFMA 16x16x32, -> LDS
//                   (zero outside the image: the depthwise conv pads the EXPANDED activation)
//     P2  depthwise D[q][hc]  = relu6(sd * sum_taps E[..][hc] Wd[tap][hc] + bd)  VALU fp32, -> LDS
//     P3  project   Y[q][co] += sum_hc D[q][hc] Wp[co][hc]                       MFMA, accumulators in VGPRs
//   epilogue: Y * sp + bp (+ x from sX), 8-byte stores, NHWC.
//
// Two barriers per chunk.  HBM traffic = block input (with a 1.13-1.56x halo) + block output + weights
// from L2: the roofline of the fused block is its input+output bytes.
#include <stdio.h>

#include "ssdk_common.h"

namespace ssdk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MbParams {
  const u16* x;
  u16* y;
  const u16* we;    // [Chid][Cin], activation dtype
  const float* se;  // [Chid] folded BN of the expand conv
  const float* be;
  const u16* wd;    // fp16 [3][3][Chid], BN scale folded in
  const u16* bd;    // fp16 [Chid]
  const u16* wp;    // fp16 [Cout][Chid]
  const float* sp;  // [Cout]
  const float* bp;
  int N, H, W, Cin, Chid, Cout, Ho, Wo, residual;
  int tiles_x, tiles_y;
  int xs;      // LDS row stride of sX (bytes)
  int wes;     // LDS row stride of the staged We chunk (bytes)
  int off_wp, off_wd, off_sb, wbuf;  // byte offsets inside / size of one staged weight buffer
  // stem mode: x is the [N,Cimg,Himg,Wimg] image (1 = NCHW, 2 = NHWC); the "expand" GEMM is the 3x3/s2 stem
  // conv on an im2col image of the tile built in LDS (K = 9*Cimg <= 32); H, W are the stem-output grid.
  int stem, Himg, Wimg, Cimg;
  int lean;                  // 1: the two-workgroups-per-CU instance of a 16x16 tile (see mb_lean4)
  int hc;                    // hidden channels per chunk (32 | 64)
  int ts, tsw;               // output tile rows x columns: 8x8 | 16x16 | 8x16
  unsigned long long* dbg;  // SSDK_MB_DBG=1: cycle stamps of workgroup 0 (debug builds of the schedule only)
};

template <int DT> __device__ __forceinline__ u32 mb_to16(float v) {
  if constexpr (DT == SSDK_BF16) {
    u32 b = __builtin_bit_cast(u32, v);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
  } else {
    _Float16 h = (_Float16)v;
    return (u32)__builtin_bit_cast(u16, h);
  }
}
template <int DT> __device__ __forceinline__ float mb_from16(u32 h) {
  if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
  else return f16_bits_to_f32(h);
}
template <int DT>
__device__ __forceinline__ f32x4 mb_mfma(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (DT == SSDK_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// NOTE: never __builtin_bit_cast straight from a vector-element lvalue (v[e]): clang reads element 0.
__device__ __forceinline__ h2 as_h2(u32 w) { return __builtin_bit_cast(h2, w); }
__device__ __forceinline__ u32 pk_relu6_f16(float a, float b) {  // clamp to [0,6], pack two halves
  return __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_fmed3f(a, 0.f, 6.f),
                                                            __builtin_amdgcn_fmed3f(b, 0.f, 6.f)));
}

// Instances whose registers fit 128 per lane (almost) without spilling also exist compiled for 4 waves per SIMD (LEAN4),
// i.e. TWO workgroups per CU; the host picks them wherever the LDS (<= 80 KB) allows two workgroups as well: the block kernels are latency-bound (LDS -> MFMA -> LDS chains
// between barriers), and a second workgroup fills the issue slots the first one leaves idle.  Today: the 16x16 tiles
// with one expand k-step and two projection n-frags (Cin <= 32, Cout <= 32; round 1 held 160-193 registers there:
// one workgroup per CU although the LDS allowed two).
constexpr bool mb_lean4(int ts, int tsw, int nfo, int ksmax, bool stem) {
  return ts == 16 && tsw == 16 && nfo == 2 && (ksmax == 1 || stem);
}
constexpr int kMbThreads = 512;  // 8 waves: two per SIMD, so LDS / MFMA latencies of one wave hide under the other
constexpr int kMbWaves = kMbThreads / 64;
constexpr int MAX_KS = 5;    // Cin <= 160
// HC = hidden channels per chunk (32 or 64): ES = HC halves + 16 B pad is the LDS row stride of sE / sD / the
// staged Wp (an odd number of 16-byte slots); the staged per-chunk scalars are se, be (fp32 x HC), bd (fp16 x HC).
constexpr int es_of(int hc) { return hc * 2 + 16; }
constexpr int sb_of(int hc) { return hc * 10; }

// E (expanded) and D (depthwise output) are INTERNAL tensors: they are kept in fp16 whatever the model
// dtype (values are ReLU6-bounded, fp16 carries 3 more mantissa bits than bf16), so P2 runs on packed
// fp16 math (v_pk_fma_f16: 2 channels per instruction) and P3 on the f16 MFMA with fp16 projection weights.
template <int DT, int S, int NFO, int KSMAX, bool STEM = false, bool RESIDENT = false, int HC = 32, int TS = 8, int TSW = TS,
          bool LEAN4 = false>
__global__ __launch_bounds__(kMbThreads, LEAN4 ? 4 : 2) void mbconv_kernel(const MbParams p) {
  constexpr int ES = es_of(HC);
  constexpr int NJ = HC / 16;   // n-frags of the expand GEMM per chunk
  constexpr int KP = HC / 32;   // k-steps of the projection per chunk
  constexpr int NI = HC / 32;   // depthwise items (4 channels of one pixel) per thread
  constexpr int HP = HC / 8;    // 16-byte pieces per HC halves
  // TS x TS output pixels per workgroup.  16 x 16 (stride-1 blocks on large maps) quarters the phases (barriers,
  // LDS round trips) per pixel, shrinks the halo overhead of the expand GEMM from 1.56x to 1.27x and lets the
  // depthwise phase reuse its LDS reads over 1 x 4 pixel strips.
  // TS x TSW: 8x8 (any block), 16x16 (stride 1, 32-channel chunks) or 8x16 (stride 2: twice the work per phase of 8x8)
  static_assert((TS == 8 && (TSW == 8 || (TSW == 16 && !STEM))) || (TS == 16 && TSW == 16 && S == 1 && HC == 32), "tile shape");
  constexpr int OP = TS * TSW;                  // output pixels per tile
  constexpr int RH = TS * S + (3 - S);          // input rows per tile: 10 | 18 (s=1) or 17 (s=2)
  constexpr int RW = TSW * S + (3 - S);         // input columns per tile (= row stride of the region): .. | 33 (8x16, s=2)
  constexpr int P = RH * RW;                   // region pixels
  constexpr int MF = (P + 15) / 16;            // m-frags of the expand GEMM
  constexpr int P16 = MF * 16;
  constexpr int MFW = (MF + kMbWaves - 1) / kMbWaves;                   // m-frags per wave (max)
  constexpr int NPA = (HC * KSMAX * 4 + kMbThreads - 1) / kMbThreads;   // We pieces per thread
  constexpr int NPB = (NFO * 16 * HP + kMbThreads - 1) / kMbThreads;    // Wp pieces per thread
  constexpr int NM_WD = 9 * HP, NM_S = HC / 4, NM_B = HP;               // misc pieces: Wd, se|be, bd
  constexpr int MF3 = OP / 16;                   // pixel fragments of the projection GEMM: 4 | 16
  constexpr int NSPLIT = MF3 >= kMbWaves ? 1 : kMbWaves / MF3;          // waves sharing one pixel fragment
  constexpr int MPW = MF3 >= kMbWaves ? MF3 / kMbWaves : 1;             // pixel fragments per wave
  constexpr int NFH = NFO / NSPLIT;                                     // projection n-frags per wave
  constexpr int PR = 2 * RW + 1;   // stem mode: rows / columns of the image patch behind the RW x RW stem outputs
  constexpr int PC = PR + 1;       // LDS row stride of the patch in pixels (even: 16-byte aligned pixel pairs)
  constexpr int NPX = STEM ? (PR * PR * 3 + kMbThreads - 1) / kMbThreads                  // image elements
                           : (P16 * KSMAX * 4 + kMbThreads - 1) / kMbThreads;             // 16-byte pieces of sX
  static_assert(NFO % 2 == 0, "NFO must be even");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int XS = p.xs, WES = p.wes;
  unsigned char* sX = smem;                                  // [P16][XS]
  const size_t sx_bytes = STEM ? (size_t)(((PR * PC + 8) * 8 + 15) & ~15) : (size_t)P16 * XS;  // stem: the image patch
  unsigned char* sE = sX + sx_bytes;                         // [P16][ES]
  unsigned char* sD = sE + (size_t)P16 * ES;                 // [OP][ES]
  unsigned char* sW = sD + OP * ES;                          // 2 staged weight buffers

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int Cin = p.Cin, Chid = p.Chid, Cout = p.Cout, H = p.H, W = p.W;
  const u32 ntiles = (u32)p.N * (u32)p.tiles_y * (u32)p.tiles_x;
  int n = 0, oy0 = 0, ox0 = 0, iy0 = 0, ix0 = 0;
  auto tile_coords = [&](u32 t, int& tn, int& toy0, int& tox0) {
    const int ttx = (int)(t % (u32)p.tiles_x);
    t /= (u32)p.tiles_x;
    toy0 = (int)(t % (u32)p.tiles_y) * TS;
    tox0 = ttx * TSW;
    tn = (int)(t / (u32)p.tiles_y);
  };
  const int KS = (Cin + 31) / 32;
  const int cpr = Cin / 8;  // 16-byte pieces per pixel / per We row

  // ---- weight staging: global (L2) -> registers one chunk ahead -> LDS.  The piece -> (row, column)
  //      decode is chunk-invariant and done once here. ---------------------------------------------------
  int a_src[NPA], a_dst[NPA], a_row[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int q = (int)tid + i * kMbThreads;
    const int row = q / cpr, c = q % cpr;
    a_row[i] = row < HC ? row : (1 << 30);   // invalid pieces never pass "hc0 + row < Chid"
    a_src[i] = row * Cin + c * 8;
    a_dst[i] = row * WES + c * 16;
  }
  int b_src[NPB], b_dst[NPB], b_c8[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int q = (int)tid + i * kMbThreads;
    const int row = q / HP, c = q % HP;
    b_c8[i] = (row < Cout && q < NFO * 16 * HP) ? c * 8 : (1 << 30);
    b_src[i] = row * Chid + c * 8;
    b_dst[i] = (q < NFO * 16 * HP) ? p.off_wp + row * ES + c * 16 : -1;
  }
  // tid < NM_WD + 2*NM_S + NM_B: one piece of {Wd (9 taps x HC), se, be (fp32 x HC), bd (fp16 x HC)}
  const u16* c_src = nullptr;
  int c_mul = 0, c_off = 1 << 30, c_dst = -1;
  if (tid < NM_WD) {
    const int tap = (int)tid / HP, c = (int)tid % HP;
    c_src = p.wd + (size_t)tap * Chid + c * 8;
    c_mul = 1;
    c_off = c * 8;
    c_dst = p.off_wd + tap * (HC * 2) + c * 16;
  } else if (tid < NM_WD + 2 * NM_S) {
    const int r = (int)tid - NM_WD;
    const int c = r % NM_S;
    c_src = reinterpret_cast<const u16*>((r < NM_S ? p.se : p.be) + c * 4);
    c_mul = 2;  // fp32: two u16 per element
    c_off = c * 4;
    c_dst = p.off_sb + r * 16;
  } else if (tid < NM_WD + 2 * NM_S + NM_B) {
    const int c = (int)tid - NM_WD - 2 * NM_S;
    c_src = p.bd + c * 8;
    c_mul = 1;
    c_off = c * 8;
    c_dst = p.off_sb + HC * 8 + c * 16;
  }
  u32x4 ra[NPA], rb[NPB], rc;
  auto load_w = [&](int hc0) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (hc0 + a_row[i] < Chid) v = *reinterpret_cast<const u32x4*>(p.we + (size_t)hc0 * Cin + a_src[i]);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (hc0 + b_c8[i] < Chid) v = *reinterpret_cast<const u32x4*>(p.wp + hc0 + b_src[i]);
      rb[i] = v;
    }
    {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (hc0 + c_off < Chid) v = *reinterpret_cast<const u32x4*>(c_src + (size_t)hc0 * c_mul);
      rc = v;
    }
  };
  auto store_w = [&](int buf) {
    unsigned char* w = sW + (size_t)buf * p.wbuf;
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (a_row[i] < HC) *reinterpret_cast<u32x4*>(w + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if (b_dst[i] >= 0) *reinterpret_cast<u32x4*>(w + b_dst[i]) = rb[i];
    if (c_dst >= 0) *reinterpret_cast<u32x4*>(w + c_dst) = rc;
  };

  // ---- input tile (+ halo) staging: global -> registers (fetch_x, may run one tile ahead) -> sX (put_x) --
  u32x4 xr[STEM ? 1 : NPX];
  u16 xs16[STEM ? NPX : 1];
  auto fetch_x = [&](u32 t) {
    int tn, toy0, tox0;
    tile_coords(t, tn, toy0, tox0);
    const int tiy0 = toy0 * S - 1, tix0 = tox0 * S - 1;
    if constexpr (STEM) {
      // the image patch behind the tile's RW x RW stem outputs: (channel, row, column) with columns fastest, so a
      // wave reads runs of consecutive pixels of one image row (the first version gathered 2 bytes per (pixel, tap)
      // and fetched 5.7x the image from HBM, profiles/r01_pmc_fetch_size_v4.csv)
      const int Ci = p.Cimg, Hi = p.Himg, Wi = p.Wimg;
      const u16* img = p.x + (size_t)tn * Ci * Hi * Wi;
      const size_t cstride = p.stem == 1 ? (size_t)Hi * Wi : 1, pstride = p.stem == 1 ? 1 : (size_t)Ci;
      const int py0 = 2 * tiy0 - 1, px0 = 2 * tix0 - 1;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int e = (int)tid + i * kMbThreads;
        const int ch = e / (PR * PR), rc = e % (PR * PR);
        const int iy = py0 + rc / PR, ix = px0 + rc % PR;
        u16 v = 0;
        if (ch < Ci && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi)
          v = img[((size_t)iy * Wi + ix) * pstride + ch * cstride];
        xs16[i] = v;
      }
    } else {
      const u16* xin = p.x + (size_t)tn * H * W * Cin;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int q = (int)tid + i * kMbThreads;
        const int pix = q / cpr, c = q % cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (pix < P) {
          const int iy = tiy0 + pix / RW, ix = tix0 + pix % RW;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *reinterpret_cast<const u32x4*>(xin + ((size_t)iy * W + ix) * Cin + c * 8);
        }
        xr[i] = v;
      }
    }
  };
  auto put_x = [&]() {
    if constexpr (STEM) {  // sX = the patch as [row][column (stride PC)][4 channels]; channel 3 / the pad column stay zero
      u16* patch = reinterpret_cast<u16*>(sX);
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int e = (int)tid + i * kMbThreads;
        const int ch = e / (PR * PR), rc = e % (PR * PR);
        if (ch < 3) patch[((rc / PR) * PC + rc % PR) * 4 + ch] = xs16[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int q = (int)tid + i * kMbThreads;
        const int pix = q / cpr, c = q % cpr;
        if (pix < P16) *reinterpret_cast<u32x4*>(sX + (size_t)pix * XS + c * 16) = xr[i];
      }
    }
  };

  const int nchunks = (Chid + HC - 1) / HC;
  u32 tile = blockIdx.x;
  if constexpr (STEM) {
    // (+8 pixels: the kx = 3..7 slots of the last row read past the patch; their weights are zero, the data must be finite)
    for (u32 i = tid; i < (u32)((PR * PC + 8) * 2); i += kMbThreads) reinterpret_cast<u32*>(sX)[i] = 0u;
    __syncthreads();
  }
  if constexpr (RESIDENT) {  // every chunk's weights live in LDS for the whole (persistent) workgroup
    for (int c = 0; c < nchunks; ++c) {
      load_w(c * HC);
      store_w(c);
    }
    if constexpr (!LEAN4) fetch_x(tile);
  } else {
    load_w(0);
    fetch_x(tile);
    put_x();
    store_w(0);
  }
  // P2 role: output pixel, 4-channel group.  ds_read_b64 is serviced 32 lanes at a time over 64 banks = 4 pixels x 8
  // channel groups (64 B each); a pixel row is ES = 20 | 36 banks, so four NEIGHBOURING pixels wrap onto each other
  // (2-way conflict on every tap: SQ_LDS_BANK_CONFLICT was 22 % of the 8x8 instances' time).  A half-wave therefore
  // takes every SECOND pixel -- of a column in the 8x8 tile (pitch 2 rows = 16 banks mod 64 for both strides and both
  // chunk widths), of a row in the 8x16 tile (stride 2: pitch 4 pixels = 16 banks) -- and its four 16-bank windows tile
  // the 64 banks exactly.
  const u32 d_cg = tid & 7u, d_sub = 2u * ((lane >> 3) & 3u) + (lane >> 5);
  const u32 d_oy = TSW == 8 ? d_sub : wave, d_ox = TSW == 8 ? wave : d_sub;
  const u32 d_px = TSW == 8 ? d_oy * 8u + d_ox : 8u * wave + d_sub;
  // P3 role: 8x8 tiles: pixel frag wave & 3, interleaved half (wave >> 2) of the n-frags; 16x16: frags 2*wave, 2*wave+1, all n
  const u32 m_base = NSPLIT == 2 ? (wave & 3u) : wave * (u32)MPW, n_half = NSPLIT == 2 ? (wave >> 2) : 0u;

  // projection BN: loop invariant, kept in registers for the whole workgroup -- except in the lean instances (two
  // workgroups per CU), which re-load it in the epilogue: 8 NFH registers less
  static_assert(!LEAN4 || mb_lean4(TS, TSW, NFO, KSMAX, STEM), "lean instance");
  constexpr bool LEAN = LEAN4;
  f32x4 sp4[LEAN ? 1 : NFH], bp4[LEAN ? 1 : NFH];
#pragma unroll
  for (int jj = 0; jj < (LEAN ? 0 : NFH); ++jj) {
    const int co = ((int)n_half + NSPLIT * jj) * 16 + (int)fg * 4;
    sp4[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    bp4[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (co < Cout) {
      sp4[jj] = *reinterpret_cast<const f32x4*>(p.sp + co);
      bp4[jj] = *reinterpret_cast<const f32x4*>(p.bp + co);
    }
  }
  int dbg_k = 0;
#define MB_STAMP()                                                                         \
  do {                                                                                     \
    if (p.dbg && blockIdx.x == 0 && tid == 0 && dbg_k < 60) p.dbg[dbg_k++] = __builtin_readcyclecounter(); \
  } while (0)
  for (; tile < ntiles; tile += gridDim.x) {  // one iteration unless RESIDENT (persistent grid)
  MB_STAMP();
  tile_coords(tile, n, oy0, ox0);
  iy0 = oy0 * S - 1;
  ix0 = ox0 * S - 1;
  if constexpr (RESIDENT) {
    __syncthreads();  // previous tile: every wave is done with sX (residual reads) and sE/sD
    if constexpr (LEAN4) {
      fetch_x(tile);  // lean instances: no register-held prefetch; the co-resident workgroup covers the latency
      put_x();
    } else {
      put_x();
      if (tile + gridDim.x < ntiles) fetch_x(tile + gridDim.x);  // next tile's input in flight under this tile
    }
  }
  // validity of the region pixels this lane produces in P1 (bit i: m-frag wave + 8*i)
  u32 pvalid = 0;
#pragma unroll
  for (int i = 0; i < MFW; ++i) {
    const int pix = ((int)wave + kMbWaves * i) * 16 + (int)fr;
    if (pix < P) {
      const int iy = iy0 + pix / RW, ix = ix0 + pix % RW;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) pvalid |= 1u << i;
    }
  }
  f32x4 yacc[MPW][NFH];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NFH; ++j) yacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // The expand GEMM's B operand (this wave's pixel fragments of the input tile) does not change from chunk to chunk: it is
  // read from LDS ONCE per tile into registers (round 4).  The chunk loop's P1 was LDS-bandwidth bound -- per 32-channel
  // chunk of a 16x16 tile it re-read the 42 KB of x fragments next to 32 KB of weight fragments and 21 KB of E writes.
  // (only where the registers are there: the 16x16 instances up to Cin = 96 / Cout = 96 -- the 96-channel one holds 256 VGPRs and
  //  48 B of scratch with it and is still 1-3 us faster --, the long-K 8x8 instances that run one workgroup per CU anyway; the
  //  others spill more or lose their second workgroup per CU)
  constexpr bool XREG = !STEM && !LEAN4 && MFW * KSMAX <= 10 && NFO <= 10 &&
                        ((TS == 16 && NFO * KSMAX <= 18) || (TS == 8 && TSW == 8 && KSMAX >= 3 && !(S == 2 && NFO >= 10 && HC == 64)));
  u32x4 xreg[XREG ? MFW : 1][XREG ? KSMAX : 1];
  if constexpr (XREG) {
#pragma unroll
    for (int i = 0; i < MFW; ++i) {
      int spix = ((int)wave + kMbWaves * i) * 16 + (int)fr;
      if (spix >= P) spix = 0;  // also covers mf >= MF: the result is not stored
      const unsigned char* xrow = sX + (size_t)spix * XS;
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        const int k = ks * 32 + (int)fg * 8;
        xreg[i][ks] = u32x4{0u, 0u, 0u, 0u};
        if (ks < KS && k < Cin) xreg[i][ks] = *reinterpret_cast<const u32x4*>(xrow + k * 2);
      }
    }
  }

  MB_STAMP();
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* wcur = sW + (size_t)(RESIDENT ? c : (c & 1)) * p.wbuf;
    if constexpr (!RESIDENT)
      if (c + 1 < nchunks) load_w((c + 1) * HC);  // next chunk's weights in flight under this chunk's work
    // ---- P1: expand the region for channels [hc0, hc0+32) -> sE (fp16) -----------------------------
    {
      f32x4 sev[NJ], bev[NJ];
#pragma unroll
      for (int jf = 0; jf < NJ; ++jf) {
        sev[jf] = *reinterpret_cast<const f32x4*>(wcur + p.off_sb + (jf * 16 + fg * 4) * 4);
        bev[jf] = *reinterpret_cast<const f32x4*>(wcur + p.off_sb + HC * 4 + (jf * 16 + fg * 4) * 4);
      }
      if constexpr ((TS == 16 || TSW == 16) && !LEAN4) {  // (measured: the same restructuring LOSES on the 8x8 stride-2 instances -- VGPRs / occupancy)
        // 16x16 tiles: 3 m-frags per wave.  Weight fragments are shared by them and loaded once; all operand reads
        // are issued before the first MFMA and all results are written after the last one, so the three dependent
        // LDS -> MFMA -> LDS chains overlap instead of running back to back.
        u32x4 wv[KSMAX][NJ], xf[MFW][KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
          const int k = ks * 32 + (int)fg * 8;
#pragma unroll
          for (int jf = 0; jf < NJ; ++jf) {
            wv[ks][jf] = u32x4{0u, 0u, 0u, 0u};
            if (ks < KS && k < Cin) wv[ks][jf] = *reinterpret_cast<const u32x4*>(wcur + (size_t)(jf * 16 + fr) * WES + k * 2);
          }
        }
#pragma unroll
        for (int i = 0; i < MFW; ++i) {
          const int mf = (int)wave + kMbWaves * i;
          int spix = mf * 16 + (int)fr;
          if (spix >= P) spix = 0;  // also covers mf >= MF: the result is not stored
          const unsigned char* xrow = sX + (size_t)spix * XS;
          const unsigned char* prow = sX + (size_t)(((spix / RW) * 2) * PC + (spix % RW) * 2 + 2 * (int)fg) * 8;
#pragma unroll
          for (int ks = 0; ks < KSMAX; ++ks) {
            const int k = ks * 32 + (int)fg * 8;
            xf[i][ks] = u32x4{0u, 0u, 0u, 0u};
            if (ks < KS) {
              if constexpr (STEM) xf[i][ks] = *reinterpret_cast<const u32x4*>(prow + (size_t)ks * PC * 8);
              else if constexpr (XREG) xf[i][ks] = xreg[i][ks];
              else if (k < Cin) xf[i][ks] = *reinterpret_cast<const u32x4*>(xrow + k * 2);
            }
          }
        }
        f32x4 e[MFW][NJ];
#pragma unroll
        for (int i = 0; i < MFW; ++i)
#pragma unroll
          for (int jf = 0; jf < NJ; ++jf) {
            e[i][jf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks)
              if (ks < KS) e[i][jf] = mb_mfma<DT>(wv[ks][jf], xf[i][ks], e[i][jf]);  // D[hc = fg*4+r][pixel = fr]
          }
#pragma unroll
        for (int i = 0; i < MFW; ++i) {
          const int mf = (int)wave + kMbWaves * i;
          if (mf < MF) {
            const bool ok = (pvalid >> i) & 1u;
            unsigned char* erow = sE + (size_t)(mf * 16 + (int)fr) * ES;
#pragma unroll
            for (int jf = 0; jf < NJ; ++jf) {
              uint2 o = make_uint2(0u, 0u);
              if (ok) {
                o.x = pk_relu6_f16(fmaf(e[i][jf][0], sev[jf][0], bev[jf][0]), fmaf(e[i][jf][1], sev[jf][1], bev[jf][1]));
                o.y = pk_relu6_f16(fmaf(e[i][jf][2], sev[jf][2], bev[jf][2]), fmaf(e[i][jf][3], sev[jf][3], bev[jf][3]));
              }
              *reinterpret_cast<uint2*>(erow + (jf * 16 + fg * 4) * 2) = o;
            }
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < MFW; ++i) {
        const int mf = (int)wave + kMbWaves * i;
        if (mf < MF) {  // wave-uniform
          f32x4 e[NJ];
#pragma unroll
          for (int jf = 0; jf < NJ; ++jf) e[jf] = f32x4{0.f, 0.f, 0.f, 0.f};
          const unsigned char* xrow = sX + (size_t)(mf * 16 + (int)fr) * XS;
          int spix = mf * 16 + (int)fr;  // stem mode: k-step = kernel row, a lane's 8 k-values = 2 patch pixels x 4 ch
          if (spix >= P) spix = 0;
          const unsigned char* prow = sX + (size_t)(((spix / RW) * 2) * PC + (spix % RW) * 2 + 2 * (int)fg) * 8;
#pragma unroll
          for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {  // the k-step's operand reads first, then its MFMAs (one exposed LDS latency per k-step, not 1 + NJ)
              const int k = ks * 32 + (int)fg * 8;
              u32x4 xf = {0u, 0u, 0u, 0u}, wv[NJ];
#pragma unroll
              for (int jf = 0; jf < NJ; ++jf) wv[jf] = u32x4{0u, 0u, 0u, 0u};
              if constexpr (STEM) xf = *reinterpret_cast<const u32x4*>(prow + (size_t)ks * PC * 8);
              if (k < Cin) {
                if constexpr (XREG) xf = xreg[i][ks];
                else if constexpr (!STEM) xf = *reinterpret_cast<const u32x4*>(xrow + k * 2);
#pragma unroll
                for (int jf = 0; jf < NJ; ++jf) wv[jf] = *reinterpret_cast<const u32x4*>(wcur + (size_t)(jf * 16 + fr) * WES + k * 2);
              }
#pragma unroll
              for (int jf = 0; jf < NJ; ++jf) e[jf] = mb_mfma<DT>(wv[jf], xf, e[jf]);  // D[hc = fg*4+r][pixel = fr]
            }
          }
          const bool ok = (pvalid >> i) & 1u;
          unsigned char* erow = sE + (size_t)(mf * 16 + (int)fr) * ES;
#pragma unroll
          for (int jf = 0; jf < NJ; ++jf) {
            uint2 o = make_uint2(0u, 0u);
            if (ok) {
              o.x = pk_relu6_f16(fmaf(e[jf][0], sev[jf][0], bev[jf][0]), fmaf(e[jf][1], sev[jf][1], bev[jf][1]));
              o.y = pk_relu6_f16(fmaf(e[jf][2], sev[jf][2], bev[jf][2]), fmaf(e[jf][3], sev[jf][3], bev[jf][3]));
            }
            *reinterpret_cast<uint2*>(erow + (jf * 16 + fg * 4) * 2) = o;
          }
        }
      }
      }
    }
    MB_STAMP();
    __syncthreads();
    MB_STAMP();
    // ---- P2: depthwise 3x3 stride S on the chunk, packed fp16 (4 channels per lane) -> sD ----------------
    if constexpr (TS == 8 && TSW == 8) {
      const h2 zero = {(_Float16)0.f, (_Float16)0.f}, six = {(_Float16)6.f, (_Float16)6.f};
      h2 acc0[NI], acc1[NI];
#pragma unroll
      for (int it = 0; it < NI; ++it) acc0[it] = acc1[it] = zero;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int rp = ((int)d_oy * S + ky) * RW + (int)d_ox * S + kx;
#pragma unroll
          for (int it = 0; it < NI; ++it) {  // independent chains: channel groups d_cg and d_cg + 8
            const u32 cg = d_cg + 8u * it;
            const uint2 ev = *reinterpret_cast<const uint2*>(sE + (size_t)rp * ES + cg * 8);
            const uint2 wv = *reinterpret_cast<const uint2*>(wcur + p.off_wd + (ky * 3 + kx) * (HC * 2) + cg * 8);
            acc0[it] = __builtin_elementwise_fma(as_h2(ev.x), as_h2(wv.x), acc0[it]);
            acc1[it] = __builtin_elementwise_fma(as_h2(ev.y), as_h2(wv.y), acc1[it]);
          }
        }
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const u32 cg = d_cg + 8u * it;
        const uint2 bv = *reinterpret_cast<const uint2*>(wcur + p.off_sb + HC * 8 + cg * 8);
        const h2 v0 = __builtin_elementwise_min(__builtin_elementwise_max(acc0[it] + as_h2(bv.x), zero), six);
        const h2 v1 = __builtin_elementwise_min(__builtin_elementwise_max(acc1[it] + as_h2(bv.y), zero), six);
        *reinterpret_cast<uint2*>(sD + (size_t)d_px * ES + cg * 8) =
            make_uint2(__builtin_bit_cast(u32, v0), __builtin_bit_cast(u32, v1));
      }
    } else if constexpr (TS == 8) {  // 8x16 tile: the 8x8 mapping, two pixel groups of 64 per thread
      static_assert(HC == 32, "8x16 tiles: 32-channel chunks");
      const h2 zero = {(_Float16)0.f, (_Float16)0.f}, six = {(_Float16)6.f, (_Float16)6.f};
#pragma unroll
      for (int pp = 0; pp < OP / 64; ++pp) {
        const u32 px = d_px + 64u * pp;
        const u32 poy = px / (u32)TSW, pox = px % (u32)TSW;
        h2 acc0 = zero, acc1 = zero;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int rp = ((int)poy * S + ky) * RW + (int)pox * S + kx;
            const uint2 ev = *reinterpret_cast<const uint2*>(sE + (size_t)rp * ES + d_cg * 8);
            const uint2 wv = *reinterpret_cast<const uint2*>(wcur + p.off_wd + (ky * 3 + kx) * (HC * 2) + d_cg * 8);
            acc0 = __builtin_elementwise_fma(as_h2(ev.x), as_h2(wv.x), acc0);
            acc1 = __builtin_elementwise_fma(as_h2(ev.y), as_h2(wv.y), acc1);
          }
        const uint2 bv = *reinterpret_cast<const uint2*>(wcur + p.off_sb + HC * 8 + d_cg * 8);
        const h2 v0 = __builtin_elementwise_min(__builtin_elementwise_max(acc0 + as_h2(bv.x), zero), six);
        const h2 v1 = __builtin_elementwise_min(__builtin_elementwise_max(acc1 + as_h2(bv.y), zero), six);
        *reinterpret_cast<uint2*>(sD + (size_t)px * ES + d_cg * 8) =
            make_uint2(__builtin_bit_cast(u32, v0), __builtin_bit_cast(u32, v1));
      }
    } else {
      // 16x16 tile: a lane owns a 1 x 4 strip of pixels x 4 channels; each of the 3 input rows is 6 LDS reads that
      // feed 12 taps (the 8x8 mapping reads 9 vectors per pixel): 2.7x less LDS traffic in the phase that is
      // closest to its LDS-bandwidth bound
      const h2 zero = {(_Float16)0.f, (_Float16)0.f}, six = {(_Float16)6.f, (_Float16)6.f};
      const u32 strip = tid >> 3, cg = tid & 7u;
      const u32 srow = strip >> 2, scol = (strip & 3u) * 4u;
      h2 a0[4], a1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a0[j] = a1[j] = zero;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const unsigned char* erow = sE + (size_t)((srow + ky) * RW + scol) * ES + cg * 8;
        uint2 ev[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) ev[cc] = *reinterpret_cast<const uint2*>(erow + (size_t)cc * ES);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const uint2 wv = *reinterpret_cast<const uint2*>(wcur + p.off_wd + (ky * 3 + kx) * (HC * 2) + cg * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a0[j] = __builtin_elementwise_fma(as_h2(ev[j + kx].x), as_h2(wv.x), a0[j]);
            a1[j] = __builtin_elementwise_fma(as_h2(ev[j + kx].y), as_h2(wv.y), a1[j]);
          }
        }
      }
      const uint2 bv = *reinterpret_cast<const uint2*>(wcur + p.off_sb + HC * 8 + cg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const h2 v0 = __builtin_elementwise_min(__builtin_elementwise_max(a0[j] + as_h2(bv.x), zero), six);
        const h2 v1 = __builtin_elementwise_min(__builtin_elementwise_max(a1[j] + as_h2(bv.y), zero), six);
        *reinterpret_cast<uint2*>(sD + (size_t)(srow * TS + scol + j) * ES + cg * 8) =
            make_uint2(__builtin_bit_cast(u32, v0), __builtin_bit_cast(u32, v1));
      }
    }
    if constexpr (!RESIDENT)
      if (c + 1 < nchunks) store_w((c + 1) & 1);  // visible to the next chunk's P1 after the barrier below
    MB_STAMP();
    __syncthreads();
    MB_STAMP();
    // ---- P3: project: wave (m_fr, n_half) owns pixels [16 m_fr, +16) x n-frags n_half, n_half+2, ... ----
    // Operand reads go out in batches ahead of the MFMAs that consume them (written read -> MFMA -> read -> MFMA, every
    // MFMA waited for its own ds_read: ~10 exposed LDS latencies per chunk); a weight fragment serves both pixel
    // fragments of the wave.
    {
      u32x4 df[MPW][KP];
#pragma unroll
      for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
          df[mi][kp] = *reinterpret_cast<const u32x4*>(sD + (size_t)((m_base + mi) * 16 + fr) * ES + kp * 64 + fg * 16);
      constexpr int WB = NFO >= 20 ? 4 : 8;                  // weight fragments in flight (x4 registers)
      constexpr int JB = (WB / KP) < 1 ? 1 : (WB / KP);     // n-frags per batch
#pragma unroll
      for (int j0 = 0; j0 < NFH; j0 += JB) {
        u32x4 wf[JB][KP];
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
          for (int kp = 0; kp < KP; ++kp)
            if (j0 + jb < NFH) {
              const int j = (int)n_half + NSPLIT * (j0 + jb);
              wf[jb][kp] = *reinterpret_cast<const u32x4*>(wcur + p.off_wp + (size_t)(j * 16 + fr) * ES + kp * 64 + fg * 16);
            }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink every read next to its MFMA again)
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
          for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi)
              if (j0 + jb < NFH)
                yacc[mi][j0 + jb] = mb_mfma<SSDK_F16>(wf[jb][kp], df[mi][kp], yacc[mi][j0 + jb]);  // D[co = fg*4+r][px = fr]
      }
    }
    // No barrier here.  The next chunk's P1 writes sE (P2 of this chunk finished reading it before the
    // barrier above) and reads the OTHER weight buffer; its P2 writes sD and its store_w overwrites THIS
    // weight buffer only after the barrier that follows its P1, i.e. after every wave has passed this P3.
  }

  MB_STAMP();
  // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
  for (int mi = 0; mi < MPW; ++mi) {
    const int q = (int)(m_base + mi) * 16 + (int)fr;  // output pixel inside the tile
    const int oy = oy0 + q / TSW, ox = ox0 + q % TSW;
    if (oy < p.Ho && ox < p.Wo) {
      u16* yrow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * Cout;
      const unsigned char* xres = sX + (size_t)(((q / TSW) * S + 1) * RW + (q % TSW) * S + 1) * XS;
#pragma unroll
      for (int jj = 0; jj < NFH; ++jj) {
        const int co = ((int)n_half + NSPLIT * jj) * 16 + (int)fg * 4;
        if (co < Cout) {  // Cout is a multiple of 8, so 4-channel groups are all-or-nothing
          u32 h[4];
          f32x4 spv, bpv;
          if constexpr (LEAN) {
            spv = *reinterpret_cast<const f32x4*>(p.sp + co);
            bpv = *reinterpret_cast<const f32x4*>(p.bp + co);
          } else {
            spv = sp4[jj];
            bpv = bp4[jj];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = mb_to16<DT>(fmaf(yacc[mi][jj][r], spv[r], bpv[r]));
          if (p.residual) {
            const uint2 xv = *reinterpret_cast<const uint2*>(xres + co * 2);
            const u32 xr[4] = {xv.x & 0xffffu, xv.x >> 16, xv.y & 0xffffu, xv.y >> 16};
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = mb_to16<DT>(mb_from16<DT>(h[r]) + mb_from16<DT>(xr[r]));
          }
          *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        }
      }
    }
  }
  MB_STAMP();
  }  // tile loop
#undef MB_STAMP
}

template <int DT, int S, int NFO, int KSMAX, bool STEM = false, bool RESIDENT = false>
static void launch_one(const MbParams& p, size_t lds, unsigned grid, hipStream_t stream) {
  if constexpr (S == 1 && KSMAX <= 3 && NFO <= 6) {  // the instantiations that exist with 16x16 tiles
    if (p.ts == 16) {
      if constexpr (mb_lean4(16, 16, NFO, KSMAX, STEM)) {
        if (p.lean) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 16, 16, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 16, 16, true>), dim3(grid), dim3(kMbThreads), lds,
                             stream, p);
          return;
        }
      }
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 16>), dim3(grid), dim3(kMbThreads), lds, stream, p);
      return;
    }
  }
  if constexpr (S == 2 && !STEM && KSMAX <= 1 && NFO <= 4) {  // the instantiations that exist with 8x16 tiles
    if (p.ts == 8 && p.tsw == 16) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 8, 16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32, 8, 16>), dim3(grid), dim3(kMbThreads), lds, stream, p);
      return;
    }
  }
  if (p.hc == 64) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 64>), dim3(grid), dim3(kMbThreads), lds, stream, p);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX, STEM, RESIDENT, 32>), dim3(grid), dim3(kMbThreads), lds, stream, p);
  }
}

// (k-steps of the expand GEMM, n-frags of the projection) pairs of the MobileNetV2 family get their own
// instantiation (register budget = occupancy); everything else runs on the most general one.
template <int DT, int S>
static int launch_mb(const MbParams& p, int ks, int nfo, size_t lds, unsigned grid, hipStream_t stream, bool resident) {
  if (p.stem) {
    if (resident) {
      if (nfo <= 2) launch_one<DT, S, 2, 3, true, true>(p, lds, grid, stream);
      else launch_one<DT, S, 4, 3, true, true>(p, lds, grid, stream);
    } else {
      if (nfo <= 2) launch_one<DT, S, 2, 3, true>(p, lds, grid, stream);
      else launch_one<DT, S, 4, 3, true>(p, lds, grid, stream);
    }
    return check_launch(p.ts == 16 ? (p.lean ? "mbconv_kernel(stem, 16x16, 2 per CU)" : "mbconv_kernel(stem, 16x16)") : "mbconv_kernel(stem)");
  }
  if (resident && ks <= 1 && nfo <= 4) {  // small-channel, high-resolution blocks: persistent grid, resident weights
    if (nfo <= 2) launch_one<DT, S, 2, 1, false, true>(p, lds, grid, stream);
    else launch_one<DT, S, 4, 1, false, true>(p, lds, grid, stream);
    return check_launch(p.ts == 16 ? (p.lean ? "mbconv_kernel(resident, 16x16, 2 per CU)" : "mbconv_kernel(resident, 16x16)")
                                   : p.tsw == 16 ? "mbconv_kernel(resident, 8x16)" : "mbconv_kernel(resident)");
  }
  if (ks <= 1 && nfo <= 2) launch_one<DT, S, 2, 1>(p, lds, grid, stream);
  else if (ks <= 1 && nfo <= 4) launch_one<DT, S, 4, 1>(p, lds, grid, stream);
  else if (ks <= 2 && nfo <= 4) launch_one<DT, S, 4, 2>(p, lds, grid, stream);
  else if (ks <= 2 && nfo <= 6) launch_one<DT, S, 6, 2>(p, lds, grid, stream);
  else if (ks <= 3 && nfo <= 6) launch_one<DT, S, 6, 3>(p, lds, grid, stream);
  else if (ks <= 3 && nfo <= 10) launch_one<DT, S, 10, 3>(p, lds, grid, stream);
  else if (nfo <= 10) launch_one<DT, S, 10, 5>(p, lds, grid, stream);
  else launch_one<DT, S, 20, 5>(p, lds, grid, stream);
  return check_launch(p.ts == 16 ? (p.lean ? "mbconv_kernel(16x16, 2 per CU)" : "mbconv_kernel(16x16)") : p.tsw == 16 ? "mbconv_kernel(8x16)" : "mbconv_kernel");
}

}  // namespace ssdk

namespace ssdk {
int launch_mbflow(const ssdk_mbconv_desc* d, hipStream_t stream);  // ssdk_mbflow.hip: 0 = launched, 1 = not one of its blocks
int launch_mbsplit(const ssdk_mbconv_desc* d, hipStream_t stream);  // ssdk_mbsplit.hip: same contract
int launch_mbk(const ssdk_mbconv_desc* d, hipStream_t stream);      // ssdk_mbk.hip: same contract
}
using namespace ssdk;

extern "C" int ssdk_mbconv(const ssdk_mbconv_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ssdk::lds_poison(stream);
  if (!d || !d->x || !d->y || !d->w_expand || !d->w_dw || !d->w_project || !d->scale_expand || !d->bias_expand ||
      !d->bias_dw || !d->scale_project || !d->bias_project) {
    set_error("mbconv: null pointer");
    return SSDK_E_BADARG;
  }
  if (d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) {
    set_error("mbconv: dtype must be bf16 or f16");
    return SSDK_E_BADARG;
  }
  const int stem = d->stem;
  if (stem && (stem < 1 || stem > 2 || d->Cin < 1 || 9 * d->Cin > 32 || d->residual || d->Cout > 64)) {
    set_error("mbconv(stem): needs 9*Cin <= 32, no residual, Cout <= 64 (Cin=%d Cout=%d)", d->Cin, d->Cout);
    return SSDK_E_BADARG;
  }
  if (d->N < 1 || d->H < 1 || d->W < 1 || (d->stride != 1 && d->stride != 2) ||
      (!stem && (d->Cin < 8 || (d->Cin % 8) || d->Cin > 32 * MAX_KS)) || (d->Chid % 8) || d->Chid < 8 ||
      (d->Cout % 8) || d->Cout < 8 || d->Cout > 320 || (d->residual && (d->stride != 1 || d->Cin != d->Cout))) {
    set_error("mbconv: unsupported geometry Cin=%d Chid=%d Cout=%d stride=%d residual=%d (Cin<=160, Cout<=320, "
              "channels %% 8 == 0)", d->Cin, d->Chid, d->Cout, d->stride, d->residual);
    return SSDK_E_BADARG;
  }
  if (launch_mbflow(d, stream) == 0) return check_launch(stem ? "mbflow_kernel(stem)" : "mbflow_kernel");  // high-resolution blocks: ssdk_mbflow.hip
  // the row-pair kernel first: it also takes the 32 -> 192 -> 32 @64x64 blocks ssdk_mbsplit.hip has an instance for (SSDK_MBK_FIRST=0: A/B)
  constexpr int env_mbk_first = 1;  // (round 6: the SSDK_MBK_FIRST switch is gone, its A/B is settled)  // (measured: 52 -> 39 us per block)
  if (env_mbk_first && launch_mbk(d, stream) == 0) return check_launch("mbk_kernel");
  if (launch_mbsplit(d, stream) == 0) return check_launch("mbsplit_kernel");  // mid-resolution blocks: hidden channels split over the waves
  if (launch_mbk(d, stream) == 0) return check_launch("mbk_kernel");  // 16- to 64-pixel-wide maps: row pairs, weights from L2 into MFMA operands
  MbParams p;
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
  p.stem = stem;
  p.Himg = d->H;
  p.Wimg = d->W;
  p.Cimg = d->Cin;
  if (stem) {  // the block's grid is the stem conv's output (3x3, stride 2, pad 1); K = (ky, kx padded to 8, ci padded to 4)
    p.H = (d->H + 2 - 3) / 2 + 1;
    p.W = (d->W + 2 - 3) / 2 + 1;
    p.Cin = 96;
  }
  p.Chid = d->Chid;
  p.Cout = d->Cout;
  p.Ho = (p.H + 2 - 3) / d->stride + 1;
  p.Wo = (p.W + 2 - 3) / d->stride + 1;
  p.residual = d->residual;
  // 16x16 output tiles for stride-1 blocks with few channels on maps large enough to still fill the chip
  constexpr int env_ts = 0;  // (round 6: the SSDK_MB_TS switch is gone, its A/B is settled)
  {
    const int nfo16 = (d->Cout + 15) / 16, ks16 = ((stem ? 96 : d->Cin) + 31) / 32;
    const long tiles16 = (long)d->N * ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16);
    // measured on SSD-MobileNetV2@512 batch 64: the 16x16 tile wins down to ONE workgroup per CU (256 tiles: the
    // 64-channel blocks at 32x32 go 55 -> 35 us), i.e. work per phase matters more than co-resident workgroups
    constexpr int env_min = 256;  // (round 6: the SSDK_MB_TS16_MIN switch is gone, its A/B is settled)
    constexpr int env_ks = 3;  // (round 6: the SSDK_MB_TS16_KS switch is gone, its A/B is settled)
    constexpr int env_nfo = 6;  // (round 6: the SSDK_MB_TS16_NFO switch is gone, its A/B is settled)
    const bool can16 = d->stride == 1 && nfo16 <= env_nfo && (stem || ks16 <= env_ks);
    p.ts = (can16 && env_ts != 8 && (tiles16 >= env_min || env_ts == 16)) ? 16 : 8;
  }
  p.tsw = p.ts;
  {  // 8x16 tiles for stride-2 blocks with <= 32 input channels and a long chunk loop (>= 6 chunks): twice the work per
     // phase of 8x8 but one workgroup per CU instead of two -- measured 47 -> 41 us at 192 hidden channels, a LOSS at 96 / 144
    constexpr int env_816 = 1;  // (round 6: the SSDK_MB_8X16 switch is gone, its A/B is settled)
    const long tiles816 = (long)d->N * ((p.Wo + 15) / 16) * ((p.Ho + 7) / 8);
    if (env_816 && p.ts == 8 && env_ts == 0 && d->stride == 2 && !stem && d->Cin <= 32 && (d->Cout + 15) / 16 <= 4 &&
        ((tiles816 >= 256 && d->Chid >= 192) || env_816 == 2))
      p.tsw = 16;
  }
  p.tiles_x = (p.Wo + p.tsw - 1) / p.tsw;
  p.tiles_y = (p.Ho + p.ts - 1) / p.ts;
  const int cpr = p.Cin / 8;
  p.xs = p.Cin * 2 + ((cpr % 2 == 0) ? 16 : 0);  // odd number of 16-byte slots per row
  const int rw = p.ts * d->stride + (3 - d->stride);  // region rows (stem: square)
  const int rww = p.tsw * d->stride + (3 - d->stride);
  const int p16 = ((rw * rww + 15) / 16) * 16;
  const int nfo_t = (d->Cout + 15) / 16;
  const int nfo_inst = nfo_t <= 2 ? 2 : nfo_t <= 4 ? 4 : nfo_t <= 6 ? 6 : nfo_t <= 10 ? 10 : 20;
  p.wes = p.xs;
  const int ks_t = (p.Cin + 31) / 32;
  constexpr int env_res = 1;  // (round 6: the SSDK_MB_RESIDENT switch is gone, its A/B is settled)
  constexpr int env_hc = 0;  // (round 6: the SSDK_MB_HC switch is gone, its A/B is settled)
  const long tiles_total = (long)d->N * p.tiles_x * p.tiles_y;
  bool resident = false;
  auto layout = [&](int hc) -> size_t {  // fills the staged-buffer offsets for `hc`, returns the LDS bytes
    const int es = es_of(hc);
    p.hc = hc;
    p.off_wp = hc * p.wes;
    p.off_wd = p.off_wp + nfo_inst * 16 * es;
    p.off_sb = p.off_wd + 9 * hc * 2;
    p.wbuf = (p.off_sb + sb_of(hc) + 15) & ~15;
    const int nch = (d->Chid + hc - 1) / hc;
    resident = env_res && (ks_t <= 1 || stem) && nfo_inst <= 4 && (size_t)nch * p.wbuf <= 56 * 1024;
    const int pr = 2 * rw + 1;
    const size_t sx = stem ? (size_t)(((pr * (pr + 1) + 8) * 8 + 15) & ~15) : (size_t)p16 * p.xs;
    return sx + (size_t)p16 * es + (size_t)(p.ts * p.tsw) * es + (resident ? (size_t)nch : 2) * (size_t)p.wbuf;
  };
  // SSDK_MB_HC = 32 | 64 forces the chunk width (A/B switch); 0 = the per-block rule below.
  size_t lds;
  {
    (void)tiles_total;
    const size_t l64 = layout(64);
    // 64-channel chunks halve the phases per hidden channel but cost LDS (occupancy): measured per block on
    // SSD-MobileNetV2@512 they win from Cin = 96 on (long chunk loops, one workgroup per CU anyway) and lose below
    const bool want64 = env_hc == 64 || (env_hc == 0 && !stem && p.Cin >= 96);
    const int hc = (want64 && p.ts == 8 && p.tsw == 8 && l64 <= 160 * 1024) ? 64 : 32;  // (16x16 tiles exist with 32 only)
    lds = layout(hc);
    if (p.ts == 16 && resident && lds > 80 * 1024) {  // two workgroups per CU beat resident weights
      const int nch = (d->Chid + hc - 1) / hc;
      const size_t l2 = lds - (size_t)nch * p.wbuf + 2 * (size_t)p.wbuf;
      if (l2 <= 80 * 1024) {
        resident = false;
        lds = l2;
      }
    }
  }
  if (lds > 160 * 1024) {
    set_error("mbconv: tile needs %zu bytes of LDS", lds);
    return SSDK_E_BADARG;
  }
  {  // two workgroups per CU need both halves of the CU: LDS here, registers through the LEAN4 instance
    constexpr int env_lean = 1;  // (round 6: the SSDK_MB_LEAN switch is gone, its A/B is settled)
    const long tiles = (long)d->N * p.tiles_x * p.tiles_y;
    p.lean = (env_lean && p.ts == 16 && p.tsw == 16 && lds <= 80 * 1024 && tiles >= 512) ? 1 : 0;
  }
  p.dbg = nullptr;
  static const int env_dbg = getenv("SSDK_MB_DBG") ? atoi(getenv("SSDK_MB_DBG")) : 0;
  static unsigned long long* dbg_dev = nullptr;
  if (env_dbg) {
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 64 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg_dev, 0, 64 * sizeof(unsigned long long), stream);
    p.dbg = dbg_dev;
  }
  unsigned grid = (unsigned)((long)d->N * p.tiles_x * p.tiles_y);
  if (resident) {  // persistent: as many workgroups as stay resident (LDS- and wave-limited: <= 4 x 8 waves per CU)
    static int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      int dev = 0;
      (void)hipGetDevice(&dev);
      cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    unsigned per_cu = (unsigned)((160 * 1024) / lds);
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    const unsigned cap = (unsigned)cus * per_cu;
    if (grid > cap) grid = cap;
  }
  const int nfo = (d->Cout + 15) / 16, ks = (p.Cin + 31) / 32;
  int rc;
  if (d->dtype == SSDK_BF16)
    rc = d->stride == 1 ? launch_mb<SSDK_BF16, 1>(p, ks, nfo, lds, grid, stream, resident)
                        : launch_mb<SSDK_BF16, 2>(p, ks, nfo, lds, grid, stream, resident);
  else
    rc = d->stride == 1 ? launch_mb<SSDK_F16, 1>(p, ks, nfo, lds, grid, stream, resident)
                        : launch_mb<SSDK_F16, 2>(p, ks, nfo, lds, grid, stream, resident);
  if (env_dbg && rc == 0) {  // debug only: synchronises
    unsigned long long h[64];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[mbconv dbg] Cin=%d Chid=%d Cout=%d s=%d stem=%d res=%d ts=%dx%d tiles=%dx%d grid=%u lds=%zu :", d->Cin,
            d->Chid, d->Cout, d->stride, p.stem, (int)resident, p.ts, p.tsw, p.tiles_x, p.tiles_y, grid, lds);
    for (int i = 1; i < 60 && h[i]; ++i) fprintf(stderr, " %llu", h[i] - h[i - 1]);
    fprintf(stderr, "\n");
  }
  return rc;
}
