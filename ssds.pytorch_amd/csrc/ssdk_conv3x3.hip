// ssdk_conv3x3.hip -- 3x3 / stride-1 convolution on the matrix cores with an LDS-resident halo tile (gfx950).
//
// The layers that carry the detector's dense FLOPs are 3x3 stride-1 convolutions: the multibox heads
// (reference ssd.py:100-103, N = A*(4+C) = 504 outputs per level) and the FPN/BiFPN shared towers
// (fpn.py:10-18, 256 -> 256 x4, then 256 -> A*4 | A*C).  As an implicit GEMM every input pixel is fetched nine
// times (once per tap); on MI355X a CU's L2 -> LDS fill path saturates near 13-16 B/clk, which caps a 256^2
// flat-K tile at ~25 % MFMA utilisation (measured, profiles/README.md).  This kernel fetches each input pixel
// ONCE per 64-channel slab:
//
//   * the output tile is a PATCH of pixels (imgs x th x tw <= 256, e.g. one 16x16 patch, or four whole 8x8 maps);
//     its (th+2) x (tw+2) halo for one 64-channel slab lives in LDS as rows of 128 bytes;
//   * the nine taps are nine k-steps that read the SAME halo image with a row shift ky*(tw+2)+kx -- the MFMA A
//     fragment of a lane is 16 bytes of the halo row of "its" pixel, so a tap is an address offset;
//   * only the weights stream per k-step: 128 output channels x 64 k = 16 KiB, a three-stage ring filled by
//     global_load_lds_dwordx4 two steps ahead; the next slab's halo is fetched piecewise behind the taps of the
//     current one.  VMEM operations are counted by hand (s_waitcnt vmcnt(N), raw s_barrier) so that loads stay in
//     flight across the per-step barrier; every step issues a fixed number of loads to make the count static.
//   * LDS images are lane-linear per load instruction; the bank swizzle (16-byte chunk ^ (row >> 1) & 7) is
//     applied on the source address side and recomputed per tap on the fragment-read side.
//
// Tile: 256 pixels x 128 channels x (9 taps x 64 k), 8 waves as 4(M) x 2(N), 4x4 accumulator fragments of
// v_mfma_f32_16x16x32 per wave.  LDS: 2 halo buffers (56 KiB each) + 3 weight stages (16 KiB) = 160 KiB.
// Epilogue as in ssdk_conv.hip: fp32 scale/bias/activation, tile transposed through LDS, 16-byte stores NHWC or
// NCHW (split loc | conf).
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int H3_THREADS = 512, H3_BN = 128, H3_BK = 64;
constexpr int H3_AROWS = 448;                    // halo rows per buffer = 7 pieces x 8 waves x 8 rows
constexpr int H3_A_BYTES = H3_AROWS * 128;       // 57344
constexpr int H3_B_BYTES = H3_BN * 128;          // 16384
constexpr int H3_NB = 3;
constexpr int H3_LDS = 2 * H3_A_BYTES + H3_NB * H3_B_BYTES;  // 163840 = all 160 KiB of the CU
constexpr int H3_NPIECE = H3_AROWS / 64;         // halo pieces (1 KiB wave instructions) per wave

struct HaloParams {
  ConvParams c;
  int imgs, th, tw;          // patch: imgs images x th x tw output pixels (imgs*th*tw <= 256)
  int tiles_y, tiles_x;      // patches per image (imgs == 1) -- 1 x 1 when imgs > 1
  int groups;                // ceil(N / imgs)
  int hrows;                 // imgs*(th+2)*(tw+2) <= H3_AROWS
  int n_tiles;               // ceil(Cout / 128)
  int vec_nchw;              // tw % 8 == 0 && Wo % 8 == 0: 16-byte NCHW stores
  unsigned x_bytes, w_bytes;  // buffer-descriptor ranges (tensors < 4 GiB)
};

#define H3_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int DT>
__global__ __launch_bounds__(H3_THREADS) void conv3x3_halo_kernel(const HaloParams hp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvParams& p = hp.c;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 wm = wave >> 1, wn = wave & 1u;

  // XCD-aware (bijective) order; n-tile fastest so that the tiles of one patch share an XCD's L2
  const u32 nwg = gridDim.x, id = blockIdx.x;
  const u32 q8 = nwg >> 3, r8 = nwg & 7u, xcd = id & 7u;
  const u32 lin = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + (id >> 3);
  const u32 nt = lin % (u32)hp.n_tiles;
  u32 pt = lin / (u32)hp.n_tiles;
  const u32 tx = pt % (u32)hp.tiles_x;
  pt /= (u32)hp.tiles_x;
  const u32 ty = pt % (u32)hp.tiles_y;
  const u32 grp = pt / (u32)hp.tiles_y;
  const int b0 = (int)grp * hp.imgs, y0 = (int)ty * hp.th, x0 = (int)tx * hp.tw;
  const u32 n0 = nt * H3_BN;

  const int Cin = p.Cin, H = p.H, W = p.W;
  const int HW2 = hp.tw + 2, HH2 = hp.th + 2;
  const int Ktot = 9 * Cin;
  const int cchunks = (Cin + H3_BK - 1) / H3_BK;

  // ---- loader roles ------------------------------------------------------------------------------------------
  // Loads are buffer_load_dwordx4 ... lds through raw buffer descriptors: address = base + soffset (uniform: slab /
  // tap) + voffset (per lane, fixed for the whole kernel), and a lane that must read zeros (outside the image,
  // channel >= Cin, row >= Cout) simply carries an out-of-range voffset -- no per-step address arithmetic at all.
  const u32 lrow = lane >> 3;
  const u32 lchunk = (lane & 7u) ^ (((lane >> 4) + 4u * (wave & 1u)) & 7u);  // logical chunk of this lane's slot
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, hp.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, hp.w_bytes, 0x00020000);
  constexpr u32 OOB = 0xfffffff0u;
  const int lci = (int)lchunk * 8;               // channel offset of this lane's chunk inside a slab
  const int tail = Cin - (cchunks - 1) * H3_BK;  // channels of the last slab (1..64)
  const u32 tailmask = lci < tail ? 0u : OOB;    // OR-ed into the offsets of the last slab: channel >= Cin -> zeros
  // halo pieces: piece t of this wave = rows (t*8 + wave)*8 + lrow
  u32 a_vo[H3_NPIECE];
#pragma unroll
  for (int t = 0; t < H3_NPIECE; ++t) {
    const int hr = (t * 8 + (int)wave) * 8 + (int)lrow;
    u32 off = OOB;
    if (hr < hp.hrows) {
      const int img = hr / (HH2 * HW2), rr = hr % (HH2 * HW2);
      const int hy = rr / HW2, hx = rr % HW2;
      const int b = b0 + img, iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (b < p.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        off = (u32)(((((long)b * H + iy) * W + ix) * Cin + lci) * 2);
    }
    a_vo[t] = off;
  }
  u32 b_vo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const u32 n = n0 + ((u32)j * 8u + wave) * 8u + lrow;
    const u32 off = n < (u32)p.Cout ? (u32)(((long)n * Ktot + lci) * 2) : OOB;
    b_vo[j] = off;
  }

  // ---- fragment roles ----------------------------------------------------------------------------------------
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int npix = hp.imgs * hp.th * hp.tw;
  u32 a_ad[9][4];  // LDS byte address (halo buffer 0, k-substep 0) of fragment i for every tap
  {
    int a_hr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ml = (int)(wm * 64u + i * 16 + fr);
      if (ml >= npix) ml = 0;  // rows past the patch compute garbage that is never stored
      const int img = ml / (hp.th * hp.tw), rr = ml % (hp.th * hp.tw);
      const int y = rr / hp.tw, x = rr % hp.tw;
      a_hr[i] = (img * HH2 + y) * HW2 + x;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32 hr = (u32)(a_hr[i] + (tap / 3) * HW2 + (tap % 3));
        a_ad[tap][i] = hr * 128u + (((fg ^ (hr >> 1)) & 7u) << 4);
      }
  }
  // weights: stage / fragment column are immediates on top of these two (k-substep 0 / 1)
  const u32 b_ad0 = 2u * H3_A_BYTES + (wn * 64u + fr) * 128u + ((fg ^ (fr >> 1)) << 4);
  const u32 b_ad1 = b_ad0 ^ 64u;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_b = [&](int stage, int cc, int tap) {  // weights of (slab cc, tap) -> stage; past the end: a harmless re-read
    const bool live = cc < cchunks;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? (tap * Cin + cc * H3_BK) * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + 2 * H3_A_BYTES + stage * H3_B_BYTES + wave * 1024u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst, 16, b_vo[0] | tm, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst + 8192, 16, b_vo[1] | tm, soff, 0, 0);
  };
  auto load_a = [&](int t, int cc) {  // halo piece t of slab cc -> buffer cc & 1; past the end: re-read slab 0
    const bool live = cc < cchunks;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? cc * H3_BK * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + (cc & 1) * H3_A_BYTES + (t * 8 + (int)wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, a_vo[t] | tm, soff, 0, 0);
  };

  // ---- prologue: halo of slab 0, weights of steps 0 and 1 ------------------------------------------------------
#pragma unroll
  for (int t = 0; t < H3_NPIECE; ++t) load_a(t, 0);
  load_b(0, 0, 0);
  load_b(1, 0, 1);
  H3_WAIT(2);  // everything but the weights of step 1
  __builtin_amdgcn_s_barrier();

  const bool tail_half = tail <= 32;  // the last slab holds <= 32 channels: one k-substep is enough
  u32 abuf = 0;                       // byte offset of the current halo buffer
  for (int cc = 0; cc < cchunks; ++cc) {
    const bool half = tail_half && cc == cchunks - 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // loads: weights two steps ahead + (taps 0..6) one halo piece of the next slab
      if (tap + 2 < 9) load_b((tap + 2) % 3, cc, tap + 2);
      else load_b((tap + 2) % 3, cc + 1, tap + 2 - 9);
      if (tap < H3_NPIECE) load_a(tap, cc + 1);
      const u32 sboff = (u32)((tap % 3) * H3_B_BYTES);
      {
        u32x4 fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(smem + b_ad0 + sboff + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(smem + (a_ad[tap][i] + abuf));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
      }
      if (!half) {
        u32x4 fa[4], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(smem + b_ad1 + sboff + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(smem + ((a_ad[tap][i] + abuf) ^ 64u));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
      }
      if (tap < H3_NPIECE) H3_WAIT(3);
      else H3_WAIT(2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    abuf ^= (u32)H3_A_BYTES;  // (buffer 0 starts at 0, so toggling the offset is an XOR with its size: 57344 = 0xE000)
  }
  H3_WAIT(0);
  __builtin_amdgcn_s_barrier();

  // ---- epilogue ------------------------------------------------------------------------------------------------
  u16* sC = reinterpret_cast<u16*>(smem);
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  constexpr int LDC_M = H3_BN + 8;  // NHWC image sC[m][n]
  constexpr int LDC_N = 256 + 8;    // NCHW image sC[n][m]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 nl = wn * 64u + j * 16 + fr;
    const u32 n = n0 + nl;
    float sc = 1.f, bi = 0.f;
    int act = p.act;
    if (n < (u32)p.Cout) {
      if (p.scale) sc = p.scale[n];
      bi = p.bias[n];
      if ((int)n >= p.split) act = p.act2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32 ml = wm * 64u + i * 16 + fg * 4;
      u32 h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = f32_to_bits16<DT>(apply_act(acc[i][j][r] * sc + bi, act));
      if (nchw) {
        *reinterpret_cast<uint2*>(&sC[nl * LDC_N + ml]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(ml + r) * LDC_M + nl] = (u16)h[r];
      }
    }
  }
  __syncthreads();

  const int ppi = hp.th * hp.tw;  // pixels per image of the patch
  const u32 hw = (u32)(p.Ho * p.Wo);
  if (!nchw) {
    for (u32 qd = tid; qd < 256u * 16u; qd += H3_THREADS) {
      const u32 row = qd >> 4, cch = qd & 15u;
      const u32 n = n0 + cch * 8;
      if ((int)row >= npix || n >= (u32)p.Cout) continue;
      const int img = (int)row / ppi, rr = (int)row % ppi;
      const int oy = y0 + rr / hp.tw, ox = x0 + rr % hp.tw, b = b0 + img;
      if (b >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
      const size_t m = ((size_t)b * p.Ho + oy) * p.Wo + ox;
      u32x4 v = *reinterpret_cast<const u32x4*>(&sC[row * LDC_M + cch * 8]);
      u16* dst = (u16*)p.y + m * p.Cout + n;
      if (n + 8 <= (u32)p.Cout) {
        if (p.res) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>((const u16*)p.res + m * p.Cout + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bits16_to_f32<DT>(v[e] & 0xffffu) + bits16_to_f32<DT>(rv[e] & 0xffffu);
            const float hi = bits16_to_f32<DT>(v[e] >> 16) + bits16_to_f32<DT>(rv[e] >> 16);
            v[e] = f32_to_bits16<DT>(lo) | (f32_to_bits16<DT>(hi) << 16);
          }
        }
        *reinterpret_cast<u32x4*>(dst) = v;
      } else {
        for (u32 e = 0; e < 8 && n + e < (u32)p.Cout; ++e) {
          float f = bits16_to_f32<DT>((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          if (p.res) f += bits16_to_f32<DT>(((const u16*)p.res)[m * p.Cout + n + e]);
          dst[e] = (u16)f32_to_bits16<DT>(f);
        }
      }
    }
  } else {
    for (u32 qd = tid; qd < (u32)H3_BN * 32u; qd += H3_THREADS) {
      const u32 nl = qd >> 5, cch = qd & 31u;
      const u32 n = n0 + nl;
      const int ml = (int)cch * 8;
      if (n >= (u32)p.Cout || ml >= npix) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&sC[nl * LDC_N + ml]);
      u16* ybase;
      u32 ch, cy;
      if ((int)n < p.split) {
        ybase = (u16*)p.y;
        ch = n;
        cy = (u32)p.split;
      } else {
        ybase = (u16*)p.y2;
        ch = n - (u32)p.split;
        cy = (u32)(p.Cout - p.split);
      }
      if (hp.vec_nchw) {  // the 8 pixels are consecutive in x and 16-byte aligned in global memory
        const int img = ml / ppi, rr = ml % ppi;
        const int oy = y0 + rr / hp.tw, ox = x0 + rr % hp.tw, b = b0 + img;
        if (b >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
        u16* dst = ybase + ((size_t)b * cy + ch) * hw + (size_t)oy * p.Wo + ox;
        *reinterpret_cast<u32x4*>(dst) = v;
      } else {
        for (int e = 0; e < 8 && ml + e < npix; ++e) {
          const int mm = ml + e, img = mm / ppi, rr = mm % ppi;
          const int oy = y0 + rr / hp.tw, ox = x0 + rr % hp.tw, b = b0 + img;
          if (b >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
          ybase[((size_t)b * cy + ch) * hw + (size_t)oy * p.Wo + ox] = (u16)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
      }
    }
  }
}

// Patch shape for an (N, Ho, Wo) output: maximise the fraction of the 256 tile rows that are real pixels.
static bool plan_patch(int N, int Ho, int Wo, HaloParams* hp) {
  int best_imgs = 0, best_th = 0, best_tw = 0;
  double best_u = 0.0;
  auto consider = [&](int imgs, int th, int tw) {
    if (imgs < 1 || th < 1 || tw < 1 || imgs * th * tw > 256) return;
    if (imgs * (th + 2) * (tw + 2) > H3_AROWS) return;
    const long groups = (N + imgs - 1) / imgs;
    const long tiles = groups * ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
    double u = (double)N * Ho * Wo / ((double)tiles * 256.0);
    if (tw % 8 == 0 && Wo % 8 == 0) u *= 1.02;  // prefer shapes that keep the NCHW stores vectorised
    if (u > best_u) {
      best_u = u;
      best_imgs = imgs;
      best_th = th;
      best_tw = tw;
    }
  };
  if (Ho * Wo <= 256) {
    for (int imgs = 256 / (Ho * Wo); imgs >= 1; --imgs) consider(imgs, Ho, Wo);  // whole maps, as many as fit
  }
  const int tws[] = {8, 16, 24, 32, 40, 48, 64, 80, 96, 128, Wo};
  for (int tw : tws) {
    if (tw > Wo && tw != Wo) continue;
    if (tw > 254) continue;
    int th = 256 / tw;
    if (th > Ho) th = Ho;
    while (th >= 1 && (th + 2) * (tw + 2) > H3_AROWS) --th;
    consider(1, th, tw);
  }
  if (best_imgs == 0 || best_u < 0.5) return false;
  hp->imgs = best_imgs;
  hp->th = best_th;
  hp->tw = best_tw;
  hp->tiles_y = (Ho + best_th - 1) / best_th;
  hp->tiles_x = (Wo + best_tw - 1) / best_tw;
  hp->groups = (N + best_imgs - 1) / best_imgs;
  hp->hrows = best_imgs * (best_th + 2) * (best_tw + 2);
  hp->vec_nchw = (best_tw % 8 == 0 && Wo % 8 == 0) ? 1 : 0;
  return true;
}

int launch_conv3x3_halo(const ConvParams& p, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_CONV3X3_HALO") ? atoi(getenv("SSDK_CONV3X3_HALO")) : 1;
  if (!env || p.k != 3 || p.stride != 1 || p.pad != 1 || (p.Cin % 8) || p.ksplits > 1) return 1;
  if (p.Cout < 96 || p.Cin < 32) return 1;
  HaloParams hp;
  hp.c = p;
  if (!plan_patch(p.N, p.Ho, p.Wo, &hp)) return 1;
  hp.n_tiles = (p.Cout + H3_BN - 1) / H3_BN;
  const long xb = (long)p.N * p.H * p.W * p.Cin * 2, wb = (long)p.Cout * 9 * p.Cin * 2;
  if (xb >= 0xfffffff0l || wb >= 0xfffffff0l) return 1;  // 32-bit buffer offsets
  hp.x_bytes = (unsigned)xb;
  hp.w_bytes = (unsigned)wb;
  const long tiles = (long)hp.groups * hp.tiles_y * hp.tiles_x * hp.n_tiles;
  if (env != 2 && tiles < 96) return 1;  // too few tiles to fill the chip: the split-K path is faster
  if (tiles >= (1l << 31)) return 1;
  static bool attr_done[2] = {false, false};
  const int di = dtype == SSDK_BF16 ? 0 : 1;
  if (!attr_done[di]) {
    if (di == 0)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<SSDK_BF16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
    else
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<SSDK_F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
    attr_done[di] = true;
  }
  if (di == 0)
    hipLaunchKernelGGL((conv3x3_halo_kernel<SSDK_BF16>), dim3((unsigned)tiles), dim3(H3_THREADS), H3_LDS, stream, hp);
  else
    hipLaunchKernelGGL((conv3x3_halo_kernel<SSDK_F16>), dim3((unsigned)tiles), dim3(H3_THREADS), H3_LDS, stream, hp);
  return check_launch("conv3x3_halo_kernel");
}

}  // namespace ssdk
