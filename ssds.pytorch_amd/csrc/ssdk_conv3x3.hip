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
  unsigned mg_ntiles, mg_tx, mg_ty, mg_ppi, mg_tw, mg_hw2, mg_halo;  // ceil(2^32 / d) of the divisors below
  int ksplits, c_per;        // split-K over the 64-channel slabs: slice z owns slabs [z*c_per, min((z+1)*c_per, cchunks))
  unsigned mg_ks;            // (grids too small to fill the chip: few pixels, long K -- the 8x8 ... 4x4 head levels)
  float* slabs;              // [tile][slice][16 fragments][512 lanes][4] fp32 partial accumulators
  unsigned* counters;        // [tile] arrival tickets, zero on entry, re-armed by the last arriver
  unsigned y_bytes, y2_bytes, res_bytes;  // persistent form: ranges of the output / residual descriptors (0: tensor absent)
  unsigned mg_wo;            // ceil(2^32 / Wo)
  unsigned tq, tr;           // persistent form: workgroup g owns tiles [g*tq + min(g, tr), +tq + (g < tr)) of the launch's tiles
  long long* dbg;             // SSDK_H3_DBG=1: cycle stamps of one workgroup / wave 0 (4 per k-step)
  unsigned dbg_wg;            // which workgroup is stamped (SSDK_H3_DBG_WG: 0 = the first, cold one; -1 = the last to start)
};

// n / d for n*d < 2^32 with M = ceil(2^32 / d) (host side: magic()); d == 1 has no 32-bit magic
// (branch-free: the d == 1 test as a select -- written as a ternary around the multiply it compiled to ~60 uniform branches in
//  the kernel's set-up, which the stamps showed at 5 k cycles per workgroup)
__device__ __forceinline__ u32 fdiv(u32 n, u32 d, u32 M) {
  const u32 q = __umulhi(n, M);
  const u32 one = (u32)-(int)(d == 1u);  // all ones when d == 1
  return (n & one) | (q & ~one);
}

#define H3_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define H3_STAMP(slot)                                                                      \
  do {                                                                                      \
    if (hp.dbg && blockIdx.x == hp.dbg_wg && tid == 0 && stamp_i < 60)                              \
      hp.dbg[stamp_i * 4 + (slot)] = (long long)__builtin_readcyclecounter();               \
  } while (0)

template <int DT>
__global__ __launch_bounds__(H3_THREADS) void conv3x3_halo_kernel(const HaloParams hp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvParams& p = hp.c;
  const u32 tid = threadIdx.x, lane = tid & 63u;
#define H3_MARK(slot)                                                                                   \
  do {                                                                                                  \
    if (hp.dbg && blockIdx.x == hp.dbg_wg && tid == 0) hp.dbg[240 + (slot)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
  H3_MARK(0);
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 wm = wave >> 1, wn = wave & 1u;

  // XCD-aware (bijective) order; n-tile fastest so that the tiles of one patch share an XCD's L2
  const u32 nwg = gridDim.x, id = blockIdx.x;
  const u32 q8 = nwg >> 3, r8 = nwg & 7u, xcd = id & 7u;
  const u32 lin = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + (id >> 3);
  const u32 tile_id = fdiv(lin, (u32)hp.ksplits, hp.mg_ks);  // a tile's slices are neighbours: same XCD
  const u32 kz = lin - tile_id * (u32)hp.ksplits;
  u32 pt = fdiv(tile_id, (u32)hp.n_tiles, hp.mg_ntiles);
  const u32 nt = tile_id - pt * (u32)hp.n_tiles;
  u32 pq = fdiv(pt, (u32)hp.tiles_x, hp.mg_tx);
  const u32 tx = pt - pq * (u32)hp.tiles_x;
  const u32 grp = fdiv(pq, (u32)hp.tiles_y, hp.mg_ty);
  const u32 ty = pq - grp * (u32)hp.tiles_y;
  const int b0 = (int)grp * hp.imgs, y0 = (int)ty * hp.th, x0 = (int)tx * hp.tw;
  const u32 n0 = nt * H3_BN;


  const int Cin = p.Cin, H = p.H, W = p.W;
  const int HW2 = hp.tw + 2, HH2 = hp.th + 2;
  const int Ktot = 9 * Cin;
  const int cchunks = (Cin + H3_BK - 1) / H3_BK;
  const int c0 = (int)kz * hp.c_per;                                        // this slice's slabs: [c0, c1)
  const int c1 = c0 + hp.c_per < cchunks ? c0 + hp.c_per : cchunks;

  // ---- loader roles ------------------------------------------------------------------------------------------
  // Loads are buffer_load_dwordx4 ... lds through raw buffer descriptors: address = base + soffset (uniform: slab /
  // tap) + voffset (per lane, fixed for the whole kernel), and a lane that must read zeros (outside the image,
  // channel >= Cin, row >= Cout) simply carries an out-of-range voffset -- no per-step address arithmetic at all.
  const u32 lrow = lane >> 3;
  // Bank swizzle of the 128-byte LDS rows.  Round 1-3: physical chunk = logical ^ ((row >> 1) & 7), conflict-free for sixteen
  // lane-CONTIGUOUS reads of consecutive rows.  ds_read_b128 is serviced in four groups of 16 lanes that are NOT contiguous
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: eight pixels of k-piece fg with the other eight of piece fg + 1,
  // MI355X_MICROARCH.md), and for those the XOR form collides on every tap with an odd row shift (measured: 26 % of the
  // LDS cycles).  Round 4: physical chunk = (logical + row) & 7 -- the sixteen 16-byte slots
  // ((row & 1) * 8 + ((fg' + row) & 7)) of such a group are all different for every alignment (enumerated: tools/lds_groups.py).
  const u32 lchunk = ((lane & 7u) - (lane >> 3)) & 7u;  // logical chunk of this lane's slot
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
    // (computed for every lane, then selected: no divergent branches in the set-up; the tensor is < 4 GiB, so the offset of an
    //  in-image pixel fits 32 bits and the others are discarded)
    const int img = (int)fdiv((u32)hr, (u32)(HH2 * HW2), hp.mg_halo), rr = hr - img * (HH2 * HW2);
    const int hy = (int)fdiv((u32)rr, (u32)HW2, hp.mg_hw2), hx = rr - hy * HW2;
    const int b = b0 + img, iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool inside = hr < hp.hrows && b < p.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const u32 pix = ((u32)b * (u32)H + (u32)iy) * (u32)W + (u32)ix;
    a_vo[t] = inside ? (pix * (u32)Cin + (u32)lci) * 2u : OOB;
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
      const int img = (int)fdiv((u32)ml, (u32)(hp.th * hp.tw), hp.mg_ppi), rr = ml - img * (hp.th * hp.tw);
      const int y = (int)fdiv((u32)rr, (u32)hp.tw, hp.mg_tw), x = rr - y * hp.tw;
      a_hr[i] = (img * HH2 + y) * HW2 + x;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32 hr = (u32)(a_hr[i] + (tap / 3) * HW2 + (tap % 3));
        a_ad[tap][i] = hr * 128u + (((fg + hr) & 7u) << 4);
      }
  }
  // weights: stage / fragment column are immediates on top of these two (k-substep 0 / 1)
  const u32 b_ad0 = 2u * H3_A_BYTES + (wn * 64u + fr) * 128u + (((fg + fr) & 7u) << 4);
  const u32 b_ad1 = b_ad0 ^ 64u;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_b = [&](int stage, int cc, int tap) {  // weights of (slab cc, tap) -> stage; past the end: a harmless re-read
    const bool live = cc < c1;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? (tap * Cin + cc * H3_BK) * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + 2 * H3_A_BYTES + stage * H3_B_BYTES + wave * 1024u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst, 16, b_vo[0] | tm, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst + 8192, 16, b_vo[1] | tm, soff, 0, 0);
  };
  auto load_a = [&](int t, int cc) {  // halo piece t of slab cc -> buffer (cc - c0) & 1; past the end: re-read slab 0
    const bool live = cc < c1;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? cc * H3_BK * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + ((cc - c0) & 1) * H3_A_BYTES + (t * 8 + (int)wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, a_vo[t] | tm, soff, 0, 0);
  };

  // ---- prologue: halo of slab 0, weights of steps 0 and 1 ------------------------------------------------------
#pragma unroll
  for (int t = 0; t < H3_NPIECE; ++t) load_a(t, c0);
  load_b(0, c0, 0);
  load_b(1, c0, 1);
  // per-column scale / bias of this lane's four accumulator columns (epilogue).  Requested HERE, behind the prologue's LDS-DMA
  // loads: at the top of the kernel the compiler waits for every ordinary VGPR load before the first buffer_load ... lds is
  // issued (a whole memory round trip in the set-up of every workgroup), in the epilogue -- rounds 1-3 -- the round trip was
  // exposed there (~2 k of its 7.5 k cycles); here it rides under the wait for the first tile.
  // (eight unconditional loads from clamped indices, issued back to back; the selects happen in the epilogue -- written with
  //  the loads inside `if (n < Cout)` the compiler put an s_waitcnt vmcnt(0) behind every pair: four round trips in a row)
  float ld_sc[4], ld_bi[4];
  // H3_WAIT(10) below counts on this ORDER (nine tile loads, then 2 + 8 younger requests): pinned for the IR passes (memory
  // clobber) and for the machine scheduler (sched_barrier) -- a constant load hoisted above a tile load would let the
  // barrier pass before that tile has landed
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  {
    const float* scp = p.scale ? p.scale : p.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 n = n0 + wn * 64u + j * 16 + (lane & 15u);
      n = n < (u32)p.Cout ? n : (u32)p.Cout - 1u;
      ld_sc[j] = scp[n];
      ld_bi[j] = p.bias[n];
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  H3_MARK(1);
  H3_WAIT(10);  // everything but the weights of step 1 and the eight scale / bias loads behind them
  __builtin_amdgcn_s_barrier();
  H3_MARK(2);

  // ---- main loop: two phases per k-step, the two wave groups staggered by one phase ---------------------------
  //   R(k): issue the loads of step k+2, read ALL fragments of step k into registers, counted vmcnt wait
  //   M(k): 32 MFMAs
  // separated by workgroup barriers.  Waves w and w+4 share a SIMD; group 1 (waves 4..7) runs one extra barrier
  // first and therefore stays one phase behind: while a SIMD's group-0 wave is in M(k) its group-1 wave is in
  // R(k), then group 1 runs M(k) while group 0 reads R(k+1) -- the matrix pipe of every SIMD always has a wave
  // with operands ready, and no wave waits for LDS latency while holding the pipe.
  // Hazards (slot 2k = group 0 in R(k) / group 1 in M(k-1); slot 2k+1 = group 0 in M(k) / group 1 in R(k)):
  //   weights of step k+1 are waited for at the end of R(k) by both groups (slots 2k, 2k+1) and first read in slot
  //   2k+2; their stage was last read in slot 2k-3 and is refilled from slot 2k-2 on; halo pieces likewise.
  const bool tail_half = tail <= 32;  // the last slab holds <= 32 channels: one k-substep is enough
  const bool g1 = wave >= 4u;
  u32 abuf = 0;                       // byte offset of the current halo buffer
  if (g1) __builtin_amdgcn_s_barrier();
  for (int cc = c0; cc < c1; ++cc) {
    const bool half = tail_half && cc == cchunks - 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int stamp_i = (cc - c0) * 9 + tap;
      H3_STAMP(0);
      // R: loads two steps ahead (+ taps 0..5: one halo piece of the next slab), then this step's fragments
      if (tap + 2 < 9) load_b((tap + 2) % 3, cc, tap + 2);
      else load_b((tap + 2) % 3, cc + 1, tap + 2 - 9);
      if (tap < H3_NPIECE) load_a(tap, cc + 1);
      const u32 sboff = (u32)((tap % 3) * H3_B_BYTES);
      u32x4 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const u32x4*>(smem + b_ad0 + sboff + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa0[i] = *reinterpret_cast<const u32x4*>(smem + (a_ad[tap][i] + abuf));
      if (!half) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const u32x4*>(smem + b_ad1 + sboff + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa1[i] = *reinterpret_cast<const u32x4*>(smem + ((a_ad[tap][i] + abuf) ^ 64u));
      }
      H3_STAMP(1);
      if (tap < H3_NPIECE) H3_WAIT(3);
      else H3_WAIT(2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      H3_STAMP(2);
      // M
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa0[i], fb0[j], acc[i][j]);
      if (!half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa1[i], fb1[j], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      H3_STAMP(3);
    }
    abuf ^= (u32)H3_A_BYTES;
  }
  if (!g1) __builtin_amdgcn_s_barrier();  // re-align the groups
  H3_WAIT(0);
  __builtin_amdgcn_s_barrier();
  H3_MARK(3);

  if (hp.ksplits > 1) {
    // ---- split-K hand-off (cdna_hip_programming.md G16, counter form; same protocol as conv_gemm_kernel): slab stores ->
    //      every wave drains vmcnt -> barrier -> one lane: agent release + drained wait -> relaxed ticket; the last arriver
    //      acquires once, then every wave sums the slabs IN SLICE ORDER (bit-reproducible whoever arrives last)
    float* my = hp.slabs + ((size_t)tile_id * hp.ksplits + kz) * (size_t)(16 * H3_THREADS * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(my + ((size_t)(i * 4 + j) * H3_THREADS + tid) * 4) = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32* flag = reinterpret_cast<u32*>(smem);  // (the halo buffers are free now)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(hp.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned last = (old == (unsigned)hp.ksplits - 1u) ? 1u : 0u;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(hp.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
      *flag = last;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();  // everyone has read the flag before the epilogue reuses the LDS
    if (!last) return;
    const float* base = hp.slabs + (size_t)tile_id * hp.ksplits * (size_t)(16 * H3_THREADS * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(base + ((size_t)(i * 4 + j) * H3_THREADS + tid) * 4);
    for (int z = 1; z < hp.ksplits; ++z) {
      const float* other = base + (size_t)z * (size_t)(16 * H3_THREADS * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(other + ((size_t)(i * 4 + j) * H3_THREADS + tid) * 4);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  float e_sc[4], e_bi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 n = n0 + wn * 64u + j * 16 + (lane & 15u);
    e_sc[j] = (n < (u32)p.Cout && p.scale) ? ld_sc[j] : 1.f;
    e_bi[j] = n < (u32)p.Cout ? ld_bi[j] : 0.f;
  }
  u16* sC = reinterpret_cast<u16*>(smem);
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  constexpr int LDC_M = H3_BN + 8;  // NHWC image sC[m][n]
  constexpr int LDC_N = 256 + 8;    // NCHW image sC[n][m]
  const bool any_sig = act_is_sig(p.act) || act_is_sig(p.act2);
  const bool any_clamp = act_is_clamp(p.act) || act_is_clamp(p.act2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 nl = wn * 64u + j * 16 + fr;
    const u32 n = n0 + nl;
    const float sc = e_sc[j], bi = e_bi[j];
    const ActSel as = act_sel((int)n >= p.split ? p.act2 : p.act);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32 ml = wm * 64u + i * 16 + fg * 4;
      const uint2 h = epilogue4<DT>(acc[i][j], sc, bi, as, any_sig, any_clamp);
      if (nchw) {
        *reinterpret_cast<uint2*>(&sC[nl * LDC_N + ml]) = h;
      } else {
        sC[(ml + 0) * LDC_M + nl] = (u16)(h.x & 0xffffu);
        sC[(ml + 1) * LDC_M + nl] = (u16)(h.x >> 16);
        sC[(ml + 2) * LDC_M + nl] = (u16)(h.y & 0xffffu);
        sC[(ml + 3) * LDC_M + nl] = (u16)(h.y >> 16);
      }
    }
  }
  __syncthreads();
  H3_MARK(4);

  const int ppi = hp.th * hp.tw;  // pixels per image of the patch
  const u32 hw = (u32)(p.Ho * p.Wo);
  auto pixel_of = [&](int ml, u32* b_out, u32* pix_out) -> bool {  // tile row -> (image, oy*Wo + ox)
    if (ml >= npix) return false;
    const int img = (int)fdiv((u32)ml, (u32)ppi, hp.mg_ppi), rr = ml - img * ppi;
    const int yy = (int)fdiv((u32)rr, (u32)hp.tw, hp.mg_tw);
    const int oy = y0 + yy, ox = x0 + (rr - yy * hp.tw), b = b0 + img;
    if (b >= p.N || oy >= p.Ho || ox >= p.Wo) return false;
    *b_out = (u32)b;
    *pix_out = (u32)(oy * p.Wo + ox);
    return true;
  };
  if (!nchw) {
    const u32 cch = tid & 15u, n = n0 + cch * 8;  // this thread's 8-channel column group in every iteration
    if (n < (u32)p.Cout) {
#pragma unroll 2
      for (u32 it = 0; it < 8; ++it) {
        const u32 row = (tid >> 4) + it * 32u;
        u32 pb, pp;
        if (!pixel_of((int)row, &pb, &pp)) continue;
        const size_t m = (size_t)pb * hw + pp;
        u32x4 v = *reinterpret_cast<const u32x4*>(&sC[row * LDC_M + cch * 8]);
        u16* dst = (u16*)p.y + m * p.Cout + n;
        if (n + 8 <= (u32)p.Cout) {
          if (p.res) v = add_residual8<DT>(v, (const u16*)p.res + res_pixel_offset(p, (u32)m) + n, p.post);
          *reinterpret_cast<u32x4*>(dst) = v;
        } else {
          for (u32 e = 0; e < 8 && n + e < (u32)p.Cout; ++e) {
            float f = bits16_to_f32<DT>((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            if (p.res) f = post_act(f + bits16_to_f32<DT>(((const u16*)p.res)[res_pixel_offset(p, (u32)m) + n + e]), p.post);
            dst[e] = (u16)f32_to_bits16<DT>(f);
          }
        }
      }
    }
  } else {
    // this thread owns the same 8 pixels (tile rows ml .. ml+7) in every iteration; only the channel changes
    const int ml = (int)(tid & 31u) * 8;
    u32 b8[8], p8[8];
    bool ok8[8];
    if (hp.vec_nchw) {
      ok8[0] = pixel_of(ml, &b8[0], &p8[0]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) ok8[e] = pixel_of(ml + e, &b8[e], &p8[e]);
    }
    for (u32 it = 0; it < (u32)H3_BN / 16u; ++it) {
      const u32 nl = (tid >> 5) + it * 16u;
      const u32 n = n0 + nl;
      if (n >= (u32)p.Cout) break;
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
        if (!ok8[0]) break;
        *reinterpret_cast<u32x4*>(ybase + ((size_t)b8[0] * cy + ch) * hw + p8[0]) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (!ok8[e]) continue;
          ybase[((size_t)b8[e] * cy + ch) * hw + p8[e]] = (u16)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
      }
    }
  }
  H3_MARK(5);
}

// ---- persistent form (round 6) -----------------------------------------------------------------------------------------------
// One workgroup per CU walks a contiguous range of the launch's tiles (the n-tiles of a patch are neighbours: the second one
// finds the halo in L2).  Per tile the first form pays 2.5 k cycles of set-up, ~2 k of first-load latency, 7.5 k of epilogue
// arithmetic + LDS staging and 4.6 k of stores + drain around a 51 k loop (256 -> 256 tower tile, stamps of round 5), all of it
// serial because 160 KiB of LDS admit one workgroup per CU.  Here
//   * the epilogue goes from the accumulator registers straight to global memory: NCHW -- a lane holds four consecutive pixels
//     of a channel = 8 contiguous bytes of a plane; NHWC -- the LOADER places weight row n0 + (q & 64) + (q & 15) * 4 + ((q >> 4) & 3)
//     at LDS row q, so that the four accumulator columns of a lane are four CONSECUTIVE channels: 8 contiguous bytes of a pixel,
//     16 lanes = 128 bytes (the permutation costs nothing: it is a different voffset per loader lane).  No LDS image, no
//     barrier, no 2-byte staging writes;
//   * the next tile's halo slab 0 and first two weight stages are requested BEFORE the epilogue arithmetic of the current tile
//     and land behind it; the set-up (roles, descriptors, fragment addresses) happens once per workgroup.
// Takes ksplits == 1, patches with tw % 4 == 0, NCHW with Wo % 4 == 0, NHWC with Cout % 4 == 0; everything else stays on the first form.
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
// epilogue4 (ssdk_conv_common.h) with the per-lane activation choice as bit selects: the ternaries of the shared helper compile to
// an exec-masked branch per VALUE on split heads (64 per tile), which a workgroup that is alone on its CU pays in full
template <int DT, bool SIG>
__device__ __forceinline__ uint2 epilogue4_flat(const f32x4 acc, float sc, float bi, float lo, float hi, u32 m_sig, u32 m_silu,
                                                bool any_clamp) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r] * sc + bi;
  if constexpr (SIG) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
      const u32 vb = __builtin_bit_cast(u32, v[r]), sb = __builtin_bit_cast(u32, sg), pb = __builtin_bit_cast(u32, v[r] * sg);
      const u32 t = (pb & m_silu) | (vb & ~m_silu);
      v[r] = __builtin_bit_cast(float, (sb & m_sig) | (t & ~m_sig));
    }
  }
  if (any_clamp) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], lo), hi);
  }
  return make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
}

template <int DT, bool NHWC>
__global__ __launch_bounds__(H3_THREADS) void conv3x3_halop_kernel(const HaloParams hp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvParams& p = hp.c;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 wm = wave >> 1, wn = wave & 1u;

  const u32 nwg = gridDim.x, id = blockIdx.x;
  const u32 q8 = nwg >> 3, r8 = nwg & 7u, xcd = id & 7u;
  const u32 lin = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + (id >> 3);  // neighbours in lin share an XCD
  const u32 t_begin = lin * hp.tq + (lin < hp.tr ? lin : hp.tr);
  const u32 t_cnt = hp.tq + (lin < hp.tr ? 1u : 0u);
#define H3P_STAMP(idx)                                                                                        \
  do {                                                                                                        \
    if (hp.dbg && lin == hp.dbg_wg && tid == 0 && (idx) < 256) hp.dbg[(idx)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
  H3P_STAMP(240);

  const int Cin = p.Cin, H = p.H, W = p.W;
  const int HW2 = hp.tw + 2, HH2 = hp.th + 2;
  const int Ktot = 9 * Cin;
  const int cchunks = (Cin + H3_BK - 1) / H3_BK;

  const u32 lrow = lane >> 3;
  const u32 lchunk = ((lane & 7u) - (lane >> 3)) & 7u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, hp.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, hp.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, hp.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc(p.y2 ? p.y2 : p.y, 0, hp.y2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x), 0, hp.res_bytes, 0x00020000);
  constexpr u32 OOB = 0xfffffff0u;
  const int lci = (int)lchunk * 8;
  const int tail = Cin - (cchunks - 1) * H3_BK;
  const u32 tailmask = lci < tail ? 0u : OOB;

  const u32 fr = lane & 15u, fg = lane >> 4;
  const int npix = hp.imgs * hp.th * hp.tw;
  const int ppi = hp.th * hp.tw;
  u32 a_ad[9][4];
  {
    int a_hr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ml = (int)(wm * 64u + i * 16 + fr);
      if (ml >= npix) ml = 0;
      const int img = (int)fdiv((u32)ml, (u32)ppi, hp.mg_ppi), rr = ml - img * ppi;
      const int y = (int)fdiv((u32)rr, (u32)hp.tw, hp.mg_tw), x = rr - y * hp.tw;
      a_hr[i] = (img * HH2 + y) * HW2 + x;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32 hr = (u32)(a_hr[i] + (tap / 3) * HW2 + (tap % 3));
        a_ad[tap][i] = hr * 128u + (((fg + hr) & 7u) << 4);
      }
  }
  const u32 b_ad0 = 2u * H3_A_BYTES + (wn * 64u + fr) * 128u + (((fg + fr) & 7u) << 4);
  const u32 b_ad1 = b_ad0 ^ 64u;

  // ---- per-tile state ------------------------------------------------------------------------------------------------------
  int b0 = 0, y0 = 0, x0 = 0;
  u32 n0 = 0;
  u32 a_vo[H3_NPIECE], b_vo[2];
  u32 cur_pt = 0xffffffffu;
  auto set_tile = [&](u32 tile) {
    const u32 pt = fdiv(tile, (u32)hp.n_tiles, hp.mg_ntiles);
    const u32 nt = tile - pt * (u32)hp.n_tiles;
    n0 = nt * H3_BN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32 q = ((u32)j * 8u + wave) * 8u + lrow;  // LDS row of this lane's chunk
      const u32 n = n0 + (NHWC ? (q & 64u) + (q & 15u) * 4u + ((q >> 4) & 3u) : q);
      b_vo[j] = n < (u32)p.Cout ? (u32)(((long)n * Ktot + lci) * 2) : OOB;
    }
    if (pt == cur_pt) return;  // the next n-tile of the same patch: the halo offsets stand
    cur_pt = pt;
    const u32 pq = fdiv(pt, (u32)hp.tiles_x, hp.mg_tx);
    const u32 tx = pt - pq * (u32)hp.tiles_x;
    const u32 grp = fdiv(pq, (u32)hp.tiles_y, hp.mg_ty);
    const u32 ty = pq - grp * (u32)hp.tiles_y;
    b0 = (int)grp * hp.imgs;
    y0 = (int)ty * hp.th;
    x0 = (int)tx * hp.tw;
#pragma unroll
    for (int t = 0; t < H3_NPIECE; ++t) {
      const int hr = (t * 8 + (int)wave) * 8 + (int)lrow;
      const int img = (int)fdiv((u32)hr, (u32)(HH2 * HW2), hp.mg_halo), rr = hr - img * (HH2 * HW2);
      const int hy = (int)fdiv((u32)rr, (u32)HW2, hp.mg_hw2), hx = rr - hy * HW2;
      const int b = b0 + img, iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool inside = hr < hp.hrows && b < p.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const u32 pix = ((u32)b * (u32)H + (u32)iy) * (u32)W + (u32)ix;
      a_vo[t] = inside ? (pix * (u32)Cin + (u32)lci) * 2u : OOB;
    }
  };
  auto load_b = [&](int stage, int cc, int tap) {
    const bool live = cc < cchunks;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? (tap * Cin + cc * H3_BK) * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + 2 * H3_A_BYTES + stage * H3_B_BYTES + wave * 1024u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst, 16, b_vo[0] | tm, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst + 8192, 16, b_vo[1] | tm, soff, 0, 0);
  };
  auto load_a = [&](int t, int cc) {
    const bool live = cc < cchunks;
    const u32 tm = (live && cc == cchunks - 1) ? tailmask : 0u;
    const int soff = live ? cc * H3_BK * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + (cc & 1) * H3_A_BYTES + (t * 8 + (int)wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, a_vo[t] | tm, soff, 0, 0);
  };
  auto issue_prologue = [&]() {  // 7 halo pieces, then the weights of steps 0 and 1: eleven LDS-DMA instructions, in this order
#pragma unroll
    for (int t = 0; t < H3_NPIECE; ++t) load_a(t, 0);
    load_b(0, 0, 0);
    load_b(1, 0, 1);
  };
  // the four accumulator columns of this lane: channels col_n(j)
  auto col_n = [&](u32 tn0, int j) -> u32 { return tn0 + wn * 64u + (NHWC ? fr * 4u + (u32)j : (u32)j * 16u + fr); };
  float ld_sc[4], ld_bi[4];
  auto load_scbi = [&]() {  // (unconditional loads from clamped indices: see the first form)
    const float* scp = p.scale ? p.scale : p.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 n = col_n(n0, j);
      n = n < (u32)p.Cout ? n : (u32)p.Cout - 1u;
      ld_sc[j] = scp[n];
      ld_bi[j] = p.bias[n];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool tail_half = tail <= 32;
  const bool g1 = wave >= 4u;
  const bool any_sig = act_is_sig(p.act) || act_is_sig(p.act2);
  const bool any_clamp = act_is_clamp(p.act) || act_is_clamp(p.act2);
  const u32 hw = (u32)(p.Ho * p.Wo);
  const bool has_res = p.res != nullptr, two_out = p.split < p.Cout;
  const u32 rsh = (u32)(p.res_mode & 1);
  const float post_lo = (p.post == SSDK_ACT_RELU || p.post == SSDK_ACT_RELU6) ? 0.f : -__builtin_inff();
  const float post_hi = p.post == SSDK_ACT_RELU6 ? 6.f : __builtin_inff();

  set_tile(t_begin);
  issue_prologue();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  H3_WAIT(2);  // everything but the weights of step 1
  load_scbi();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  H3P_STAMP(241);

  for (u32 it = 0; it < t_cnt; ++it) {
    H3P_STAMP(it * 8u + 0u);
    // ---- main loop: exactly the first form's (two phases per k-step, the wave groups one phase apart) -------------------------
    u32 abuf = 0;
    if (g1) __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < cchunks; ++cc) {
      const bool half = tail_half && cc == cchunks - 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 2 < 9) load_b((tap + 2) % 3, cc, tap + 2);
        else load_b((tap + 2) % 3, cc + 1, tap + 2 - 9);
        if (tap < H3_NPIECE) load_a(tap, cc + 1);
        const u32 sboff = (u32)((tap % 3) * H3_B_BYTES);
        u32x4 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const u32x4*>(smem + b_ad0 + sboff + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa0[i] = *reinterpret_cast<const u32x4*>(smem + (a_ad[tap][i] + abuf));
        if (!half) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const u32x4*>(smem + b_ad1 + sboff + j * 2048);
#pragma unroll
          for (int i = 0; i < 4; ++i) fa1[i] = *reinterpret_cast<const u32x4*>(smem + ((a_ad[tap][i] + abuf) ^ 64u));
        }
        if (tap < H3_NPIECE) H3_WAIT(3);
        else H3_WAIT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa0[i], fb0[j], acc[i][j]);
        if (!half) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa1[i], fb1[j], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      abuf ^= (u32)H3_A_BYTES;
    }
    if (!g1) __builtin_amdgcn_s_barrier();  // re-align the groups
    H3P_STAMP(it * 8u + 1u);
    H3_WAIT(0);                             // (the loop's harmless loads past the end; the scale / bias loads of this tile)
    __builtin_amdgcn_s_barrier();
    H3P_STAMP(it * 8u + 2u);

    // ---- tile boundary -----------------------------------------------------------------------------------------------------------
    // this tile's epilogue constants and coordinates, before the per-tile state moves on
    float e_sc[4], e_bi[4], e_lo[4], e_hi[4];
    u32 e_m1[4], e_m2[4], e_n[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 n = col_n(n0, j);
      e_n[j] = n;
      e_sc[j] = (n < (u32)p.Cout && p.scale) ? ld_sc[j] : 1.f;
      e_bi[j] = n < (u32)p.Cout ? ld_bi[j] : 0.f;
      {  // (act_sel without its if-chain: the activation differs per lane on split heads)
        const int a = (int)n >= p.split ? p.act2 : p.act;
        e_lo[j] = (a == SSDK_ACT_RELU || a == SSDK_ACT_RELU6) ? 0.f : -__builtin_inff();
        e_hi[j] = a == SSDK_ACT_RELU6 ? 6.f : __builtin_inff();
        e_m1[j] = (u32)-(int)(a == SSDK_ACT_SIGMOID);
        e_m2[j] = (u32)-(int)(a == SSDK_ACT_SILU);
      }
      asm volatile("" : "+v"(e_sc[j]), "+v"(e_bi[j]));  // the loads are consumed HERE: no compiler wait behind the LDS-DMA requests below
    }
    const int cb0 = b0, cy0 = y0, cx0 = x0;
    const bool more = it + 1u < t_cnt;
    if (more) {
      set_tile(t_begin + it + 1u);
      issue_prologue();  // lands behind the arithmetic below; the buffers are free: every wave is past its last fragment read
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    H3P_STAMP(it * 8u + 3u);
    // (branch-free: an element that must not be stored carries an out-of-range buffer offset -- the set-up of the first form
    //  taught that ~60 short branches cost a workgroup 5 k cycles)
    // tile row -> (image, oy, ox) of THIS tile; false: not a pixel of the output (the four rows ml .. ml+3 of an accumulator
    // fragment are consecutive x of one map row: tw % 4 == 0)
    auto pixel_of = [&](int ml, u32* b_out, u32* oy_out, u32* ox_out) -> bool {
      const int img = (int)fdiv((u32)ml, (u32)ppi, hp.mg_ppi), rr = ml - img * ppi;
      const int yy = (int)fdiv((u32)rr, (u32)hp.tw, hp.mg_tw);
      const int oy = cy0 + yy, ox = cx0 + (rr - yy * hp.tw), b = cb0 + img;
      *b_out = (u32)b;
      *oy_out = (u32)oy;
      *ox_out = (u32)ox;
      return (ml < npix) & (b < p.N) & (oy < p.Ho) & (ox < p.Wo);
    };
    // one accumulator row fragment (16 pixels x the lane's four columns) at a time: arithmetic, then its stores
    auto fragments = [&](auto SIG, auto FLAG) {  // FLAG: NHWC -- a residual is added; NCHW -- two output tensors (split heads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = epilogue4_flat<DT, decltype(SIG)::value>(acc[i][j], e_sc[j], e_bi[j], e_lo[j], e_hi[j], e_m1[j], e_m2[j], any_clamp);
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int ml = (int)(wm * 64u + i * 16 + fg * 4);
      if constexpr (NHWC) {
        const u32 nb = e_n[0];  // four consecutive channels from here (Cout % 4 == 0: all inside or all outside)
        uint2 o[4];             // row r: channels nb .. nb+3
        o[0] = make_uint2(__builtin_amdgcn_perm(h[1].x, h[0].x, 0x05040100u), __builtin_amdgcn_perm(h[3].x, h[2].x, 0x05040100u));
        o[1] = make_uint2(__builtin_amdgcn_perm(h[1].x, h[0].x, 0x07060302u), __builtin_amdgcn_perm(h[3].x, h[2].x, 0x07060302u));
        o[2] = make_uint2(__builtin_amdgcn_perm(h[1].y, h[0].y, 0x05040100u), __builtin_amdgcn_perm(h[3].y, h[2].y, 0x05040100u));
        o[3] = make_uint2(__builtin_amdgcn_perm(h[1].y, h[0].y, 0x07060302u), __builtin_amdgcn_perm(h[3].y, h[2].y, 0x07060302u));
        u32 pb, oy, ox0;
        const bool okb = pixel_of(ml, &pb, &oy, &ox0) & (nb < (u32)p.Cout);
        const u32 m0 = (pb * (u32)p.Ho + oy) * (u32)p.Wo + ox0;
        const u32 rrow = (pb * ((u32)p.Ho >> rsh) + (oy >> rsh)) * ((u32)p.Wo >> rsh);  // rsh = 1: half-resolution residual
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = okb & (ox0 + (u32)r < (u32)p.Wo);
          const u32 m = m0 + (u32)r;
          uint2 v = o[r];
          if constexpr (decltype(FLAG)::value) {  // (values already rounded to 16 bit, like the framework's tensor add)
            const u32 rpix = rrow + ((ox0 + (u32)r) >> rsh);
            const v2u rv = __builtin_amdgcn_raw_buffer_load_b64(rr_, (int)(ok ? (rpix * (u32)p.Cout + nb) * 2u : OOB), 0, 0);
            auto addc = [&](u32 a, u32 b) { return __builtin_fminf(__builtin_fmaxf(bits16_to_f32<DT>(a) + bits16_to_f32<DT>(b), post_lo), post_hi); };
            v.x = pack2_16<DT>(addc(v.x & 0xffffu, rv.x & 0xffffu), addc(v.x >> 16, rv.x >> 16));
            v.y = pack2_16<DT>(addc(v.y & 0xffffu, rv.y & 0xffffu), addc(v.y >> 16, rv.y >> 16));
          }
          __builtin_amdgcn_raw_buffer_store_b64(v2u{v.x, v.y}, yr, (int)(ok ? (m * (u32)p.Cout + nb) * 2u : OOB), 0, 0);
        }
      } else {
        // four consecutive pixels (tw % 4 == 0, Wo % 4 == 0: same map row, 8-byte aligned) of channel e_n[j]
        u32 pb, oy, ox0;
        const bool okp = pixel_of(ml, &pb, &oy, &ox0);
        const u32 pp = oy * (u32)p.Wo + ox0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 n = e_n[j];
          const bool first = (int)n < p.split;
          const u32 ch = first ? n : n - (u32)p.split;
          const u32 cy = first ? (u32)p.split : (u32)(p.Cout - p.split);
          const u32 off = (okp & (n < (u32)p.Cout)) ? ((pb * cy + ch) * hw + pp) * 2u : OOB;
          // (two descriptors, one store each: the invalid one is out of range)
          __builtin_amdgcn_raw_buffer_store_b64(v2u{h[j].x, h[j].y}, yr, (int)(first ? off : OOB), 0, 0);
          if constexpr (decltype(FLAG)::value) __builtin_amdgcn_raw_buffer_store_b64(v2u{h[j].x, h[j].y}, y2r, (int)(first ? OOB : off), 0, 0);
        }
      }
    }
    };
    const bool flag = NHWC ? has_res : two_out;
    if (any_sig) {
      if (flag) fragments(std::true_type{}, std::true_type{});
      else fragments(std::true_type{}, std::false_type{});
    } else {
      if (flag) fragments(std::false_type{}, std::true_type{});
      else fragments(std::false_type{}, std::false_type{});
    }
    __builtin_amdgcn_sched_barrier(0);
    // loads complete in order among loads: <= 2 operations outstanding means the next tile's halo and step-0 weights have landed
    // (and all but two of the stores above are acknowledged -- the first k-step's counted wait would ask for that anyway)
    H3P_STAMP(it * 8u + 4u);
    if (more) H3_WAIT(2);
    __builtin_amdgcn_sched_barrier(0);
    H3P_STAMP(it * 8u + 5u);
    if (more) load_scbi();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // every wave's pieces of the next tile have landed (each waited for its own above)
    H3P_STAMP(it * 8u + 6u);
  }
  if (hp.dbg) {  // (debug only) the store acknowledgements a workgroup waits for before it retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H3P_STAMP(242);
  }
}

// Patch shape for an (N, Ho, Wo) output: maximise the fraction of the 256 tile rows that are real pixels.
static bool plan_patch(int N, int Ho, int Wo, bool nchw, HaloParams* hp) {
  int best_imgs = 0, best_th = 0, best_tw = 0;
  double best_u = 0.0;
  auto consider = [&](int imgs, int th, int tw) {
    if (imgs < 1 || th < 1 || tw < 1 || imgs * th * tw > 256) return;
    if (imgs * (th + 2) * (tw + 2) > H3_AROWS) return;
    const long groups = (N + imgs - 1) / imgs;
    const long tiles = groups * ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
    double u = (double)N * Ho * Wo / ((double)tiles * 256.0);
    if (tw % 8 == 0 && Wo % 8 == 0) u *= 1.02;  // prefer shapes that keep the NCHW stores vectorised
    // NCHW output: a channel's pixels are contiguous along a map row, so among equally full patches the WIDEST wins --
    // full-width patches store th*Wo contiguous pixels per channel (the 32x32 head level: 512-byte runs instead of the
    // 16-byte runs of an 8-wide patch, each 64-byte line shared by four workgroups)
    if (nchw) u *= 1.0 + 0.01 * (double)tw / (double)Wo;
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

// Split-K for grids that cannot fill the chip (few output pixels, long K: the 8x8 and 4x4 head levels at batch 64): the
// number of slices (1: none) and the workspace they need.  A slice owns >= 2 slabs (>= 18 k-steps: the prologue has to
// amortise) and the grid stops growing at ~one workgroup per CU.
int halo_splitk_plan(int N, int Cin, int Ho, int Wo, int Cout, bool nchw, size_t* ws_bytes) {
  // OFF by default: measured on the 8x8 head level (M = 4096, K = 4608) it ties conv_smallmap (45.2 vs 45.3 us) and loses on
  // the 4x4 level (38.9 vs 19.2 us) -- a halo workgroup costs ~23 k cycles of set-up + epilogue whatever its K, and the reducer
  // reads 4 x 128 KB of slabs alone.  Kept (and tested) for shapes with a longer K per slice: SSDK_HALO_SPLITK=1 | slices.
  static const int env = getenv("SSDK_HALO_SPLITK") ? atoi(getenv("SSDK_HALO_SPLITK")) : 0;
  if (ws_bytes) *ws_bytes = 0;
  HaloParams hp;
  if (!env || Cout < 96 || Cin < 128 || (Cin % 8) || !plan_patch(N, Ho, Wo, nchw, &hp)) return 1;
  const long tiles = (long)hp.groups * hp.tiles_y * hp.tiles_x * ((Cout + H3_BN - 1) / H3_BN);
  const int cchunks = (Cin + H3_BK - 1) / H3_BK;
  int ks = 1;
  while (ks * 2 <= cchunks / 2 && tiles * ks * 2 <= 320 && ks < 8) ks *= 2;
  if (env > 1 && env <= cchunks) ks = env;  // (forced, A/B runs)
  if (ks <= 1 || tiles > 1024) return 1;
  if (ws_bytes) *ws_bytes = 4096 + (size_t)tiles * ks * 16 * H3_THREADS * 4 * sizeof(float);
  return ks;
}

// Workspace of the halo kernel running with `ksplits` slices, whoever planned them (ssdk_conv hands a split planned for
// conv_gemm_kernel on when the halo kernel takes the layer: its slabs are laid out per HALO tile, 1.3x the GEMM's bytes on
// the 10x10 FPN levels -- round 4: until then the check was against the GEMM's need and the slabs ran past it).  0: the
// geometry does not fit this kernel's split (no patch plan, or more tiles than the 4 KiB of arrival counters hold).
size_t halo_ws_bytes(int N, int Cin, int Ho, int Wo, int Cout, bool nchw, int ksplits) {
  HaloParams hp;
  if (ksplits <= 1 || !plan_patch(N, Ho, Wo, nchw, &hp)) return 0;
  const int cch = (Cin + H3_BK - 1) / H3_BK;
  const int c_per = (cch + ksplits - 1) / ksplits;
  const int ks = (cch + c_per - 1) / c_per;
  const long tiles = (long)hp.groups * hp.tiles_y * hp.tiles_x * ((Cout + H3_BN - 1) / H3_BN);
  if (tiles > 1024) return 0;
  return 4096 + (size_t)tiles * ks * 16 * H3_THREADS * 4 * sizeof(float);
}

int launch_conv3x3_halo(const ConvParams& p, int dtype, hipStream_t stream, bool allow_underfill) {
  static const int env = getenv("SSDK_CONV3X3_HALO") ? atoi(getenv("SSDK_CONV3X3_HALO")) : 1;
  if (!env || p.k != 3 || p.stride != 1 || p.pad != 1 || (p.Cin % 8)) return 1;
  // narrow outputs (the 256 -> 36 box heads of the FPN / BiFPN levels): a quarter-full 128-channel tile still beats the
  // 128 x 64 flat-K tiles of conv_gemm_kernel when K is long (SSDK_HALO_MIN_COUT, A/B)
  constexpr int env_minco = 32;  // (round 6: the SSDK_HALO_MIN_COUT switch is gone, its A/B is settled)
  if (p.Cout < (p.Cin >= 128 ? env_minco : 96) || p.Cin < 32) return 1;
  HaloParams hp;
  hp.c = p;
  if (!plan_patch(p.N, p.Ho, p.Wo, p.out_layout == LAYOUT_NCHW, &hp)) return 1;
  hp.n_tiles = (p.Cout + H3_BN - 1) / H3_BN;
  // p.ksplits > 1 here means: the caller planned THIS kernel's split (halo_splitk_plan) and provides slabs / counters
  hp.ksplits = p.ksplits > 1 ? p.ksplits : 1;
  {
    const int cch = (p.Cin + H3_BK - 1) / H3_BK;
    hp.c_per = (cch + hp.ksplits - 1) / hp.ksplits;
    hp.ksplits = (cch + hp.c_per - 1) / hp.c_per;  // every slice owns at least one slab
  }
  hp.slabs = p.slabs;
  hp.counters = p.counters;
  auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  hp.mg_ntiles = magic(hp.n_tiles);
  hp.mg_tx = magic(hp.tiles_x);
  hp.mg_ty = magic(hp.tiles_y);
  hp.mg_ppi = magic(hp.th * hp.tw);
  hp.mg_tw = magic(hp.tw);
  hp.mg_hw2 = magic(hp.tw + 2);
  hp.mg_halo = magic((hp.th + 2) * (hp.tw + 2));
  hp.mg_ks = magic(hp.ksplits);
  hp.mg_wo = magic(p.Wo);
  const long xb = (long)p.N * p.H * p.W * p.Cin * 2, wb = (long)p.Cout * 9 * p.Cin * 2;
  if (xb >= 0xfffffff0l || wb >= 0xfffffff0l) return 1;  // 32-bit buffer offsets
  hp.x_bytes = (unsigned)xb;
  hp.w_bytes = (unsigned)wb;
  const long tiles = (long)hp.groups * hp.tiles_y * hp.tiles_x * hp.n_tiles * hp.ksplits;
  if (env != 2 && !allow_underfill && tiles < 96) return 1;  // too few tiles to fill the chip: the split-K path is faster
  if (tiles >= (1l << 26)) return 1;  // keeps tile-id * divisor < 2^32 for the magic divisions
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
  static const int dbg = getenv("SSDK_H3_DBG") ? atoi(getenv("SSDK_H3_DBG")) : 0;
  hp.dbg = nullptr;
  hp.dbg_wg = 0;
  hp.tq = hp.tr = 0;
  // persistent form: one workgroup per CU walks tiles / CUs tiles, epilogue from registers (SSDK_HALO_PERSIST=0: first form)
  static const int persist = getenv("SSDK_HALO_PERSIST") ? atoi(getenv("SSDK_HALO_PERSIST")) : 1;
  const bool nchw_out = p.out_layout == LAYOUT_NCHW;
  const long hw_out = (long)p.Ho * p.Wo;
  const long yb = nchw_out ? (long)p.N * p.split * hw_out * 2 : (long)p.N * hw_out * p.Cout * 2;
  const long y2b = nchw_out ? (long)p.N * (p.Cout - p.split) * hw_out * 2 : 0;
  const long rb = !p.res ? 0 : ((p.res_mode & 1) ? (long)p.N * (p.Ho >> 1) * (p.Wo >> 1) * p.Cout * 2 : (long)p.N * hw_out * p.Cout * 2);
  if (persist && hp.ksplits == 1 && (nchw_out ? (hp.tw % 4 == 0 && p.Wo % 4 == 0) : (hp.tw % 4 == 0 && p.Cout % 4 == 0)) &&
      yb < 0xfffffff0l && y2b < 0xfffffff0l && rb < 0xfffffff0l && (!nchw_out || !p.res) && (p.split == p.Cout || p.y2)) {
    hp.y_bytes = (unsigned)yb;
    hp.y2_bytes = (unsigned)y2b;
    hp.res_bytes = (unsigned)rb;
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    // persist = tiles per workgroup at most (1: only the register epilogue; large: one workgroup per CU walks tiles / CUs tiles)
    long want = (tiles + persist - 1) / persist;
    if (want < cus) want = tiles < cus ? tiles : cus;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > tiles ? tiles : want));
    hp.tq = (unsigned)(tiles / grid);
    hp.tr = (unsigned)(tiles % grid);
    static bool attr2_done = false;
    if (!attr2_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halop_kernel<SSDK_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halop_kernel<SSDK_BF16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halop_kernel<SSDK_F16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halop_kernel<SSDK_F16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
      attr2_done = true;
    }
    if (dbg) {
      static const int dbg_wg = getenv("SSDK_H3_DBG_WG") ? atoi(getenv("SSDK_H3_DBG_WG")) : 0;
      hp.dbg_wg = dbg_wg < 0 ? grid - 1u : (unsigned)dbg_wg;
      (void)hipMalloc((void**)&hp.dbg, 64 * 4 * sizeof(long long));
      (void)hipMemsetAsync(hp.dbg, 0, 64 * 4 * sizeof(long long), stream);
    }
    if (di == 0) {
      if (nchw_out) hipLaunchKernelGGL((conv3x3_halop_kernel<SSDK_BF16, false>), dim3(grid), dim3(H3_THREADS), H3_LDS, stream, hp);
      else hipLaunchKernelGGL((conv3x3_halop_kernel<SSDK_BF16, true>), dim3(grid), dim3(H3_THREADS), H3_LDS, stream, hp);
    } else {
      if (nchw_out) hipLaunchKernelGGL((conv3x3_halop_kernel<SSDK_F16, false>), dim3(grid), dim3(H3_THREADS), H3_LDS, stream, hp);
      else hipLaunchKernelGGL((conv3x3_halop_kernel<SSDK_F16, true>), dim3(grid), dim3(H3_THREADS), H3_LDS, stream, hp);
    }
    if (dbg) {  // debug only: synchronises and prints the phase times of one workgroup (lane 0 of wave 0)
      long long h[64 * 4];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(h, hp.dbg, sizeof(h), hipMemcpyDeviceToHost);
      (void)hipFree(hp.dbg);
      static int printed = 0;
      if (printed++ < dbg) {
        fprintf(stderr, "[h3p dbg] %s Cin %d Cout %d %dx%d grid %u tiles %ld: set-up + first loads %lld\n", nchw_out ? "nchw" : "nhwc", p.Cin,
                p.Cout, p.Ho, p.Wo, grid, tiles, h[241] - h[240]);
        for (int i = 0; i < 30 && h[i * 8]; ++i)
          fprintf(stderr, "[h3p dbg] tile %2d: loop %6lld | drain+barrier %5lld | consts+next prologue %5lld | math+stores %5lld | wait %5lld | scale/bias+barrier %5lld\n",
                  i, h[i * 8 + 1] - h[i * 8], h[i * 8 + 2] - h[i * 8 + 1], h[i * 8 + 3] - h[i * 8 + 2], h[i * 8 + 4] - h[i * 8 + 3],
                  h[i * 8 + 5] - h[i * 8 + 4], h[i * 8 + 6] - h[i * 8 + 5]);
        fprintf(stderr, "[h3p dbg] store acknowledgements after the last tile: %lld\n", h[242] - h[(hp.tq + (hp.dbg_wg < hp.tr ? 1 : 0) - 1) * 8 + 6]);
      }
    }
    return check_launch("conv3x3_halo_kernel");
  }
  if (dbg) {
    static const int dbg_wg = getenv("SSDK_H3_DBG_WG") ? atoi(getenv("SSDK_H3_DBG_WG")) : 0;
    hp.dbg_wg = dbg_wg < 0 ? (unsigned)(tiles - 1) : (unsigned)dbg_wg;
    (void)hipMalloc((void**)&hp.dbg, 64 * 4 * sizeof(long long));
    (void)hipMemsetAsync(hp.dbg, 0, 64 * 4 * sizeof(long long), stream);
  }
  if (di == 0)
    hipLaunchKernelGGL((conv3x3_halo_kernel<SSDK_BF16>), dim3((unsigned)tiles), dim3(H3_THREADS), H3_LDS, stream, hp);
  else
    hipLaunchKernelGGL((conv3x3_halo_kernel<SSDK_F16>), dim3((unsigned)tiles), dim3(H3_THREADS), H3_LDS, stream, hp);
  if (dbg) {  // debug only: synchronises and prints the per-step phase times of workgroup 0 / wave 0
    long long h[64 * 4];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, hp.dbg, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(hp.dbg);
    static int printed = 0;
    if (printed++ < dbg) {
      fprintf(stderr, "[h3 dbg] step: issue+reads | waits+barrier | mfma+barrier | (next step start - this)\n");
      fprintf(stderr, "[h3 dbg] setup %lld | first loads %lld | loop %lld | epilogue math+stage %lld | stores %lld\n",
              h[241] - h[240], h[242] - h[241], h[243] - h[242], h[244] - h[243], h[245] - h[244]);
      for (int i = 0; i < 60 && h[i * 4] && dbg > 1; ++i)
        fprintf(stderr, "[h3 dbg] %2d: %6lld %6lld %6lld  total %6lld\n", i, h[i * 4 + 1] - h[i * 4], h[i * 4 + 2] - h[i * 4 + 1],
                h[i * 4 + 3] - h[i * 4 + 2], (i < 63 && h[(i + 1) * 4]) ? h[(i + 1) * 4] - h[i * 4] : 0ll);
    }
  }
  return check_launch("conv3x3_halo_kernel");
}

}  // namespace ssdk
