// ssdk_gemmp.hip -- 1x1 convolution (+ folded BN, activation, residual) with a LONG K as a persistent NT GEMM on the matrix
// cores (gfx950): the bottleneck / downsample / lateral 1x1 layers of the ResNet and RegNet backbones behind the FPN / BiFPN
// configurations with Cin >= 256 (reference nets/resnet.py:41-56 through torchvision's Bottleneck, ssds/fpn.py:58-101), e.g.
// 256 -> 1024 on 40x40 maps and 2048 -> 512 on 20x20 maps at batch 32.
//
// Why: on conv_gemm_kernel / conv_gemm256_kernel these layers ran at 1.5 - 2 x the time of the vendor library's plain GEMM
// (tools/gemm_ceiling_probe.py, round 6: 256 -> 1024 @40x40 94 us against 52, 2048 -> 512 @20x20 67 against 32), although
// nothing is fused there.  The loop below is conv3x3_halo_kernel's (ssdk_conv3x3.hip) with the taps taken out:
//   * tile 256 pixels x 128 channels x 64 k, 8 waves as 4 (M) x 2 (N), 4 x 4 accumulator fragments of v_mfma_f32_16x16x32;
//     both operands K-contiguous in memory (NHWC activations, KRSC = [Cout][Cin] weights): 128-byte LDS rows, three stages of
//     (32 KiB pixels + 16 KiB weights) filled by buffer_load ... lds two k-steps ahead, hand-counted vmcnt, raw barriers, the
//     two wave groups one phase apart (one reads its fragments while the other holds the matrix pipe);
//   * the bank swizzle (16-byte chunk + row) & 7 on the source-address side, as in the halo kernel;
//   * one workgroup per CU walks a contiguous range of tiles, the n-tiles of a pixel tile first (the second one finds the
//     pixels in L2); the next tile's first two stages are requested before the epilogue of the current one;
//   * epilogue from the accumulator registers: the loader places weight row n0 + (q & 64) + (q & 15) * 4 + ((q >> 4) & 3) at
//     LDS row q, so a lane's four accumulator columns are four CONSECUTIVE channels -- 8 contiguous bytes of a pixel, sixteen
//     lanes = 128 bytes; scale / bias / activation / residual (same or half resolution, activation before or after the add) on
//     those pieces, no LDS image, no barrier.
// Roofline: max(HBM time of input + output (+ residual), FLOPs at the dense MFMA peak); DESIGN.md 4.5 has the table.
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int GP_THREADS = 512, GP_BM = 256, GP_BN = 128, GP_BK = 64;
constexpr int GP_A_BYTES = GP_BM * 128;               // 32768
constexpr int GP_B_BYTES = GP_BN * 128;               // 16384
constexpr int GP_STAGE = GP_A_BYTES + GP_B_BYTES;     // 49152
constexpr int GP_LDS = 3 * GP_STAGE;                  // 147456

struct GemmpParams {
  ConvParams c;
  int n_tiles;                                   // ceil(Cout / 128)
  unsigned tiles;                                // m-tiles * n_tiles
  unsigned x_bytes, w_bytes, y_bytes, res_bytes;  // buffer-descriptor ranges (tensors < 4 GiB)
  unsigned mg_ntiles, mg_wo, mg_hwo;              // ceil(2^32 / d)
  unsigned tq, tr;                               // workgroup g owns tiles [g*tq + min(g, tr), +tq + (g < tr))
  long long* dbg;                                // SSDK_GP_DBG=1: cycle stamps of one workgroup (lane 0 of wave 0)
  unsigned dbg_wg;
};

__device__ __forceinline__ u32 gp_fdiv(u32 n, u32 d, u32 M) {  // n / d for n*d < 2^32 with M = ceil(2^32 / d); d == 1 as a select
  const u32 q = __umulhi(n, M);
  const u32 one = (u32)-(int)(d == 1u);
  return (n & one) | (q & ~one);
}

#define GP_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
typedef unsigned int gp_v2u __attribute__((ext_vector_type(2)));

// y = act(acc * scale + bias) -> 16 bit, four accumulator rows of one column (the arithmetic of epilogue4, ssdk_conv_common.h)
template <int DT, bool SIG>
__device__ __forceinline__ uint2 gp_epilogue4(const f32x4 acc, float sc, float bi, float lo, float hi, int mode, bool clampy) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r] * sc + bi;
  if constexpr (SIG) {  // (workgroup-uniform mode: one activation per layer here)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
      v[r] = mode == 1 ? sg : v[r] * sg;
    }
  }
  if (clampy) {  // (a linear layer keeps its NaNs: v_min / v_max would drop them)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], lo), hi);
  }
  return make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
}

template <int DT>
__global__ __launch_bounds__(GP_THREADS) void conv_gemmp_kernel(const GemmpParams gp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvParams& p = gp.c;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 wm = wave >> 1, wn = wave & 1u;

  const u32 nwg = gridDim.x, id = blockIdx.x;
  const u32 q8 = nwg >> 3, r8 = nwg & 7u, xcd = id & 7u;
  const u32 lin = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + (id >> 3);  // neighbours in lin share an XCD
  const u32 t_begin = lin * gp.tq + (lin < gp.tr ? lin : gp.tr);
  const u32 t_cnt = gp.tq + (lin < gp.tr ? 1u : 0u);
#define GP_STAMP(idx)                                                                                        \
  do {                                                                                                       \
    if (gp.dbg && lin == gp.dbg_wg && tid == 0 && (idx) < 256) gp.dbg[(idx)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
  GP_STAMP(240);

  const int K = p.Cin;
  const int nk = (K + GP_BK - 1) / GP_BK;
  const u32 M = (u32)p.M, HWo = (u32)(p.Ho * p.Wo), Wo = (u32)p.Wo;

  const u32 lrow = lane >> 3;
  const u32 lchunk = ((lane & 7u) - (lane >> 3)) & 7u;  // logical 16-byte chunk of this loader lane's slot (see the halo kernel)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, gp.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, gp.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, gp.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x), 0, gp.res_bytes, 0x00020000);
  constexpr u32 OOB = 0xfffffff0u;
  const int lci = (int)lchunk * 8;
  const int tail = K - (nk - 1) * GP_BK;          // k of the last step (1..64)
  const u32 tailmask = lci < tail ? 0u : OOB;     // OR-ed into the offsets of the last step: k >= K -> zeros
  const bool tail_half = tail <= 32;              // the last step holds <= 32 k: one k-substep is enough

  const u32 fr = lane & 15u, fg = lane >> 4;
  // fragment i / j of this lane: rows wm*64 + 16 i + fr / wn*64 + 16 j + fr, chunk (fg + row) & 7 -- the same for every i, j
  const u32 a_ad0 = (wm * 64u + fr) * 128u + (((fg + fr) & 7u) << 4);
  const u32 b_ad0 = (u32)GP_A_BYTES + (wn * 64u + fr) * 128u + (((fg + fr) & 7u) << 4);

  // ---- per-tile state ------------------------------------------------------------------------------------------------------
  u32 m0 = 0, n0 = 0, cur_mt = 0xffffffffu;
  u32 a_vo[4], b_vo[2];
  auto set_tile = [&](u32 tile) {
    const u32 mt = gp_fdiv(tile, (u32)gp.n_tiles, gp.mg_ntiles);
    const u32 nt = tile - mt * (u32)gp.n_tiles;
    n0 = nt * GP_BN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32 q = ((u32)j * 8u + wave) * 8u + lrow;  // LDS row of this lane's chunk
      const u32 n = n0 + (q & 64u) + (q & 15u) * 4u + ((q >> 4) & 3u);
      b_vo[j] = n < (u32)p.Cout ? (n * (u32)K + (u32)lci) * 2u : OOB;
    }
    if (mt == cur_mt) return;  // the next n-tile of the same pixels
    cur_mt = mt;
    m0 = mt * GP_BM;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const u32 m = m0 + ((u32)t * 8u + wave) * 8u + lrow;
      u32 pix = m;
      if (p.stride != 1) {  // (uniform) a stride-2 1x1 reads every second pixel of every second row
        const u32 b = gp_fdiv(m, HWo, gp.mg_hwo), r = m - b * HWo, oy = gp_fdiv(r, Wo, gp.mg_wo), ox = r - oy * Wo;
        pix = (b * (u32)p.H + oy * 2u) * (u32)p.W + ox * 2u;
      }
      a_vo[t] = m < M ? (pix * (u32)K + (u32)lci) * 2u : OOB;
    }
  };
  auto load_stage = [&](int stage, int kk) {  // pixels and weights of k-step kk -> stage; past the end: a harmless re-read of step 0
    const bool live = kk < nk;
    const u32 tm = (live && kk == nk - 1) ? tailmask : 0u;
    const int soff = live ? kk * GP_BK * 2 : 0;
    lds_u8* dst = (lds_u8*)(smem + stage * GP_STAGE + wave * 1024u);
#pragma unroll
    for (int t = 0; t < 4; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + t * 8192, 16, a_vo[t] | tm, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst + GP_A_BYTES, 16, b_vo[0] | tm, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, dst + GP_A_BYTES + 8192, 16, b_vo[1] | tm, soff, 0, 0);
  };
  float ld_sc[4], ld_bi[4];
  auto load_scbi = [&]() {  // (unconditional loads from clamped indices)
    const float* scp = p.scale ? p.scale : p.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 n = n0 + wn * 64u + fr * 4u + (u32)j;
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

  const bool g1 = wave >= 4u;
  const bool sig = act_is_sig(p.act), clampy = act_is_clamp(p.act);
  const ActSel as = act_sel(p.act);
  const u32 rsh = (u32)(p.res_mode & 1);
  const float post_lo = (p.post == SSDK_ACT_RELU || p.post == SSDK_ACT_RELU6) ? 0.f : -__builtin_inff();
  const float post_hi = p.post == SSDK_ACT_RELU6 ? 6.f : __builtin_inff();
  const bool post_clamp = p.post == SSDK_ACT_RELU || p.post == SSDK_ACT_RELU6;

  // residual pieces of this lane's 16 rows x 4 channels: requested in the shadow of the tile's LAST 32 MFMAs (no counted wait
  // follows them; the drain behind the loop covers their round trip), consumed by the epilogue.  Round 6, first version: loaded
  // inside the epilogue, four dependent HBM round trips per tile -- 256 -> 1024 @40x40 spent 22 k of its 28 k cycles per tile there
  const int rmode = !p.res ? 0 : (rsh ? 2 : 1);
  gp_v2u rv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[i][r] = gp_v2u{0u, 0u};
  auto issue_residual = [&]() {
    const u32 nb = n0 + wn * 64u + fr * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const u32 m = m0 + wm * 64u + (u32)i * 16u + fg * 4u + (u32)r;
        const bool ok = (m < M) & (nb < (u32)p.Cout);
        u32 rpix = m;
        if (rmode == 2) {  // (uniform) half resolution, nearest x2
          const u32 b = gp_fdiv(m, HWo, gp.mg_hwo), rem = m - b * HWo, oy = gp_fdiv(rem, Wo, gp.mg_wo), ox = rem - oy * Wo;
          rpix = (b * ((u32)p.Ho >> 1) + (oy >> 1)) * ((u32)p.Wo >> 1) + (ox >> 1);
        }
        rv[i][r] = __builtin_amdgcn_raw_buffer_load_b64(rr_, (int)(ok ? (rpix * (u32)p.Cout + nb) * 2u : OOB), 0, 0);
      }
  };

  set_tile(t_begin);
  load_stage(0, 0);
  load_stage(1, 1);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  GP_WAIT(6);  // stage 0 has landed (loads complete in order)
  load_scbi();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  GP_STAMP(241);

  for (u32 it = 0; it < t_cnt; ++it) {
    GP_STAMP(it * 8u + 0u);
    // ---- main loop: two phases per k-step (R: loads two steps ahead + all fragment reads + counted wait; M: 32 MFMAs), the two
    //      wave groups one phase apart (waves w and w + 4 share a SIMD): conv3x3_halo_kernel's schedule ------------------------
    if (g1) __builtin_amdgcn_s_barrier();
    for (int kk = 0; kk < nk;) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {  // stage of step kk is u (kk starts at a multiple of 3)
        if (kk >= nk) break;
        const bool half = tail_half && kk == nk - 1;
        load_stage((u + 2) % 3, kk + 2);
        u32x4 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const u32x4*>(smem + u * GP_STAGE + b_ad0 + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa0[i] = *reinterpret_cast<const u32x4*>(smem + u * GP_STAGE + a_ad0 + i * 2048);
        if (!half) {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const u32x4*>(smem + u * GP_STAGE + (b_ad0 ^ 64u) + j * 2048);
#pragma unroll
          for (int i = 0; i < 4; ++i) fa1[i] = *reinterpret_cast<const u32x4*>(smem + u * GP_STAGE + (a_ad0 ^ 64u) + i * 2048);
        }
        GP_WAIT(6);  // the stage of step kk + 1 has landed; the six requests of this step stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (rmode != 0 && kk == nk - 1) issue_residual();
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
        ++kk;
      }
    }
    if (!g1) __builtin_amdgcn_s_barrier();  // re-align the groups
    GP_STAMP(it * 8u + 1u);
    GP_WAIT(0);                             // (the loop's harmless loads past the end; the scale / bias loads of this tile)
    __builtin_amdgcn_s_barrier();
    GP_STAMP(it * 8u + 2u);

    // ---- tile boundary -----------------------------------------------------------------------------------------------------------
    float e_sc[4], e_bi[4];
    const u32 nb = n0 + wn * 64u + fr * 4u;  // this lane's four consecutive channels (Cout % 4 == 0: all inside or all outside)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e_sc[j] = (nb < (u32)p.Cout && p.scale) ? ld_sc[j] : 1.f;
      e_bi[j] = nb < (u32)p.Cout ? ld_bi[j] : 0.f;
      asm volatile("" : "+v"(e_sc[j]), "+v"(e_bi[j]));  // the loads are consumed HERE: no compiler wait behind the LDS-DMA requests below
    }
    if (rmode != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(rv[i][r]));
    }
    const u32 cm0 = m0;
    const bool more = it + 1u < t_cnt;
    if (more) {
      set_tile(t_begin + it + 1u);
      load_stage(0, 0);  // land behind the arithmetic below; the stages are free: every wave is past its last fragment read
      load_stage(1, 1);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    GP_STAMP(it * 8u + 3u);
    // one accumulator row fragment (16 pixels x the lane's four channels) at a time: arithmetic, then its four 8-byte stores
    // (branch-free: a piece that must not be stored carries an out-of-range buffer offset)
    auto fragments = [&](auto SIG, auto RES) {  // RES: 0 no residual, 1 a residual (fetched behind the last k-step)
      constexpr int R = decltype(RES)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = gp_epilogue4<DT, decltype(SIG)::value>(acc[i][j], e_sc[j], e_bi[j], as.lo, as.hi, as.mode, clampy);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        uint2 o[4];  // row r: channels nb .. nb+3
        o[0] = make_uint2(__builtin_amdgcn_perm(h[1].x, h[0].x, 0x05040100u), __builtin_amdgcn_perm(h[3].x, h[2].x, 0x05040100u));
        o[1] = make_uint2(__builtin_amdgcn_perm(h[1].x, h[0].x, 0x07060302u), __builtin_amdgcn_perm(h[3].x, h[2].x, 0x07060302u));
        o[2] = make_uint2(__builtin_amdgcn_perm(h[1].y, h[0].y, 0x05040100u), __builtin_amdgcn_perm(h[3].y, h[2].y, 0x05040100u));
        o[3] = make_uint2(__builtin_amdgcn_perm(h[1].y, h[0].y, 0x07060302u), __builtin_amdgcn_perm(h[3].y, h[2].y, 0x07060302u));
        const u32 mrow = cm0 + wm * 64u + (u32)i * 16u + fg * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32 m = mrow + (u32)r;
          const bool ok = (m < M) & (nb < (u32)p.Cout);
          uint2 v = o[r];
          if constexpr (R != 0) {  // (values already rounded to 16 bit, like the framework's tensor add)
            const gp_v2u rq = rv[i][r];
            auto addc = [&](u32 a, u32 b) {
              const float s = bits16_to_f32<DT>(a) + bits16_to_f32<DT>(b);
              return post_clamp ? __builtin_fminf(__builtin_fmaxf(s, post_lo), post_hi) : s;
            };
            v.x = pack2_16<DT>(addc(v.x & 0xffffu, rq.x & 0xffffu), addc(v.x >> 16, rq.x >> 16));
            v.y = pack2_16<DT>(addc(v.y & 0xffffu, rq.y & 0xffffu), addc(v.y >> 16, rq.y >> 16));
          }
          __builtin_amdgcn_raw_buffer_store_b64(gp_v2u{v.x, v.y}, yr, (int)(ok ? (m * (u32)p.Cout + nb) * 2u : OOB), 0, 0);
        }
      }
    };
    {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      if (sig) {
        if (rmode == 0) fragments(std::true_type{}, I0{});
        else fragments(std::true_type{}, I1{});
      } else {
        if (rmode == 0) fragments(std::false_type{}, I0{});
        else fragments(std::false_type{}, I1{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    GP_STAMP(it * 8u + 4u);
    // loads complete in order among loads: <= 6 operations outstanding means stage 0 of the next tile has landed (and all but six
    // of the stores above are acknowledged -- the first k-step's counted wait would ask for that anyway)
    if (more) GP_WAIT(6);
    __builtin_amdgcn_sched_barrier(0);
    GP_STAMP(it * 8u + 5u);
    if (more) load_scbi();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // every wave's pieces of the next tile's stage 0 have landed (each waited for its own above)
    GP_STAMP(it * 8u + 6u);
  }
  if (gp.dbg) {  // (debug only) the store acknowledgements a workgroup waits for before it retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_STAMP(242);
  }
}

// 1: not one of this kernel's layers (the caller goes on), otherwise the launch status
int launch_conv_gemmp(const ConvParams& p, int dtype, hipStream_t stream) {
  static const int env = getenv("SSDK_GEMMP") ? atoi(getenv("SSDK_GEMMP")) : 1;
  if (!env || p.k != 1 || p.pad != 0 || (p.stride != 1 && p.stride != 2)) return 1;
  if (p.in_layout != LAYOUT_NHWC || p.out_layout != LAYOUT_NHWC || p.split != p.Cout || p.ksplits != 1) return 1;
  // long K, channels for at least one full tile column, pixels for at least half the chip (fewer: the split-K kernels)
  // (measured with Cin >= 128 and >= 64 admitted, tools/run/r06_s21.sh: 128 -> 512 @80x80 146 us against pwflow's 140, 64 -> 256 @160x160
  //  235 against 180 -- one or two k-steps per tile leave the tile boundary alone: the short-K layers stay on pwflow_kernel)
  constexpr int min_cin = 256;
  if ((p.Cin % 8) || p.Cin < min_cin || (p.Cout % 4) || p.Cout < 128) return 1;
  if ((p.res_mode & 1) && ((p.Ho | p.Wo) & 1)) return 1;
  if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y | (uintptr_t)p.res) & 15) return 1;
  const long M = p.M;
  const long xb = (long)p.N * p.H * p.W * p.Cin * 2, wb = (long)p.Cout * p.Cin * 2, yb = M * p.Cout * 2;
  const long rb = !p.res ? 0 : ((p.res_mode & 1) ? (long)p.N * (p.Ho >> 1) * (p.Wo >> 1) * p.Cout * 2 : yb);
  if (xb >= 0xfffffff0l || wb >= 0xfffffff0l || yb >= 0xfffffff0l || rb >= 0xfffffff0l || M >= (1l << 30)) return 1;
  GemmpParams gp;
  gp.c = p;
  gp.n_tiles = (p.Cout + GP_BN - 1) / GP_BN;
  const long m_tiles = (M + GP_BM - 1) / GP_BM;
  const long tiles = m_tiles * gp.n_tiles;
  if (tiles < 96 || tiles >= (1l << 24)) return 1;
  {  // whole tiles per workgroup: a launch of a little more than one round of tiles (294 on 256 CUs: the 672 -> 672 layers of
     // RegNetX-800MF at 28x28, batch 16) leaves half the chip idle in its second round -- the 128-row tiles of conv_gemm_kernel pack better
    int ncu = 256;
    const long rounds = (tiles + ncu - 1) / ncu;
    if (tiles > ncu && (double)tiles / (double)(rounds * ncu) < 0.7) return 1;
  }
  gp.tiles = (unsigned)tiles;
  gp.x_bytes = (unsigned)xb;
  gp.w_bytes = (unsigned)wb;
  gp.y_bytes = (unsigned)yb;
  gp.res_bytes = (unsigned)rb;
  auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); };
  gp.mg_ntiles = magic(gp.n_tiles);
  gp.mg_wo = magic(p.Wo);
  gp.mg_hwo = magic((long)p.Ho * p.Wo);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // one workgroup per CU walks tiles / CUs tiles (measured against at most 4 / 2 / 1 tiles per workgroup under the level lanes of the
  // FPN / BiFPN plans, tools/run/r06_s30.sh: 3 417 - 3 430 vs 3 374 / 3 372 - 3 385 / 3 387 and 2 400 - 2 405 vs 2 403 / 2 351 - 2 371 / 2 393 img/s)
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  gp.tq = (unsigned)(tiles / grid);
  gp.tr = (unsigned)(tiles % grid);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemmp_kernel<SSDK_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemmp_kernel<SSDK_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS);
    attr_done = true;
  }
  static const int dbg = getenv("SSDK_GP_DBG") ? atoi(getenv("SSDK_GP_DBG")) : 0;
  gp.dbg = nullptr;
  gp.dbg_wg = 0;
  if (dbg) {
    static const int dbg_wg = getenv("SSDK_GP_DBG_WG") ? atoi(getenv("SSDK_GP_DBG_WG")) : 0;
    gp.dbg_wg = dbg_wg < 0 ? grid - 1u : (unsigned)dbg_wg;
    (void)hipMalloc((void**)&gp.dbg, 64 * 4 * sizeof(long long));
    (void)hipMemsetAsync(gp.dbg, 0, 64 * 4 * sizeof(long long), stream);
  }
  if (dtype == SSDK_BF16) hipLaunchKernelGGL((conv_gemmp_kernel<SSDK_BF16>), dim3(grid), dim3(GP_THREADS), GP_LDS, stream, gp);
  else hipLaunchKernelGGL((conv_gemmp_kernel<SSDK_F16>), dim3(grid), dim3(GP_THREADS), GP_LDS, stream, gp);
  if (dbg) {  // debug only: synchronises and prints the phase times of one workgroup
    long long h[64 * 4];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, gp.dbg, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(gp.dbg);
    static int printed = 0;
    if (printed++ < dbg) {
      fprintf(stderr, "[gemmp dbg] Cin %d Cout %d M %ld stride %d grid %u tiles %ld: set-up + first loads %lld\n", p.Cin, p.Cout, M, p.stride,
              grid, tiles, h[241] - h[240]);
      for (int i = 0; i < 30 && h[i * 8]; ++i)
        fprintf(stderr, "[gemmp dbg] tile %2d: loop %6lld | drain+barrier %5lld | consts+next prologue %5lld | math+stores %5lld | wait %5lld | scale/bias+barrier %5lld\n",
                i, h[i * 8 + 1] - h[i * 8], h[i * 8 + 2] - h[i * 8 + 1], h[i * 8 + 3] - h[i * 8 + 2], h[i * 8 + 4] - h[i * 8 + 3],
                h[i * 8 + 5] - h[i * 8 + 4], h[i * 8 + 6] - h[i * 8 + 5]);
    }
  }
  return check_launch("conv_gemmp_kernel");
}

}  // namespace ssdk
