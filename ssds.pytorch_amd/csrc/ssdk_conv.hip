// ssdk_conv.hip -- fused convolution + folded BatchNorm + activation for the detector on gfx950.
//
// Replaces the Conv2d -> BatchNorm2d -> ReLU chains of the reference (basic_layers.py:5-57, the
// torchvision MobileNetV2 blocks used by nets/mobilenet.py:56-99) and the bare head convs (ssd.py:100-103,
// fpn.py:10-18), which the reference runs as separate cuDNN/ATen launches.  Three kernels:
//
//   conv_gemm_kernel   dense 1x1 / 3x3 (stride 1|2) convolution as an implicit GEMM on the matrix cores:
//                      C[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b, oy, ox), k = (ky, kx, ci), NHWC
//                      activations, KRSC weights.  128 x BN x 32 tiles, 4 waves, v_mfma_f32_16x16x32
//                      (bf16 / f16 in, fp32 accumulate), register-staged 16-byte global loads into a
//                      double-buffered, XOR-swizzled LDS image (conflict-free ds_read_b128 fragments), one
//                      barrier per k-step.  Epilogue: per-channel scale/bias (folded BN or conv bias),
//                      activation, optional residual add, tile staged through LDS and written with
//                      16-byte coalesced stores either NHWC (next layer) or NCHW (the [B, A*C, H, W] head
//                      layout decode consumes), optionally split over two output tensors (loc | conf of
//                      one SSD level share one GEMM: N = A*(4+C)).
//   dwconv3x3_kernel   depthwise 3x3 (stride 1|2) + scale/bias + activation, NHWC, 8 channels x 4 pixels
//                      per lane, 16-byte loads -- HBM-bound.
//   conv_first_kernel  the 3-channel stem (3x3, stride 2) from the NCHW/NHWC image to NHWC, weights in
//                      SGPRs (uniform scalar loads) -- HBM-bound.
//
// Layout contract (host side: ssds/modeling/layers/fused_conv.py packs once per model):
//   x      NHWC [N][H][W][Cin]   (torch channels_last)            dtype bf16 | f16
//   w      KRSC [Cout][kh][kw][Cin/groups]                         same dtype (fp32 for conv_first)
//   scale, bias  fp32 [Cout]  (scale may be NULL = 1)
//   y      NHWC [N][Ho][Wo][Cout] or NCHW [N][Cout][Ho][Wo]        same dtype
#include "ssdk_conv_common.h"
#include "ssdk_ctx.h"

namespace ssdk {

int launch_conv_smallmap(const ConvParams& p, int dtype, hipStream_t stream);  // ssdk_smallmap.hip: 0 launched, 1 not its layer

// 16-byte chunk swizzle inside a 64-byte LDS row: makes the 16-lane groups of ds_read_b128 conflict free
// (rows r, r+4, r+8, r+12 share the same bank phase; S permutes their chunk index).
__device__ __forceinline__ u32 swz(u32 row, u32 chunk) {
  const u32 S = (0x1230u >> (((row >> 2) & 3u) * 4u)) & 3u;  // [0,3,2,1]
  return chunk ^ S;
}

constexpr int kConvThreads = 256;
constexpr int BM = 128, BK = 32;

// RES: an instance of its own for layers with a residual (NHWC, Cout % 8 == 0): the store loop requests the residual vectors
// four at a time.  (As part of the one kernel the extra registers cost every layer a wave per SIMD: measured, reverted.)
template <int DT, int WAVES_M, int WAVES_N, int FM, int FN, bool SPLITK = false, bool RES = false>
__global__ __launch_bounds__(kConvThreads) void conv_gemm_kernel(const ConvParams p) {
  constexpr int BN = WAVES_N * FN * 16;
  static_assert(WAVES_M * FM * 16 == BM, "tile");
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C_BYTES = BM * (BN + 8) * 2;
  constexpr int LDS_BYTES = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 wm = wave / WAVES_N, wn = wave % WAVES_N;
  const u32 m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int Cin = p.Cin, H = p.H, W = p.W;

  // ---- loader roles: 16-byte chunk (row, c); A: rows tid/4 and tid/4+64; B: rows tid/4 (+64) ----------
  const u32 lc = tid & 3u, lr = tid >> 2;
  long a_base[2];
  u32 a_mask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32 m = m0 + lr + i * 64;
    u32 mask = 0;
    long base = 0;
    if (m < (u32)p.M) {
      const u32 hw = (u32)(p.Ho * p.Wo);
      const u32 b = m / hw, r = m % hw;
      const int oy = (int)(r / (u32)p.Wo), ox = (int)(r % (u32)p.Wo);
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      base = (((long)b * H + iy0) * W + ix0) * Cin;
      for (int ky = 0; ky < p.k; ++ky)
        for (int kx = 0; kx < p.k; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W) mask |= 1u << (ky * p.k + kx);
    }
    a_base[i] = base;
    a_mask[i] = mask;
  }
  constexpr int B_PER = (BN * 4 + kConvThreads - 1) / kConvThreads;  // chunks of B per thread
  const int KK = p.k * p.k;
  long w_base[B_PER];
  bool w_ok[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const u32 row = lr + i * 64;
    w_ok[i] = row < (u32)BN && n0 + row < (u32)p.Cout;
    w_base[i] = (long)(n0 + row) * KK * Cin;
  }

  // k-tiles are visited tap-major, channel-chunk-minor; the running state below replaces the per-tile
  // divisions (kt / cin_chunks, kpos / k ...) that used to cost ~90 VALU + ~90 SALU per k-step.
  const int kt_begin = SPLITK ? (int)blockIdx.z * p.kt_per : 0;
  const int kt_end = (SPLITK && kt_begin + p.kt_per < p.KT) ? kt_begin + p.kt_per : p.KT;
  int t_kpos = kt_begin / p.cin_chunks, t_cc = kt_begin % p.cin_chunks;
  int t_ky = t_kpos / p.k, t_kx = t_kpos % p.k;
  u32x4 ra[2], rb[B_PER];
  auto load_tile = [&]() {  // loads the tile of the current state, then advances the state
    const int ci = t_cc * BK + (int)lc * 8;
    const bool cok = ci < Cin;
    const int tap_ci = (t_ky * W + t_kx) * Cin + ci;
    const int w_ci = t_kpos * Cin + ci;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (cok && ((a_mask[i] >> t_kpos) & 1u)) v = *reinterpret_cast<const u32x4*>((const u16*)p.x + (a_base[i] + tap_ci));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (w_ok[i] && cok) v = *reinterpret_cast<const u32x4*>((const u16*)p.w + (w_base[i] + w_ci));
      rb[i] = v;
    }
    if (++t_cc == p.cin_chunks) {
      t_cc = 0;
      ++t_kpos;
      if (++t_kx == p.k) {
        t_kx = 0;
        ++t_ky;
      }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32 row = lr + i * 64;
      *reinterpret_cast<u32x4*>(sA + row * 64 + swz(row, lc) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const u32 row = lr + i * 64;
      if (row < (u32)BN) *reinterpret_cast<u32x4*>(sB + row * 64 + swz(row, lc) * 16) = rb[i];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const u32 fr = lane & 15u, fg = lane >> 4;
  // per-column scale / bias of the epilogue: requested FIRST, unconditionally, from clamped indices (selects in the epilogue).
  // Loaded inside the epilogue's column loop under `if (n < Cout)` they were FN dependent memory round trips at the end of every
  // tile (the compiler drains vmcnt behind each conditional pair) -- a third of a short-K tile's time (round 4).
  float ld_sc[FN], ld_bi[FN];
  {
    const float* scp = p.scale ? p.scale : p.bias;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      u32 n = n0 + wn * (FN * 16) + j * 16 + fr;
      n = n < (u32)p.Cout ? n : (u32)p.Cout - 1u;
      ld_sc[j] = scp[n];
      ld_bi[j] = p.bias[n];
    }
  }
  load_tile();
  store_tile(0);
  __syncthreads();
  const int KT = kt_end - kt_begin;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile();  // global loads in flight while the MFMAs run
    const unsigned char* sA = smem + cur * STAGE;
    const unsigned char* sB = sA + A_BYTES;
    u32x4 fa[FM], fb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const u32 row = wm * (FM * 16) + i * 16 + fr;
      fa[i] = *reinterpret_cast<const u32x4*>(sA + row * 64 + swz(row, fg) * 16);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const u32 row = wn * (FN * 16) + j * 16 + fr;
      fb[j] = *reinterpret_cast<const u32x4*>(sB + row * 64 + swz(row, fg) * 16);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  if constexpr (SPLITK) {
    // ---- split-K hand-off (cdna_hip_programming.md G16, counter form): slab stores -> every wave drains
    //      vmcnt -> barrier -> one lane: agent release + drained wait -> relaxed ticket; the last arriver
    //      acquires once, then every wave reads the other slabs with plain loads. ------------------------
    const u32 tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr int FRAGS = FM * FN;
    float* my = p.slabs + ((size_t)tile_id * p.ksplits + blockIdx.z) * (size_t)(4 * FRAGS * 256);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<f32x4*>(my + ((size_t)(i * FN + j) * 256 + tid) * 4) = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32* flag = reinterpret_cast<u32*>(smem);  // staging buffers are free now
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(p.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned last = (old == (unsigned)p.ksplits - 1u) ? 1u : 0u;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
      *flag = last;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();  // everyone has read the flag before the epilogue reuses the LDS
    if (!last) return;
    // sum the slabs in slice order, the reducer's own included: the result must not depend on which slice
    // happens to arrive last (bit-reproducible outputs run to run)
    const float* base = p.slabs + (size_t)tile_id * p.ksplits * (size_t)(4 * FRAGS * 256);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(base + ((size_t)(i * FN + j) * 256 + tid) * 4);
    for (int z = 1; z < p.ksplits; ++z) {
      const float* other = base + (size_t)z * (size_t)(4 * FRAGS * 256);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(other + ((size_t)(i * FN + j) * 256 + tid) * 4);
          acc[i][j] += v;
        }
    }
  }

  // ---- epilogue: scale/bias/activation in fp32, tile through LDS, 16-byte coalesced stores ------------
  // (the last __syncthreads() above guarantees nobody still reads the staging buffers)
  u16* sC = reinterpret_cast<u16*>(smem);
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  const int post = p.post;
  constexpr int LDC_M = BN + 8;   // NHWC image: sC[m][n], row stride in elements
  constexpr int LDC_N = BM + 8;   // NCHW image: sC[n][m]
  const bool any_sig = act_is_sig(p.act) || act_is_sig(p.act2);
  const bool any_clamp = act_is_clamp(p.act) || act_is_clamp(p.act2);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const u32 nl = wn * (FN * 16) + j * 16 + fr;
    const u32 n = n0 + nl;
    const bool col = n < (u32)p.Cout;
    const float sc = (col && p.scale) ? ld_sc[j] : 1.f, bi = col ? ld_bi[j] : 0.f;
    const int act = (col && (int)n >= p.split) ? p.act2 : p.act;
    const ActSel as = act_sel(act);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const u32 ml = wm * (FM * 16) + i * 16 + fg * 4;
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

  if (!nchw && RES) {
    // ResNet / RegNet bottleneck tails: every chunk is whole (Cout % 8 == 0, checked by the host).  In the generic loop below
    // each of a thread's eight iterations waits for its own residual load: here four are in flight at a time.
    constexpr int CH = BN / 8, IT = (BM * CH + kConvThreads - 1) / kConvThreads, RB = 4;
#pragma unroll
    for (int b0 = 0; b0 < IT; b0 += RB) {
      u32x4 rr[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const u32 q = tid + (u32)(b0 + k) * kConvThreads, row = q / CH, cc = q % CH;
        const u32 m = m0 + row, n = n0 + cc * 8;
        rr[k] = u32x4{0u, 0u, 0u, 0u};
        if (b0 + k < IT && q < (u32)(BM * CH) && m < (u32)p.M && n < (u32)p.Cout)
          rr[k] = *reinterpret_cast<const u32x4*>((const u16*)p.res + res_pixel_offset(p, m) + n);
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const u32 q = tid + (u32)(b0 + k) * kConvThreads, row = q / CH, cc = q % CH;
        const u32 m = m0 + row, n = n0 + cc * 8;
        if (b0 + k < IT && q < (u32)(BM * CH) && m < (u32)p.M && n < (u32)p.Cout) {
          u32x4 v = *reinterpret_cast<const u32x4*>(&sC[row * LDC_M + cc * 8]);
          const u32x4 r4 = rr[k];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = post_act(bits16_to_f32<DT>(v[e] & 0xffffu) + bits16_to_f32<DT>(r4[e] & 0xffffu), post);
            const float hi = post_act(bits16_to_f32<DT>(v[e] >> 16) + bits16_to_f32<DT>(r4[e] >> 16), post);
            v[e] = pack2_16<DT>(lo, hi);
          }
          *reinterpret_cast<u32x4*>((u16*)p.y + (size_t)m * p.Cout + n) = v;
        }
      }
    }
  } else if (!nchw) {
    constexpr int CH = BN / 8;  // 16-byte chunks per tile row
    for (u32 q = tid; q < (u32)(BM * CH); q += kConvThreads) {
      const u32 row = q / CH, cc = q % CH;
      const u32 m = m0 + row, n = n0 + cc * 8;
      if (m >= (u32)p.M || n >= (u32)p.Cout) continue;
      u32x4 v = *reinterpret_cast<const u32x4*>(&sC[row * LDC_M + cc * 8]);
      u16* dst = (u16*)p.y + (size_t)m * p.Cout + n;
      if (n + 8 <= (u32)p.Cout) {
        if (p.res) v = add_residual8<DT>(v, (const u16*)p.res + res_pixel_offset(p, m) + n, post);
        *reinterpret_cast<u32x4*>(dst) = v;
      } else {
        for (u32 e = 0; e < 8 && n + e < (u32)p.Cout; ++e) {
          float f = bits16_to_f32<DT>((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          if (p.res) f = post_act(f + bits16_to_f32<DT>(((const u16*)p.res)[res_pixel_offset(p, m) + n + e]), post);
          dst[e] = (u16)f32_to_bits16<DT>(f);
        }
      }
    }
  } else {
    constexpr int CH = BM / 8;  // 16-byte chunks (8 consecutive m) per channel row
    const u32 hw = (u32)(p.Ho * p.Wo);
    for (u32 q = tid; q < (u32)(BN * CH); q += kConvThreads) {
      const u32 nl = q / CH, cc = q % CH;
      const u32 n = n0 + nl, m = m0 + cc * 8;
      if (n >= (u32)p.Cout || m >= (u32)p.M) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&sC[nl * LDC_N + cc * 8]);
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
      const u32 b = m / hw, pix = m % hw;
      u16* dst = ybase + ((size_t)b * cy + ch) * hw + pix;
      if (pix + 8 <= hw && m + 8 <= (u32)p.M && (((uintptr_t)dst) & 15u) == 0) {
        *reinterpret_cast<u32x4*>(dst) = v;
      } else {
        for (u32 e = 0; e < 8 && m + e < (u32)p.M; ++e) {
          const u32 mm = m + e, bb = mm / hw, pp = mm % hw;
          ybase[((size_t)bb * cy + ch) * hw + pp] = (u16)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_wave_kernel: the tiny-layer variant (SSD extras at 8x8 .. 1x1, heads of the last levels).
//
// These layers hold a few MFLOP each and are pure latency: with the tiled kernel every k-step is a
// global-load -> LDS -> barrier round trip (~1 us) and a layer costs 15-40 us whatever its size.  Here ONE WAVE
// owns a 16-pixel x 64-channel output tile and needs no LDS and no barrier: each lane loads its own MFMA
// fragments (16 bytes of one pixel row / one weight row) straight from L2, four k-steps of loads are in flight
// ahead of the matrix pipe, and the k range is split over blockIdx-derived slices when the tile grid is small.
// Split-K partials go to fp32 slabs in fragment order; the last slice to arrive on a tile (agent-scope
// release / acquire around one relaxed ticket, all inside the wave) sums them and runs the epilogue.
// Operand re-reads (x by every n-tile, w by every m-tile) come from L2 and bound the kernel's use to small M*N*K.
// ------------------------------------------------------------------------------------------------
constexpr int WV_DEPTH = 4;

template <int DT>
__global__ __launch_bounds__(256) void conv_wave_kernel(const ConvParams p, int mt, int ntl) {
  const u32 lane = threadIdx.x & 63u;
  const u32 gid = blockIdx.x * 4u + (threadIdx.x >> 6);
  const u32 tiles = (u32)mt * (u32)ntl;
  if (gid >= tiles * (u32)p.ksplits) return;  // wave-uniform
  const u32 z = gid / tiles, tile = gid % tiles;
  const u32 m0 = (tile % (u32)mt) * 16u, n0 = (tile / (u32)mt) * 64u;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int Cin = p.Cin, H = p.H, W = p.W;
  const int KK = p.k * p.k;

  // A fragment row of this lane = output pixel m0 + fr
  long a_base = 0;
  u32 a_mask = 0;
  {
    const u32 m = m0 + fr;
    if (m < (u32)p.M) {
      const u32 hw = (u32)(p.Ho * p.Wo);
      const u32 b = m / hw, r = m % hw;
      const int oy = (int)(r / (u32)p.Wo), ox = (int)(r % (u32)p.Wo);
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      a_base = (((long)b * H + iy0) * W + ix0) * Cin;
      for (int ky = 0; ky < p.k; ++ky)
        for (int kx = 0; kx < p.k; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W) a_mask |= 1u << (ky * p.k + kx);
    }
  }
  long w_base[4];
  bool w_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 n = n0 + j * 16 + fr;
    w_ok[j] = n < (u32)p.Cout;
    w_base[j] = (long)n * KK * Cin;
  }
  const int kt_begin = (int)z * p.kt_per;
  const int kt_end = (kt_begin + p.kt_per < p.KT) ? kt_begin + p.kt_per : p.KT;
  int t_kpos = kt_begin / p.cin_chunks, t_cc = kt_begin % p.cin_chunks;
  int t_ky = t_kpos / p.k, t_kx = t_kpos % p.k;
  int t_left = kt_end - kt_begin;  // k-steps not yet loaded
  auto load = [&](u32x4& ra, u32x4 (&rb)[4]) {  // fragments of the next k-step (zeros past the end), then advance
    const int ci = t_cc * 32 + (int)fg * 8;
    const bool live = t_left > 0 && ci < Cin;
    ra = u32x4{0u, 0u, 0u, 0u};
    if (live && ((a_mask >> t_kpos) & 1u))
      ra = *reinterpret_cast<const u32x4*>((const u16*)p.x + (a_base + (long)(t_ky * W + t_kx) * Cin + ci));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb[j] = u32x4{0u, 0u, 0u, 0u};
      if (live && w_ok[j]) rb[j] = *reinterpret_cast<const u32x4*>((const u16*)p.w + (w_base[j] + (long)t_kpos * Cin + ci));
    }
    --t_left;
    if (++t_cc == p.cin_chunks) {
      t_cc = 0;
      ++t_kpos;
      if (++t_kx == p.k) {
        t_kx = 0;
        ++t_ky;
      }
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 ra[WV_DEPTH], rb[WV_DEPTH][4];
#pragma unroll
  for (int d = 0; d < WV_DEPTH; ++d) load(ra[d], rb[d]);
  const int nsteps = kt_end - kt_begin;
  for (int s0 = 0; s0 < nsteps; s0 += WV_DEPTH) {
#pragma unroll
    for (int d = 0; d < WV_DEPTH; ++d) {
      // steps past the end hold zero fragments: the MFMAs are harmless and keep the loop branch-free
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma16<DT>(ra[d], rb[d][j], acc[j]);
      load(ra[d], rb[d]);
    }
  }

  if (p.ksplits > 1) {
    float* my = p.slabs + ((size_t)tile * p.ksplits + z) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(my + (j * 64 + (int)lane) * 4) = acc[j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32 last = 0;
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (old == (unsigned)p.ksplits - 1u) ? 1u : 0u;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
    }
    last = (u32)__builtin_amdgcn_readfirstlane((int)last);
    if (!last) return;
    // slabs summed in slice order (own slice included): independent of the arrival order, bit-reproducible
    const float* base = p.slabs + (size_t)tile * p.ksplits * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(base + (j * 64 + (int)lane) * 4);
    for (int zz = 1; zz < p.ksplits; ++zz) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += *reinterpret_cast<const f32x4*>(base + (size_t)zz * 1024 + (j * 64 + (int)lane) * 4);
    }
  }

  // epilogue: D[row = fg*4 + r][col = fr] of fragment j  ->  pixel m0 + fg*4 + r, channel n0 + j*16 + fr
  const bool any_sig = act_is_sig(p.act) || act_is_sig(p.act2);
  const bool any_clamp = act_is_clamp(p.act) || act_is_clamp(p.act2);
  const u32 hw = (u32)(p.Ho * p.Wo);
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  const int post = p.post;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 n = n0 + j * 16 + fr;
    if (n >= (u32)p.Cout) continue;
    const float sc = p.scale ? p.scale[n] : 1.f, bi = p.bias[n];
    const ActSel as = act_sel((int)n >= p.split ? p.act2 : p.act);
    const uint2 h = epilogue4<DT>(acc[j], sc, bi, as, any_sig, any_clamp);
    const u32 hv[4] = {h.x & 0xffffu, h.x >> 16, h.y & 0xffffu, h.y >> 16};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u32 m = m0 + fg * 4 + r;
      if (m >= (u32)p.M) continue;
      if (!nchw) {
        u32 v = hv[r];
        if (p.res) v = add_residual1<DT>(v, (const u16*)p.res + res_pixel_offset(p, m) + n, post);
        ((u16*)p.y)[(size_t)m * p.Cout + n] = (u16)v;
      } else {
        const u32 b = m / hw, pix = m % hw;
        if ((int)n < p.split) ((u16*)p.y)[((size_t)b * p.split + n) * hw + pix] = (u16)hv[r];
        else ((u16*)p.y2)[((size_t)b * (p.Cout - p.split) + (n - p.split)) * hw + pix] = (u16)hv[r];
      }
    }
  }
}

// tiny layers: operand re-reads from L2 stay below ~100 MB and the tile grid is far from filling the chip
static bool wave_plan(long M, int Cin, int Cout, int k, int* splits_out, int* kt_per_out, int* mt_out, int* ntl_out) {
  static const int env = getenv("SSDK_CONV_WAVE") ? atoi(getenv("SSDK_CONV_WAVE")) : 1;
  if (!env || (Cin % 8)) return false;
  const double work = (double)M * Cout * Cin * k * k;
  if (env != 2 && (work > 3e8 || M > 256)) return false;  // measured: the tiled kernel wins again from M = 1024 on
  const int mt = (int)((M + 15) / 16), ntl = (Cout + 63) / 64;
  const long tiles = (long)mt * ntl;
  const int KT = k * k * ((Cin + 31) / 32);
  long s = (768 + tiles - 1) / tiles;
  if (s > KT / 4) s = KT / 4;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  if (tiles > 1024) s = 1;  // counters live in the first 4 KiB of the workspace
  int kt_per = (int)((KT + s - 1) / s);
  s = (KT + kt_per - 1) / kt_per;
  if (splits_out) *splits_out = (int)s;
  if (kt_per_out) *kt_per_out = kt_per;
  if (mt_out) *mt_out = mt;
  if (ntl_out) *ntl_out = ntl;
  return true;
}

// ------------------------------------------------------------------------------------------------
// conv_gemm256_kernel: the large-layer variant (SSD heads L0-L2, FPN/BiFPN towers, wide 1x1 convs).
//
// 256 x 256 x 64 tiles, 8 waves as 2(M) x 4(N), each wave a 128 x 64 sub-tile = 8 x 4 accumulator fragments of
// v_mfma_f32_16x16x32.  Both operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no staging VGPRs, no ds_write pass) into two 64 KiB stages; the LDS image is lane-linear per
// instruction, so the bank swizzle (16-byte chunk index ^ (row >> 1) & 7 inside each 128-byte row) is applied to
// the SOURCE address: lane -> (row, logical chunk) is chosen such that the physical chunk is the lane's slot.
// The reduction index is the flat k = (ky*kw + kx)*Cin + ci (tap-major, channel-minor = KRSC weight order), so a
// 64-wide k-tile may straddle taps: every lane tracks (tap, ci) of its own 8-channel chunk.  Out-of-image taps,
// rows >= M / >= Cout and k >= K read a 16-byte zero page instead (a masked lane would leave stale LDS).
// One workgroup per CU (128 KiB LDS); tile ids are remapped so that the tiles sharing an M-panel run on one XCD.
// ------------------------------------------------------------------------------------------------

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64, G2_THREADS = 512;
constexpr int G2_STAGE = (G2_BM + G2_BN) * G2_BK * 2;  // 64 KiB
constexpr int G2_LDS = 2 * G2_STAGE;

template <int DT>
__global__ __launch_bounds__(G2_THREADS) void conv_gemm256_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const u32 wm = wave >> 2, wn = wave & 3u;

  // XCD-aware (bijective) tile order: ids i, i+8, i+16 ... share an XCD; give each XCD a contiguous tile range
  const u32 nwg = gridDim.x, id = blockIdx.x;
  const u32 q = nwg >> 3, r8 = nwg & 7u, xcd = id & 7u;
  const u32 lin = (xcd < r8 ? xcd * (q + 1u) : r8 * (q + 1u) + (xcd - r8) * q) + (id >> 3);
  const u32 NT = (u32)((p.Cout + G2_BN - 1) / G2_BN);
  const u32 m0 = (lin / NT) * G2_BM, n0 = (lin % NT) * G2_BN;

  const int Cin = p.Cin, H = p.H, W = p.W;
  const int Ktot = p.k * p.k * Cin;

  // ---- loader roles: wave instruction g = j*8 + wave covers tile rows g*8 .. g*8+7 (1 KiB of LDS) -------------
  const u32 lrow = lane >> 3;                                        // row inside the 8-row group
  const u32 lchunk = (lane & 7u) ^ (((lane >> 4) + 4u * (wave & 1u)) & 7u);  // logical 16-byte chunk of this lane
  long a_off[4];
  u32 a_mask[4];
  long b_off[4];
  bool b_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 row = ((u32)j * 8u + wave) * 8u + lrow;
    const u32 m = m0 + row;
    u32 mask = 0;
    long base = 0;
    if (m < (u32)p.M) {
      const u32 hw = (u32)(p.Ho * p.Wo);
      const u32 b = m / hw, r = m % hw;
      const int oy = (int)(r / (u32)p.Wo), ox = (int)(r % (u32)p.Wo);
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      base = (((long)b * H + iy0) * W + ix0) * Cin;
      for (int ky = 0; ky < p.k; ++ky)
        for (int kx = 0; kx < p.k; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W) mask |= 1u << (ky * p.k + kx);
    }
    a_off[j] = base;
    a_mask[j] = mask;
    const u32 n = n0 + row;
    b_ok[j] = n < (u32)p.Cout;
    b_off[j] = (long)n * Ktot;
  }
  // k-tiles are visited channel-chunk-major, tap-minor: the nine taps of one 64-channel slab re-read the same
  // 256-pixel (+halo) footprint of 128 B per pixel back to back, which stays in L2 (tap-major order re-reads the
  // whole Cin-deep footprint only every Cin/64 steps and thrashes the 4 MiB L2 of an XCD at Cin >= 256).
  const int KK = p.k * p.k;
  const int cchunks = (Cin + G2_BK - 1) / G2_BK;
  int t_tap = 0, t_cc = 0;
  // masked chunks read the zero page: select the byte OFFSET (one v_cndmask pair), not the pointer (branches)
  const unsigned char* xg = (const unsigned char*)p.x;
  const unsigned char* wg = (const unsigned char*)p.w;
  const long zx = reinterpret_cast<const unsigned char*>(g_zero16) - xg;
  const long zw = reinterpret_cast<const unsigned char*>(g_zero16) - wg;

  auto issue = [&](int buf) {
    unsigned char* sA = smem + buf * G2_STAGE + wave * 1024u;
    unsigned char* sB = sA + G2_BM * G2_BK * 2;
    const int ci = t_cc * G2_BK + (int)lchunk * 8;
    const bool cok = ci < Cin;
    const int ky = (p.k == 3) ? ((t_tap * 11) >> 5) : 0;
    const int kx = t_tap - ky * p.k;
    const int tap = (ky * W + kx) * Cin + ci;
    const int wk = t_tap * Cin + ci;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = cok && ((a_mask[j] >> t_tap) & 1u);
      const long off = ok ? (a_off[j] + tap) * 2 : zx;
      glds16(xg + off, sA + j * 8192);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = cok && b_ok[j];
      const long off = ok ? (b_off[j] + wk) * 2 : zw;
      glds16(wg + off, sB + j * 8192);
    }
    if (++t_tap == KK) {
      t_tap = 0;
      ++t_cc;
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const u32 fr = lane & 15u, fg = lane >> 4;
  const u32 a_rd = (wm * 128u + fr) * 128u;  // byte offset of fragment row 0 of this wave (i adds 16 rows = 2 KiB)
  const u32 b_rd = (u32)(G2_BM * G2_BK * 2) + (wn * 64u + fr) * 128u;
  const u32 sw = fr >> 1;  // (row >> 1) & 7 for every fragment row of this lane

  const int KT = KK * cchunks;
  issue(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) issue(cur ^ 1);
    const unsigned char* st = smem + cur * G2_STAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const u32 ch = (((u32)s2 * 4u + fg) ^ sw) * 16u;
      u32x4 fa[8], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(st + b_rd + j * 2048 + ch);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + a_rd + i * 2048 + ch);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue in two passes of 128 output channels (the staging image must fit the 128 KiB) ------------------
  u16* sC = reinterpret_cast<u16*>(smem);
  const bool nchw = p.out_layout == LAYOUT_NCHW;
  const int post = p.post;
  constexpr int LDC_M = 128 + 8;      // NHWC image sC[m][n-local]
  constexpr int LDC_N = G2_BM + 8;    // NCHW image sC[n-local][m]
  const u32 hw = (u32)(p.Ho * p.Wo);
  // this wave's eight per-column constants in ONE round trip (unconditional loads from clamped indices; inside the column loop
  // under `if (n < Cout)` they were four dependent round trips per wave)
  float ld_sc[4], ld_bi[4];
  {
    const float* scp = p.scale ? p.scale : p.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 n = n0 + (wn >> 1) * 128u + (wn & 1u) * 64u + j * 16 + fr;
      n = n < (u32)p.Cout ? n : (u32)p.Cout - 1u;
      ld_sc[j] = scp[n];
      ld_bi[j] = p.bias[n];
    }
  }
  for (u32 half = 0; half < 2; ++half) {
    if (n0 + half * 128u >= (u32)p.Cout) break;  // workgroup-uniform
    if ((wn >> 1) == half) {
      const bool any_sig = act_is_sig(p.act) || act_is_sig(p.act2);
      const bool any_clamp = act_is_clamp(p.act) || act_is_clamp(p.act2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 nl = (wn & 1u) * 64u + j * 16 + fr;
        const u32 n = n0 + half * 128u + nl;
        const bool col = n < (u32)p.Cout;
        const float sc = (col && p.scale) ? ld_sc[j] : 1.f, bi = col ? ld_bi[j] : 0.f;
        const int act = (col && (int)n >= p.split) ? p.act2 : p.act;
        const ActSel as = act_sel(act);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const u32 ml = wm * 128u + i * 16 + fg * 4;
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
    }
    __syncthreads();
    if (!nchw) {
      for (u32 qd = tid; qd < (u32)(G2_BM * 16); qd += G2_THREADS) {
        const u32 row = qd >> 4, cc = qd & 15u;
        const u32 m = m0 + row, n = n0 + half * 128u + cc * 8;
        if (m >= (u32)p.M || n >= (u32)p.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(&sC[row * LDC_M + cc * 8]);
        u16* dst = (u16*)p.y + (size_t)m * p.Cout + n;
        if (n + 8 <= (u32)p.Cout) {
          if (p.res) v = add_residual8<DT>(v, (const u16*)p.res + res_pixel_offset(p, m) + n, post);
          *reinterpret_cast<u32x4*>(dst) = v;
        } else {
          for (u32 e = 0; e < 8 && n + e < (u32)p.Cout; ++e) {
            float f = bits16_to_f32<DT>((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            if (p.res) f = post_act(f + bits16_to_f32<DT>(((const u16*)p.res)[res_pixel_offset(p, m) + n + e]), post);
            dst[e] = (u16)f32_to_bits16<DT>(f);
          }
        }
      }
    } else {
      for (u32 qd = tid; qd < (u32)(128 * 32); qd += G2_THREADS) {
        const u32 nl = qd >> 5, cc = qd & 31u;
        const u32 n = n0 + half * 128u + nl, m = m0 + cc * 8;
        if (n >= (u32)p.Cout || m >= (u32)p.M) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&sC[nl * LDC_N + cc * 8]);
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
        const u32 b = m / hw, pix = m % hw;
        u16* dst = ybase + ((size_t)b * cy + ch) * hw + pix;
        if (pix + 8 <= hw && m + 8 <= (u32)p.M && (((uintptr_t)dst) & 15u) == 0) {
          *reinterpret_cast<u32x4*>(dst) = v;
        } else {
          for (u32 e = 0; e < 8 && m + e < (u32)p.M; ++e) {
            const u32 mm = m + e, bb = mm / hw, pp = mm % hw;
            ybase[((size_t)bb * cy + ch) * hw + pp] = (u16)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 3x3
// ------------------------------------------------------------------------------------------------
struct DwParams {
  const void* x;
  const void* w;  // [3][3][C]
  const float* scale;
  const float* bias;
  void* y;
  int N, C, H, W, stride, Ho, Wo, act;
  int cgroups;   // C/8
  int wgroups;   // ceil(Wo/4)
  long total;    // N*Ho*wgroups*cgroups
};

template <int DT, int S>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const DwParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const int cg = (int)(t % p.cgroups);
  long r = t / p.cgroups;
  const int wg = (int)(r % p.wgroups);
  r /= p.wgroups;
  const int oy = (int)(r % p.Ho);
  const int n = (int)(r / p.Ho);
  const int c0 = cg * 8, ox0 = wg * 4;
  constexpr int s = S;
  const u16* x = (const u16*)p.x + (size_t)n * p.H * p.W * p.C + c0;
  const u16* w = (const u16*)p.w + c0;

  float acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;

  constexpr int ncol = 3 * s + 3;  // input columns covering 4 outputs: 6 (s=1) or 9 (s=2)
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * s + ky - 1;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    float wk[3][8];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(w + (size_t)(ky * 3 + kx) * p.C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wk[kx][2 * e] = bits16_to_f32<DT>(wv[e] & 0xffffu);
        wk[kx][2 * e + 1] = bits16_to_f32<DT>(wv[e] >> 16);
      }
    }
    const u16* xr = x + (size_t)iy * p.W * p.C;
#pragma unroll
    for (int col = 0; col < ncol; ++col) {
      const int ix = ox0 * s + col - 1;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const u32x4 xv = *reinterpret_cast<const u32x4*>(xr + (size_t)ix * p.C);
      float xf[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xf[2 * e] = bits16_to_f32<DT>(xv[e] & 0xffffu);
        xf[2 * e + 1] = bits16_to_f32<DT>(xv[e] >> 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        constexpr int dummy = 0;
        (void)dummy;
        const int kx = col - i * s;  // this column feeds output i with tap kx (compile-time after unrolling)
        if (kx >= 0 && kx < 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[i][e] = fmaf(xf[e], wk[kx][e], acc[i][e]);
        }
      }
    }
  }
  float sc[8], bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = p.scale ? p.scale[c0 + e] : 1.f;
    bi[e] = p.bias[c0 + e];
  }
  u16* y = (u16*)p.y + (((size_t)n * p.Ho + oy) * p.Wo) * p.C + c0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ox = ox0 + i;
    if (ox >= p.Wo) break;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32 lo = f32_to_bits16<DT>(apply_act(acc[i][2 * e] * sc[2 * e] + bi[2 * e], p.act));
      const u32 hi = f32_to_bits16<DT>(apply_act(acc[i][2 * e + 1] * sc[2 * e + 1] + bi[2 * e + 1], p.act));
      o[e] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(y + (size_t)ox * p.C) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// 3-channel stem: 3x3 stride s, Cin <= 4, Cout multiple of 8 (<= 64), image NCHW or NHWC -> NHWC
// ------------------------------------------------------------------------------------------------
struct FirstParams {
  const void* x;
  const float* w;  // fp32 [Cout][3][3][Cin], BN scale already folded in by the host
  const float* bias;
  void* y;
  int N, Cin, H, W, Cout, stride, Ho, Wo, act, in_layout;
  long total;  // N*Ho*Wo
};

template <int DT, int COUT>
__global__ __launch_bounds__(256) void conv_first_kernel(const FirstParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const int ox = (int)(t % p.Wo);
  const long r = t / p.Wo;
  const int oy = (int)(r % p.Ho);
  const int n = (int)(r / p.Ho);
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = p.bias[c];
  const float* __restrict__ w = p.w;
  if (p.Cin == 3) {
    // the usual case, unrolled: all 27 input values are requested before the first one is used (clamped address + zero mask
    // instead of branches), the weights are uniform (scalar loads with static offsets).  As a triple loop with `continue`s
    // every 2-byte load waited a full memory round trip: 271 us for the RegNet stem (3 -> 32 @896x896, batch 16).
    float xv[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * p.stride + ky - 1, ix = ox * p.stride + kx - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int cy = ok ? iy : 0, cx = ok ? ix : 0;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const size_t off = p.in_layout == LAYOUT_NCHW ? (((size_t)n * 3 + ci) * p.H + cy) * p.W + cx
                                                        : (((size_t)n * p.H + cy) * p.W + cx) * 3 + ci;
          const float v = bits16_to_f32<DT>(((const u16*)p.x)[off]);
          xv[(ky * 3 + kx) * 3 + ci] = ok ? v : 0.f;
        }
      }
#pragma unroll
    for (int q = 0; q < 27; ++q)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xv[q], w[(size_t)c * 27 + q], acc[c]);
  } else {
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * p.stride + ky - 1;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * p.stride + kx - 1;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      for (int ci = 0; ci < p.Cin; ++ci) {
        const size_t off = p.in_layout == LAYOUT_NCHW
                               ? (((size_t)n * p.Cin + ci) * p.H + iy) * p.W + ix
                               : (((size_t)n * p.H + iy) * p.W + ix) * p.Cin + ci;
        const float xv = bits16_to_f32<DT>(((const u16*)p.x)[off]);
        const float* wp = w + ((ky * 3 + kx) * p.Cin + ci);  // uniform -> scalar loads
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xv, wp[(size_t)c * 9 * p.Cin], acc[c]);
      }
    }
  }
  }
  u16* y = (u16*)p.y + (size_t)t * COUT;
#pragma unroll
  for (int c8 = 0; c8 < COUT / 8; ++c8) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32 lo = f32_to_bits16<DT>(apply_act(acc[c8 * 8 + 2 * e], p.act));
      const u32 hi = f32_to_bits16<DT>(apply_act(acc[c8 * 8 + 2 * e + 1], p.act));
      o[e] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(y + c8 * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int gemm_bn(int cout) { return cout > 64 ? 128 : cout > 32 ? 64 : cout > 16 ? 32 : 16; }

// split-K plan: only when the tile grid cannot fill the chip and the k-loop is long enough to amortise
// the slab round trip; aims at ~2 workgroups per CU.
static int plan_splits(int M, int Cout, int KT) {
  static const int env = getenv("SSDK_SPLITK") ? atoi(getenv("SSDK_SPLITK")) : 1;
  constexpr int target = 512;  // (round 6: the SSDK_SPLITK_WGS switch is gone, its A/B is settled)
  constexpr int min_kt = 8;  // (round 6: the SSDK_SPLITK_MINKT switch is gone, its A/B is settled)
  if (!env) return 1;
  const int bn = gemm_bn(Cout);
  const long tiles = (long)((M + BM - 1) / BM) * ((Cout + bn - 1) / bn);
  if (tiles >= 192 || KT < 2 * min_kt) return 1;
  long s = (target + tiles - 1) / tiles;
  if (s > KT / min_kt) s = KT / min_kt;
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}

// the 256x256 glds kernel pays when there are enough output tiles to fill the chip and N is wide
static bool use_gemm256(const ConvParams& p) {
  static const int env = getenv("SSDK_GEMM256") ? atoi(getenv("SSDK_GEMM256")) : 1;
  if (!env || p.ksplits > 1 || (p.Cin % 8)) return false;
  const long tiles = (long)((p.M + G2_BM - 1) / G2_BM) * ((p.Cout + G2_BN - 1) / G2_BN);
  if (env == 2) return true;
  // ... and a k-loop long enough to fill its two 64 KiB stages: measured per layer on FPN-ResNet50@640 (batch 32) against
  // conv_gemm_kernel, 1x1 layers: 64 -> 256 @160x160 355 vs 262 us, 128 -> 512 @80x80 224 vs 173, 256 -> 1024 @40x40 131 vs
  // 100, 512 -> 256 @80x80 148 vs 136, 512 -> 2048 @20x20 79 vs 77, but 1024 -> 256 @40x40 53 vs 68: with K < 1024 the
  // 256 x 256 tile is one or two loads, one burst of MFMAs and a 128 KiB store, one workgroup per CU, nothing overlapping
  return p.Cout >= 192 && tiles >= 128 && p.KT >= 32;
}

template <int DT>
static int launch_gemm256(const ConvParams& p, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm256_kernel<DT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    attr_done = true;
  }
  const unsigned tiles = (unsigned)((p.M + G2_BM - 1) / G2_BM) * (unsigned)((p.Cout + G2_BN - 1) / G2_BN);
  hipLaunchKernelGGL((conv_gemm256_kernel<DT>), dim3(tiles), dim3(G2_THREADS), G2_LDS, stream, p);
  return check_launch("conv_gemm256_kernel");
}

template <int DT>
static int launch_gemm(const ConvParams& p, hipStream_t stream) {
  if (use_gemm256(p)) return launch_gemm256<DT>(p, stream);
  const unsigned gm = (unsigned)((p.M + BM - 1) / BM);
  const unsigned gz = (unsigned)p.ksplits;
  constexpr int env_resv = 1;  // (round 6: the SSDK_GEMM_RESV switch is gone, its A/B is settled)
  const bool resv = env_resv && p.res != nullptr && p.out_layout == LAYOUT_NHWC && (p.Cout & 7) == 0 && p.Cout > 32;
#define SSDK_GEMM(WM, WN, FM_, FN_, GY)                                                                          \
  do {                                                                                                          \
    dim3 grid(gm, (unsigned)(GY), gz);                                                                          \
    if (gz > 1) hipLaunchKernelGGL((conv_gemm_kernel<DT, WM, WN, FM_, FN_, true>), grid, dim3(kConvThreads), 0, stream, p); \
    else if (resv) hipLaunchKernelGGL((conv_gemm_kernel<DT, WM, WN, FM_, FN_, false, true>), grid, dim3(kConvThreads), 0, stream, p); \
    else hipLaunchKernelGGL((conv_gemm_kernel<DT, WM, WN, FM_, FN_, false>), grid, dim3(kConvThreads), 0, stream, p);      \
  } while (0)
  // short K on a grid that gives 128-wide tiles one workgroup per CU at most (the 1x1 320 -> 256 layer of the first SSD
  // extra: 10 k-steps, 256 tiles): 64-wide tiles = two workgroups per CU to overlap the per-k-step latency chain
  constexpr int env_n64 = 1;  // (round 6: the SSDK_GEMM_N64 switch is gone, its A/B is settled)
  const bool n64 = env_n64 && gz == 1 && p.Cout > 64 && (p.Cout % 64) == 0 && p.KT <= 12 && (long)gm * ((p.Cout + 127) / 128) <= 256;
  if (n64) SSDK_GEMM(4, 1, 2, 4, (p.Cout + 63) / 64);
  else if (p.Cout > 64) SSDK_GEMM(2, 2, 4, 4, (p.Cout + 127) / 128);
  else if (p.Cout > 32) SSDK_GEMM(4, 1, 2, 4, (p.Cout + 63) / 64);
  else if (p.Cout > 16) SSDK_GEMM(4, 1, 2, 2, 1);
  else SSDK_GEMM(4, 1, 2, 1, 1);
#undef SSDK_GEMM
  return check_launch("conv_gemm_kernel");
}

}  // namespace ssdk

using namespace ssdk;

static size_t splitk_ws_bytes(long M, int Cin, int Cout, int k, int* splits_out, int* kt_per_out) {
  const int cin_chunks = (Cin + BK - 1) / BK;
  const int KT = k * k * cin_chunks;
  int splits = plan_splits((int)M, Cout, KT);
  const int kt_per = (KT + splits - 1) / splits;
  splits = (KT + kt_per - 1) / kt_per;  // every split owns at least one k-tile
  if (splits_out) *splits_out = splits;
  if (kt_per_out) *kt_per_out = kt_per;
  if (splits <= 1) return 0;
  const int bn = gemm_bn(Cout);
  const size_t tiles = (size_t)((M + BM - 1) / BM) * ((Cout + bn - 1) / bn);
  return 4096 /* counters */ + tiles * splits * (size_t)BM * bn * 4;
}

extern "C" size_t ssdk_conv_workspace_bytes(int N, int Cin, int H, int W, int Cout, int k, int stride, int dtype) {
  (void)dtype;
  if (Cin <= 4 || (k != 1 && k != 3) || (stride != 1 && stride != 2)) return 0;
  const int pad = k / 2;
  const long M = (long)N * ((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
  int gemm_splits = 1;
  size_t need = splitk_ws_bytes(M, Cin, Cout, k, &gemm_splits, nullptr);
  if (k == 3 && stride == 1 && gemm_splits > 1) {  // the halo kernel may take the layer WITH the GEMM's split: its slabs are larger
    for (int nchw = 0; nchw < 2; ++nchw) {
      const size_t hb = halo_ws_bytes(N, Cin, H, W, Cout, nchw != 0, gemm_splits);
      if (hb > need) need = hb;
    }
  }
  if (k == 3 && stride == 1) {  // the halo kernel's own split-K (small maps, long K)
    size_t hb = 0;
    if (halo_splitk_plan(N, Cin, H, W, Cout, true, &hb) > 1 && hb > need) need = hb;
    if (halo_splitk_plan(N, Cin, H, W, Cout, false, &hb) > 1 && hb > need) need = hb;
  }
  int ws = 1, wk = 0, mt = 0, ntl = 0;
  if (wave_plan(M, Cin, Cout, k, &ws, &wk, &mt, &ntl) && ws > 1) {
    const size_t w = 4096 + (size_t)mt * ntl * ws * 4096;
    if (w > need) need = w;
  }
  return need;
}

static thread_local bool g_underfill_ok = false;  // set by ssdk_run_ops around side-lane ops

static int halo_splitk_min_pixels() {  // smallest map (pixels) that goes to the halo kernel's split-K instead of conv_smallmap
  return 64;  // (round 6: the SSDK_HALO_SPLITK_MINP switch is gone)
}

extern "C" size_t ssdk_weight_frag_bytes(int rows, int K) {
  if (rows < 1 || K < 32 || (K % 32)) return 0;
  return (size_t)((rows + 15) / 16) * 16 * (size_t)K * 2;
}

extern "C" int ssdk_conv(const ssdk_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ssdk::lds_poison(stream);
  if (!d || !d->x || !d->w || !d->bias || !d->y) {
    set_error("conv: null pointer (x, w, bias and y are mandatory)");
    return SSDK_E_BADARG;
  }
  if (d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) {
    set_error("conv: dtype must be bf16 or f16 (got %d)", d->dtype);
    return SSDK_E_BADARG;
  }
  if (d->N < 1 || d->Cin < 1 || d->H < 1 || d->W < 1 || d->Cout < 1 || (d->k != 1 && d->k != 3) ||
      (d->stride != 1 && d->stride != 2)) {
    set_error("conv: unsupported geometry N=%d Cin=%d H=%d W=%d Cout=%d k=%d stride=%d", d->N, d->Cin, d->H,
              d->W, d->Cout, d->k, d->stride);
    return SSDK_E_BADARG;
  }
  if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y | (uintptr_t)d->residual | (uintptr_t)d->y2) & 15) {
    set_error("conv: tensors must be 16-byte aligned");
    return SSDK_E_BADARG;
  }
  const int pad = d->k / 2;
  const int Ho = (d->H + 2 * pad - d->k) / d->stride + 1;
  const int Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
  const long M = (long)d->N * Ho * Wo;
  if (M >= (1l << 31)) {
    set_error("conv: too many output pixels");
    return SSDK_E_BADARG;
  }
  const int split = (d->y2 && d->split > 0 && d->split < d->Cout) ? d->split : d->Cout;

  if (d->groups == d->Cin && d->groups == d->Cout && d->groups > 1) {  // depthwise
    if (d->k != 3 || (d->Cin % 8) || d->in_layout != LAYOUT_NHWC || d->out_layout != LAYOUT_NHWC || d->residual ||
        split != d->Cout) {
      set_error("conv: depthwise needs k=3, C%%8==0, NHWC in/out, no residual/split");
      return SSDK_E_BADARG;
    }
    DwParams p;
    p.x = d->x;
    p.w = d->w;
    p.scale = d->scale;
    p.bias = d->bias;
    p.y = d->y;
    p.N = d->N;
    p.C = d->Cin;
    p.H = d->H;
    p.W = d->W;
    p.stride = d->stride;
    p.Ho = Ho;
    p.Wo = Wo;
    p.act = d->act;
    p.cgroups = d->Cin / 8;
    p.wgroups = (Wo + 3) / 4;
    p.total = (long)d->N * Ho * p.wgroups * p.cgroups;
    const unsigned grid = (unsigned)((p.total + 255) / 256);
    if (d->dtype == SSDK_BF16) {
      if (d->stride == 1) hipLaunchKernelGGL((dwconv3x3_kernel<SSDK_BF16, 1>), dim3(grid), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((dwconv3x3_kernel<SSDK_BF16, 2>), dim3(grid), dim3(256), 0, stream, p);
    } else {
      if (d->stride == 1) hipLaunchKernelGGL((dwconv3x3_kernel<SSDK_F16, 1>), dim3(grid), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((dwconv3x3_kernel<SSDK_F16, 2>), dim3(grid), dim3(256), 0, stream, p);
    }
    return check_launch("dwconv3x3_kernel");
  }
  if (d->groups != 1) {  // grouped: 16 channels per group (RegNetX bottlenecks)
    if (d->groups * 16 == d->Cin && d->Cin == d->Cout) return launch_gconv3x3_g16(d, Ho, Wo, stream);
    set_error("conv: groups must be 1 (dense), Cin (depthwise) or Cin/16 (16-channel groups); got %d for Cin=%d",
              d->groups, d->Cin);
    return SSDK_E_BADARG;
  }
  if (d->Cin <= 4) {  // stem: w is fp32 [Cout][3][3][Cin] with the BN scale folded in
    if (d->k != 3 || d->out_layout != LAYOUT_NHWC || d->residual || split != d->Cout || d->scale ||
        (d->Cout != 32 && d->Cout != 64 && d->Cout != 16)) {
      set_error("conv: stem path needs k=3, Cout in {16,32,64}, NHWC out, scale folded into fp32 weights");
      return SSDK_E_BADARG;
    }
    FirstParams p;
    p.x = d->x;
    p.w = (const float*)d->w;
    p.bias = d->bias;
    p.y = d->y;
    p.N = d->N;
    p.Cin = d->Cin;
    p.H = d->H;
    p.W = d->W;
    p.Cout = d->Cout;
    p.stride = d->stride;
    p.Ho = Ho;
    p.Wo = Wo;
    p.act = d->act;
    p.in_layout = d->in_layout;
    p.total = M;
    const unsigned grid = (unsigned)((M + 255) / 256);
#define SSDK_FIRST(DT, CO) hipLaunchKernelGGL((conv_first_kernel<DT, CO>), dim3(grid), dim3(256), 0, stream, p)
    if (d->dtype == SSDK_BF16) {
      if (d->Cout == 16) SSDK_FIRST(SSDK_BF16, 16);
      else if (d->Cout == 32) SSDK_FIRST(SSDK_BF16, 32);
      else SSDK_FIRST(SSDK_BF16, 64);
    } else {
      if (d->Cout == 16) SSDK_FIRST(SSDK_F16, 16);
      else if (d->Cout == 32) SSDK_FIRST(SSDK_F16, 32);
      else SSDK_FIRST(SSDK_F16, 64);
    }
#undef SSDK_FIRST
    return check_launch("conv_first_kernel");
  }
  if ((d->Cin % 8) || d->in_layout != LAYOUT_NHWC) {
    set_error("conv: the MFMA path needs NHWC input and Cin %% 8 == 0 (Cin=%d, layout=%d)", d->Cin, d->in_layout);
    return SSDK_E_BADARG;
  }
  if (d->out_layout == LAYOUT_NCHW && d->residual) {
    set_error("conv: residual add is only built for NHWC output");
    return SSDK_E_BADARG;
  }
  if (d->out_layout == LAYOUT_NHWC && split != d->Cout) {
    set_error("conv: split outputs are only built for NCHW output (multibox heads)");
    return SSDK_E_BADARG;
  }
  ConvParams p;
  p.x = d->x;
  p.w = d->w;
  p.w_frag = d->w_frag;
  p.scale = d->scale;
  p.bias = d->bias;
  p.res = d->residual;
  p.y = d->y;
  p.y2 = d->y2;
  p.N = d->N;
  p.Cin = d->Cin;
  p.H = d->H;
  p.W = d->W;
  p.Cout = d->Cout;
  p.k = d->k;
  p.stride = d->stride;
  p.pad = pad;
  p.Ho = Ho;
  p.Wo = Wo;
  p.M = (int)M;
  p.cin_chunks = (d->Cin + BK - 1) / BK;
  p.KT = d->k * d->k * p.cin_chunks;
  p.act = d->act;
  p.act2 = d->act2;
  p.res_mode = d->residual ? d->res_mode : 0;
  p.post = SSDK_ACT_NONE;
  if (d->residual && (d->res_mode & 2)) {  // y = act(conv + residual): the staged value is linear
    if (d->act != SSDK_ACT_NONE && d->act != SSDK_ACT_RELU && d->act != SSDK_ACT_RELU6) {
      set_error("conv: only relu / relu6 can follow the residual add");
      return SSDK_E_BADARG;
    }
    p.post = d->act;
    p.act = SSDK_ACT_NONE;
    p.act2 = SSDK_ACT_NONE;
  }
  if (d->residual && (d->res_mode & 1) && ((Ho | Wo) & 1)) {
    set_error("conv: a half-resolution residual needs even output dims (%dx%d)", Ho, Wo);
    return SSDK_E_BADARG;
  }
  p.split = split;
  p.in_layout = d->in_layout;
  p.out_layout = d->out_layout;
  p.ksplits = 1;
  p.kt_per = p.KT;
  p.slabs = nullptr;
  p.counters = nullptr;
  if (d->k == 3 && d->stride == 1 && !g_underfill_ok) {  // maps of 16 .. 64 pixels with a long K: halo tiles + split-K
    size_t hb = 0;
    const int hs = halo_splitk_plan(d->N, d->Cin, Ho, Wo, d->Cout, d->out_layout == SSDK_LAYOUT_NCHW, &hb);
    if (hs > 1 && Ho * Wo >= halo_splitk_min_pixels() && workspace && workspace_bytes >= hb && !((uintptr_t)workspace & 255)) {
      ConvParams q = p;
      q.ksplits = hs;
      q.counters = (unsigned*)workspace;
      q.slabs = (float*)((char*)workspace + 4096);
      const int rc = launch_conv3x3_halo(q, d->dtype, stream, true);
      if (rc != 1) return rc;
    }
  }
  if (launch_conv_smallmap(p, d->dtype, stream) == 0) return check_launch("conv_smallmap_kernel");  // maps of <= 64 pixels
  {
    const int rc = launch_conv_gemmp(p, d->dtype, stream);  // 1x1, long K, enough tiles: persistent GEMM
    if (rc != 1) return rc;
  }
  if (launch_conv_pwflow(p, d->dtype, stream) == 0) return check_launch("pwflow_kernel");  // 1x1, short K, large maps: streaming
  // Side-lane ops (small heads running next to the extras chain) prefer the halo kernel however few tiles they have:
  // an underfilled grid is free there, and unlike the split-K kernels it has no agent-scope fences, which slow down
  // every kernel running concurrently (measured: split-K heads on the side lane +0.6 %, halo heads +2.7 %).
  if (g_underfill_ok) {
    const int rc = launch_conv3x3_halo(p, d->dtype, stream, true);
    if (rc != 1) return rc;
  }
  {
    int ws = 1, wk = p.KT, mt = 0, ntl = 0;
    if (wave_plan(M, d->Cin, d->Cout, d->k, &ws, &wk, &mt, &ntl)) {
      const size_t need = ws > 1 ? 4096 + (size_t)mt * ntl * ws * 4096 : 0;
      if (ws > 1 && !(workspace && workspace_bytes >= need && !((uintptr_t)workspace & 255))) {
        ws = 1;  // no scratch: unsplit
        wk = p.KT;
      }
      p.ksplits = ws;
      p.kt_per = wk;
      if (ws > 1) {
        p.counters = (unsigned*)workspace;
        p.slabs = (float*)((char*)workspace + 4096);
      }
      const unsigned waves = (unsigned)mt * ntl * ws;
      if (d->dtype == SSDK_BF16)
        hipLaunchKernelGGL((conv_wave_kernel<SSDK_BF16>), dim3((waves + 3) / 4), dim3(256), 0, stream, p, mt, ntl);
      else
        hipLaunchKernelGGL((conv_wave_kernel<SSDK_F16>), dim3((waves + 3) / 4), dim3(256), 0, stream, p, mt, ntl);
      return check_launch("conv_wave_kernel");
    }
  }
  {
    int splits = 1, kt_per = p.KT;
    const size_t need = splitk_ws_bytes(M, d->Cin, d->Cout, d->k, &splits, &kt_per);
    const int tiles = ((int)((M + BM - 1) / BM)) * ((d->Cout + gemm_bn(d->Cout) - 1) / gemm_bn(d->Cout));
    // the workspace is optional: without one (or a too small one) the layer simply runs unsplit
    if (splits > 1 && workspace && workspace_bytes >= need && !((uintptr_t)workspace & 255) && tiles * 4 <= 4096) {
      p.ksplits = splits;
      p.kt_per = kt_per;
      p.counters = (unsigned*)workspace;  // zero on entry (allocated zeroed; re-armed by every last arriver)
      p.slabs = (float*)((char*)workspace + 4096);
    }
  }
  if (launch_conv3x3_short(p, d->dtype, stream) == 0) return check_launch("conv3x3_short_kernel");  // short K, two workgroups per CU
  {
    // 3x3 stride-1 layers with enough tiles.  A split planned above for conv_gemm_kernel carries over (few output pixels, long
    // K: the 10x10 FPN levels) only if the workspace also holds the HALO kernel's slabs; otherwise the halo kernel runs unsplit
    // (or declines: too few tiles) and conv_gemm_kernel takes the layer with its own split.
    ConvParams q = p;
    if (q.ksplits > 1) {
      const size_t hb = halo_ws_bytes(d->N, d->Cin, Ho, Wo, d->Cout, d->out_layout == SSDK_LAYOUT_NCHW, q.ksplits);
      if (!hb || workspace_bytes < hb) {
        q.ksplits = 1;
        q.kt_per = q.KT;
        q.counters = nullptr;
        q.slabs = nullptr;
      }
    }
    const int rc = launch_conv3x3_halo(q, d->dtype, stream, false);
    if (rc != 1) return rc;
  }
  return d->dtype == SSDK_BF16 ? launch_gemm<SSDK_BF16>(p, stream) : launch_gemm<SSDK_F16>(p, stream);
}

// Runs a whole pre-planned network (array of descriptors, buffers already assigned by the host) with one
// call: one launch per layer on the caller's stream, ~2 us of host time per layer, hipGraph-capturable.
extern "C" int ssdk_conv_sequence(const ssdk_conv_desc* descs, int n, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  if (!descs || n < 0) {
    set_error("conv_sequence: bad arguments");
    return SSDK_E_BADARG;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = ssdk_conv(&descs[i], workspace, workspace_bytes, stream);
    if (rc) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", ssdk_last_error());
      set_error("conv_sequence: layer %d of %d: %s", i, n, msg);
      return rc;
    }
  }
  return SSDK_OK;
}

// Per-op timing of ssdk_run_ops (tools / bench.py layer table): one hipEvent before every op and one after the
// last, on the caller's stream; read back with ssdk_get_op_timings after the stream has been synchronised.
// Events, like the side lane below, belong to the caller's context (ssdk_ctx.h).
extern "C" int ssdk_ctx_set_op_profiling(ssdk_ctx* ctx, int enable) {
  if (int rc = ssdk::ctx_enter(ctx)) return rc;
  if (enable && !ctx->op_ev_ready) {
    for (int i = 0; i <= kSsdkMaxProfOps; ++i)
      if (hipEventCreate(&ctx->op_ev[i]) != hipSuccess) {
        set_error("set_op_profiling: hipEventCreate failed");
        return SSDK_E_LAUNCH;
      }
    ctx->op_ev_ready = true;
  }
  ctx->op_prof = enable ? 1 : 0;
  ctx->op_n = 0;
  return SSDK_OK;
}
extern "C" int ssdk_set_op_profiling(int enable) { return ssdk_ctx_set_op_profiling(ssdk::default_ctx(), enable); }

extern "C" int ssdk_ctx_get_op_timings(ssdk_ctx* ctx, float* ms, const char** kernels, int n_max) {
  if (int rc = ssdk::ctx_enter(ctx)) return rc;
  if (!ms || n_max < ctx->op_n) {
    set_error("get_op_timings: need room for %d ops", ctx->op_n);
    return SSDK_E_BADARG;
  }
  for (int i = 0; i < ctx->op_n; ++i) {
    if (hipEventElapsedTime(&ms[i], ctx->op_ev[i], ctx->op_ev[i + 1]) != hipSuccess) {
      set_error("get_op_timings: events of op %d not complete (synchronise the stream first)", i);
      return SSDK_E_LAUNCH;
    }
    if (kernels) kernels[i] = ctx->op_kernel[i];
  }
  return ctx->op_n;
}
extern "C" int ssdk_get_op_timings(float* ms, const char** kernels, int n_max) {
  return ssdk_ctx_get_op_timings(ssdk::default_ctx(), ms, kernels, n_max);
}

// Fork/join onto a side stream owned by the CONTEXT: ops whose `lane` is 1 (the multibox heads: they only depend on
// their feature map, not on each other or on the layers that follow) run concurrently with the main chain.
// The tiny tail layers (extras at 8x8..1x1, heads of the last levels) are launch/latency bound; overlapped with
// the big head GEMMs they disappear from the critical path.  Everything is joined back onto the caller's stream
// before returning, so the caller sees ordinary stream semantics (and the whole call is graph-capturable).
// A failed event / wait is an error (SSDK_E_LAUNCH): dropping one silently would drop an ordering edge.
static bool side_wanted(const ssdk_ctx* ctx) {
  // Round 1 (small heads on split-K / one-wave kernels of 20-65 us each): +2.7 % with levels 2..5 on the side lane.
  // Round 2 (conv_smallmap_kernel, fused extras): the same heads take 17-53 us and the in-line plan is 1.1 % FASTER
  // than the forked one (fork / join events, two queues competing for the same CUs) -- off by default;
  // SSDK_SIDE_STREAM=1 or ssdk_ctx_set_side_lane(ctx, 1) turns it on.
  if (ctx->side_lane >= 0) return ctx->side_lane != 0;
  const char* e = getenv("SSDK_SIDE_STREAM");
  return e && *e && atoi(e) != 0;
}

static int side_init(ssdk_ctx* ctx) {
  if (ctx->side_ready) return SSDK_OK;
  if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) goto fail;
  for (int i = 0; i < 32; ++i)
    if (hipEventCreateWithFlags(&ctx->fork[i], hipEventDisableTiming) != hipSuccess) goto fail;
  if (hipEventCreateWithFlags(&ctx->join, hipEventDisableTiming) != hipSuccess) goto fail;
  ctx->side_ready = true;
  return SSDK_OK;
fail:
  (void)hipGetLastError();
  set_error("run_ops: cannot create the side stream / its events");
  return SSDK_E_LAUNCH;
}

static int edge(hipStream_t from, hipEvent_t e, hipStream_t to) {  // work recorded on `from` so far precedes `to`
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
    (void)hipGetLastError();
    set_error("run_ops: stream fork/join failed");
    return SSDK_E_LAUNCH;
  }
  return SSDK_OK;
}

// Grouped launch of neighbouring small-map layers (launch_conv_smallmap_group): the ConvParams of a plan op that is a plain
// dense 3x3 / stride-1 / NHWC-input convolution without residual -- anything else (and anything ssdk_conv would reject) is
// not a member and goes through ssdk_conv, with its checks and error messages, as before.
static bool group_member_params(const ssdk_conv_desc* d, ConvParams* p) {
  if (!d->x || !d->w || !d->w_frag || !d->bias || !d->y || d->residual) return false;
  if ((d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) || d->k != 3 || d->stride != 1 || d->groups != 1) return false;
  if (d->N < 1 || d->Cin < 8 || (d->Cin % 8) || d->H < 1 || d->W < 1 || d->Cout < 1 || d->in_layout != LAYOUT_NHWC) return false;
  if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->w_frag | (uintptr_t)d->y | (uintptr_t)d->y2) & 15) return false;
  const int split = (d->y2 && d->split > 0 && d->split < d->Cout) ? d->split : d->Cout;
  if (d->out_layout == LAYOUT_NHWC && split != d->Cout) return false;
  if ((long)d->N * d->H * d->W >= (1l << 31)) return false;
  memset(p, 0, sizeof(*p));
  p->x = d->x;
  p->w = d->w;
  p->w_frag = d->w_frag;
  p->scale = d->scale;
  p->bias = d->bias;
  p->y = d->y;
  p->y2 = d->y2;
  p->N = d->N;
  p->Cin = d->Cin;
  p->H = d->H;
  p->W = d->W;
  p->Cout = d->Cout;
  p->k = 3;
  p->stride = 1;
  p->pad = 1;
  p->Ho = d->H;
  p->Wo = d->W;
  p->M = d->N * d->H * d->W;
  p->cin_chunks = (d->Cin + BK - 1) / BK;
  p->KT = 9 * p->cin_chunks;
  p->act = d->act;
  p->act2 = d->act2;
  p->split = split;
  p->in_layout = d->in_layout;
  p->out_layout = d->out_layout;
  p->post = SSDK_ACT_NONE;
  p->ksplits = 1;
  p->kt_per = p->KT;
  return true;
}

// Byte ranges of a group member: input [x, +N*H*W*Cin), outputs y (split channels, or all) and y2 (the rest).
struct GroupRange {
  uintptr_t lo, hi;
  bool overlaps(const GroupRange& o) const { return lo < o.hi && o.lo < hi; }
};
static inline GroupRange group_range(const void* ptr, size_t bytes) {
  return GroupRange{(uintptr_t)ptr, (uintptr_t)ptr + (ptr ? bytes : 0)};
}
static bool group_members_independent(const ConvParams& a, const ConvParams& b) {
  const size_t es = 2;  // bf16 / f16 only (group_member_params)
  auto ranges = [&](const ConvParams& p, GroupRange (&r)[3]) {
    const size_t px = (size_t)p.N * p.H * p.W;
    r[0] = group_range(p.x, px * p.Cin * es);
    r[1] = group_range(p.y, px * (size_t)(p.y2 ? p.split : p.Cout) * es);
    r[2] = group_range(p.y2, p.y2 ? px * (size_t)(p.Cout - p.split) * es : 0);
  };
  GroupRange ra[3], rb[3];
  ranges(a, ra);
  ranges(b, rb);
  for (int i = 0; i < 3; ++i)       // anything of b against a's outputs
    for (int j = 1; j < 3; ++j)
      if (rb[i].overlaps(ra[j])) return false;
  for (int j = 1; j < 3; ++j)       // b's outputs against a's input
    if (rb[j].overlaps(ra[0])) return false;
  return true;
}

extern "C" int ssdk_run_ops_ctx(ssdk_ctx* ctx, const ssdk_op* ops, int n, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (int rc = ssdk::ctx_enter(ctx)) return rc;
  if (!ops || n < 0) {
    set_error("run_ops: bad arguments");
    return SSDK_E_BADARG;
  }
  hipStream_t main_s = (hipStream_t)stream;
  const bool prof = ctx->op_prof && ctx->op_ev_ready && n <= kSsdkMaxProfOps;
  if (prof) ctx->op_n = 0;
  // side-lane ops fork off the main stream once per RUN of consecutive side ops (a chain of small-level tower layers is one
  // run: one fork behind its producers, the join at the end of the plan); 32 fork events per context
  int n_side = 0, n_runs = 0;
  for (int i = 0; i < n; ++i) {
    n_side += ops[i].lane != 0 ? 1 : 0;
    n_runs += (ops[i].lane != 0 && (i == 0 || ops[i - 1].lane == 0)) ? 1 : 0;
  }
  // concurrent lanes need disjoint split-K scratch: the side lane gets the upper half of the workspace
  bool use_side = n_side > 0 && n_runs <= 32 && !prof && side_wanted(ctx);
  if (use_side)
    if (int rc = side_init(ctx)) return rc;
  char* ws_main = (char*)workspace;
  size_t ws_main_bytes = workspace_bytes;
  char* ws_side = nullptr;
  size_t ws_side_bytes = 0;
  if (use_side && workspace) {
    const size_t half = (workspace_bytes / 2) & ~(size_t)255;
    ws_main_bytes = half;
    ws_side = ws_main + half;
    ws_side_bytes = workspace_bytes - half;
  }
  int forks = 0;
  for (int i = 0; i < n; ++i) {
    if (prof && hipEventRecord(ctx->op_ev[i], main_s) != hipSuccess) {
      set_error("run_ops: hipEventRecord failed");
      return SSDK_E_LAUNCH;
    }
    const bool side = use_side && ops[i].lane != 0;
    hipStream_t st = main_s;
    void* w = ws_main;
    size_t wb = ws_main_bytes;
    int rc = SSDK_OK;
    if (side) {
      if (i == 0 || ops[i - 1].lane == 0) {  // first op of a run: everything recorded so far (its producers) precedes it
        rc = edge(main_s, ctx->fork[forks], ctx->side);
        ++forks;
      }
      constexpr int env_same = 0;  // (round 6: the SSDK_SIDE_DEBUG switch is gone, its A/B is settled)
      st = env_same ? main_s : ctx->side;  // (debug: the side lane's bookkeeping without its concurrency)
      w = ws_side;
      wb = ws_side_bytes;
    }
    if (!rc && ops[i].kind == SSDK_OP_CONV) {
      // neighbouring small-map layers (of the same lane) that do not read each other's outputs: one launch
      ConvParams gp[kSmallmapGroupMax];
      int m = 0;
      while (m < kSmallmapGroupMax && i + m < n && ops[i + m].kind == SSDK_OP_CONV && (use_side && ops[i + m].lane != 0) == side &&
             ops[i + m].conv.dtype == ops[i].conv.dtype && group_member_params(&ops[i + m].conv, &gp[m])) {
        // Members run as ONE kernel in no particular order: a candidate joins only if none of its byte ranges (input,
        // outputs) overlaps an earlier member's OUTPUT ranges (read-after-write, write-after-write) and its outputs do
        // not overlap an earlier member's INPUT range (write-after-read: the plan arena hands a buffer out again as soon
        // as its last reader is recorded, so the op behind a tower's head may write the buffer the head still reads).
        bool indep = true;
        for (int a = 0; a < m && indep; ++a) indep = group_members_independent(gp[a], gp[m]);
        if (!indep) break;
        ++m;
      }
      if (m >= 2) {
        ssdk::lds_poison(st);
        if (launch_conv_smallmap_group(gp, m, ops[i].conv.dtype, st) == 0) {
          rc = check_launch("conv_smallmap_group_kernel");
          if (!rc) {
            for (int a = 0; a < m; ++a) {  // (profiling: the group's time lands on its first op, the others read ~0)
              if (prof) ctx->op_kernel[i + a] = ssdk_last_kernel();
              if (prof && a > 0 && hipEventRecord(ctx->op_ev[i + a], main_s) != hipSuccess) {
                set_error("run_ops: hipEventRecord failed");
                return SSDK_E_LAUNCH;
              }
            }
            i += m - 1;
            continue;
          }
        }
      }
    }
    if (!rc) {
      ssdk::lds_poison(st);
      if (ops[i].kind == SSDK_OP_CONV) {
        // lane 1 (leaf heads): next to the main chain an underfilled grid is free, so on the side stream the kernel choice differs
        // from the in-line one.  lane 2 (chains of small-level layers): ALWAYS the choice for an op that may underfill the chip
        // (an unsplit halo launch of 32 workgroups instead of 128 split-K ones that each take a whole CU's LDS and exchange
        // 640 KB of slabs per tile: FPN-ResNet50@640 3 225 -> 3 281 img/s) -- tied to the op's tag, not to whether the side
        // stream is in use, so the outputs are the same bits either way (SSDK_LANE2_UNDERFILL=0: the in-line choice).
        constexpr int env_l2 = 1;  // (round 6: the SSDK_LANE2_UNDERFILL switch is gone, its A/B is settled)
        g_underfill_ok = (side && ops[i].lane == 1) || (env_l2 && ops[i].lane == 2);
        rc = ssdk_conv(&ops[i].conv, w, wb, st);
        g_underfill_ok = false;
      }
      else if (ops[i].kind == SSDK_OP_MBCONV) rc = ssdk_mbconv(&ops[i].mb, st);
      else if (ops[i].kind == SSDK_OP_FUSE) rc = ssdk_fuse(&ops[i].fuse, st);
      else if (ops[i].kind == SSDK_OP_STEM7) rc = ssdk_conv_stem7(&ops[i].stem, st);
      else if (ops[i].kind == SSDK_OP_POOL) rc = ssdk_maxpool3x3s2(&ops[i].pool, st);
      else if (ops[i].kind == SSDK_OP_XPAIR) rc = ssdk_xpair(&ops[i].xpair, st);
      else {
        set_error("unknown op kind %d", ops[i].kind);
        rc = SSDK_E_BADARG;
      }
    }
    if (rc) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", ssdk_last_error());
      set_error("run_ops: op %d of %d: %s", i, n, msg);
      if (forks) (void)edge(ctx->side, ctx->join, main_s);  // never leave the side stream dangling
      return rc;
    }
    if (prof) ctx->op_kernel[i] = ssdk_last_kernel();
    static const int env_trace = getenv("SSDK_OPS_TRACE") ? atoi(getenv("SSDK_OPS_TRACE")) : 0;
    if (env_trace) {  // debug: one line per op, synchronised (a faulting kernel is the line that never gets its "ok")
      fprintf(stderr, "[run_ops] op %d/%d kind %d lane %d %s", i, n, ops[i].kind, ops[i].lane, ssdk_last_kernel());
      if (ops[i].kind == SSDK_OP_CONV)
        fprintf(stderr, " x=%p y=%p y2=%p N=%d %dx%d Cin=%d Cout=%d k=%d s=%d out=%d", ops[i].conv.x, ops[i].conv.y, ops[i].conv.y2,
                ops[i].conv.N, ops[i].conv.H, ops[i].conv.W, ops[i].conv.Cin, ops[i].conv.Cout, ops[i].conv.k, ops[i].conv.stride,
                ops[i].conv.out_layout);
      fprintf(stderr, " ...");
      const hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
    }
  }
  if (forks)
    if (int rc = edge(ctx->side, ctx->join, main_s)) return rc;
  if (prof) {
    if (hipEventRecord(ctx->op_ev[n], main_s) != hipSuccess) {
      set_error("run_ops: hipEventRecord failed");
      return SSDK_E_LAUNCH;
    }
    ctx->op_n = n;
  }
  return SSDK_OK;
}

extern "C" int ssdk_run_ops(const ssdk_op* ops, int n, void* workspace, size_t workspace_bytes, void* stream) {
  return ssdk_run_ops_ctx(ssdk::default_ctx(), ops, n, workspace, workspace_bytes, stream);
}

extern "C" int ssdk_conv_bn_act(const void* x, const void* w, const float* scale, const float* bias, int N,
                                int Cin, int H, int W, int Cout, int k, int stride, int act, int dtype,
                                int out_dtype, void* y, void* workspace, size_t workspace_bytes, void* stream) {
  if (out_dtype != dtype) {
    set_error("conv_bn_act: out_dtype must equal dtype");
    return SSDK_E_BADARG;
  }
  ssdk_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = x;
  d.w = w;
  d.scale = scale;
  d.bias = bias;
  d.y = y;
  d.N = N;
  d.Cin = Cin;
  d.H = H;
  d.W = W;
  d.Cout = Cout;
  d.k = k;
  d.stride = stride;
  d.groups = 1;
  d.act = act;
  d.act2 = act;
  d.split = Cout;
  d.dtype = dtype;
  d.in_layout = LAYOUT_NHWC;
  d.out_layout = LAYOUT_NHWC;
  return ssdk_conv(&d, workspace, workspace_bytes, stream);
}
