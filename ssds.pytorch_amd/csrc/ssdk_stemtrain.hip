// ssdk_stemtrain.hip -- the network's FIRST convolution inside the training step: 3x3 / stride 2 / pad 1 on an image of
// Cin <= 3 channels (9 * Cin <= 27 taps) to Cout <= 32 channels (torchvision MobileNetV2 features[0][0] behind
// nets/mobilenet.py:180-192; reference training step pipeline_anchor_apex.py:37-72).  Forward and weight gradient; an image has no
// input gradient.  16-bit NCHW tensors as autograd hands them over, fp32 master weights.
//
// Why its own kernels: it was the last convolution of the step on the vendor library -- an implicit-GEMM forward (140 us), an
// implicit-GEMM weight gradient (185 us), three NCHW <-> NHWC transposes of the 268 MB output / its gradient (120 us) and cast /
// zero passes at SSD-MobileNetV2@512, batch 64 (profiles/r06_train_kernel_split_final_v3.txt).  The layer is a STREAM: 27 taps,
// 100 MB of image in, 268 MB out (forward), the same two tensors in (weight gradient): ~60 us of HBM time each.
//
//   forward      stem_fwd_mfma_kernel (even widths): the layer as a GEMM on the matrix cores with k = 4 r + j -- r one of the
//                <= 9 (channel, kernel row) pairs, j the columns 2 ox - 2 .. 2 ox + 1 -- so that a lane's eight consecutive k are
//                two 8-byte loads of an image row (no gather), the weights a loop-invariant A operand in 16 VGPRs; see the
//                kernel.  Odd widths: stem_fwd_kernel, a thread per output pixel pair with the weights read from LDS (slow:
//                kept as the fallback only).
//   weight grad  dW[co][tap] = sum over pixels dy[co][p] patch[tap][p] contracts over PIXELS, which are contiguous in dy
//                (NCHW) and stride-2 in x: v_mfma_f32_16x16x32 with A = dy (16 channels x 32 pixels: one 16-byte load per
//                lane) and B = patch (16 taps x 32 pixels: 32 + 4 bytes of an image row per lane, the even / odd columns
//                picked with v_perm_b32).  A wave walks whole output rows; the four waves of a workgroup are added in wave
//                order, workgroup partials go to the workspace and stem_wgrad_reduce_kernel adds them in index order: no float
//                atomics, bit-reproducible.
#include "ssdk_conv_common.h"

namespace ssdk {

struct StemTrainParams {
  const u16* x;     // [N, Cin, H, W]
  const float* w;   // forward: [Cout, Cin, 3, 3] fp32 master weights
  u16* y;           // forward: [N, Cout, Ho, Wo]
  const u16* dy;    // weight gradient: [N, Cout, Ho, Wo]
  float* part;      // weight gradient: [partials][32 * 32]
  float* dw;        // weight gradient: [Cout, Cin, 3, 3]
  int N, Cin, H, W, Cout, Ho, Wo;
  int rows_per_wave;  // weight gradient: output rows (n, oy) per wave
  int partials;
};

// ---- forward ---------------------------------------------------------------------------------------------------------------
template <int DT, int CIN, bool FAST>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemTrainParams p) {
  // weights -> LDS as [tap][co] fp32 carrying the tensor dtype's rounding (what autocast's cast of the parameter does), zero
  // padded.  (Through SCALAR loads -- uniform addresses, SGPR operands of v_pk_fma_f32 -- the kernel took 460 us: 54 blocking
  // 64-byte scalar loads per wave; LDS broadcast reads: see DESIGN 6.)
  __shared__ __attribute__((aligned(16))) float swq[27 * 32];
  {
    const int taps = p.Cin * 9;
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
      const int tap = i >> 5, co = i & 31;
      float v = 0.f;
      if (tap < taps && co < p.Cout) v = bits16_to_f32<DT>(f32_to_bits16<DT>(p.w[(size_t)co * taps + tap]));
      swq[i] = v;
    }
  }
  __syncthreads();
  const int Wo2 = (p.Wo + 1) >> 1;                       // pixel pairs per output row
  const int pair = blockIdx.x * 256 + (int)threadIdx.x;  // pair index inside the image
  const int n = blockIdx.y;
  if (pair >= p.Ho * Wo2) return;
  const int oy = pair / Wo2, oxp = pair - oy * Wo2;
  const int ox0 = 2 * oxp, c0 = 4 * oxp;  // input columns c0 - 1 .. c0 + 3
  // FAST (W a multiple of 4, 8-byte aligned tensor): columns c0 .. c0 + 3 are one aligned 8-byte load inside the row.
  // Every load is UNCONDITIONAL on a clamped address and masked afterwards: the 18 loads of a thread are in flight together
  // (as first written each sat in its own branch behind an s_waitcnt: 462 us per launch instead of ~100).
  // the window: v[ci][ky][0..4] = x[ci][2 oy + ky - 1][c0 - 1 + j] as fp32 (zero outside the image)
  u32 rl[CIN][3], r0[CIN][3], r1[CIN][3];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky - 1;
      const int iyc = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
      const u16* r = p.x + (((size_t)n * p.Cin + ci) * p.H + iyc) * p.W;
      if constexpr (FAST) {
        const uint2 q = *reinterpret_cast<const uint2*>(r + c0);
        r0[ci][ky] = q.x;
        r1[ci][ky] = q.y;
      } else {
        const int last = p.W - 1;
        const u32 e0 = r[c0 < last ? c0 : last], e1 = r[c0 + 1 < last ? c0 + 1 : last];
        const u32 e2 = r[c0 + 2 < last ? c0 + 2 : last], e3 = r[c0 + 3 < last ? c0 + 3 : last];
        r0[ci][ky] = (c0 < p.W ? e0 : 0u) | ((c0 + 1 < p.W ? e1 : 0u) << 16);
        r1[ci][ky] = (c0 + 2 < p.W ? e2 : 0u) | ((c0 + 3 < p.W ? e3 : 0u) << 16);
      }
      rl[ci][ky] = r[c0 > 0 ? c0 - 1 : 0];
    }
  float v[CIN][3][5];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky - 1;
      const bool row = (unsigned)iy < (unsigned)p.H;
      const u32 lo = row ? r0[ci][ky] : 0u, hi = row ? r1[ci][ky] : 0u, left = (row && c0 > 0) ? rl[ci][ky] : 0u;
      v[ci][ky][0] = bits16_to_f32<DT>(left);
      v[ci][ky][1] = bits16_to_f32<DT>(lo & 0xffffu);
      v[ci][ky][2] = bits16_to_f32<DT>(lo >> 16);
      v[ci][ky][3] = bits16_to_f32<DT>(hi & 0xffffu);
      v[ci][ky][4] = bits16_to_f32<DT>(hi >> 16);
    }
  // acc[j] = (channel 2j, channel 2j + 1) of pixel 0 / pixel 1
  f32x2 a0[16], a1[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a0[j] = a1[j] = f32x2{0.f, 0.f};
  const f32x2* wq = reinterpret_cast<const f32x2*>(swq);
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tap = (ci * 3 + ky) * 3 + kx;
        const float x0 = v[ci][ky][kx], x1 = v[ci][ky][kx + 2];
        const f32x2 b0 = {x0, x0}, b1 = {x1, x1};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const f32x2 w2 = wq[tap * 16 + j];
          a0[j] = __builtin_elementwise_fma(w2, b0, a0[j]);
          a1[j] = __builtin_elementwise_fma(w2, b1, a1[j]);
        }
      }
  const size_t plane = (size_t)p.Ho * p.Wo;
  u16* y = p.y + (size_t)n * p.Cout * plane + (size_t)oy * p.Wo + ox0;
  const bool two = ox0 + 1 < p.Wo;
  const bool vec = (p.Wo & 1) == 0 && (((uintptr_t)p.y) & 3u) == 0;  // uniform (an even row has no single pixel at its end)
#pragma unroll
  for (int j = 0; j < 16; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int co = 2 * j + h;
      if (co < p.Cout) {  // uniform
        const float f0 = h ? a0[j][1] : a0[j][0], f1 = h ? a1[j][1] : a1[j][0];
        if (vec) {
          *reinterpret_cast<u32*>(y + (size_t)co * plane) = pack2_16<DT>(f0, f1);
        } else {
          y[(size_t)co * plane] = (u16)f32_to_bits16<DT>(f0);
          if (two) y[(size_t)co * plane + 1] = (u16)f32_to_bits16<DT>(f1);
        }
      }
    }
}

// ---- forward on the matrix cores (even W >= 4) ------------------------------------------------------------------------------------
// y[co][px] = sum_k A[co][k] B[k][px] with k = 4 r + j: r = 3 ci + ky one of the <= 9 (channel, kernel row) pairs, j = 0 .. 3 the
// columns 2 ox - 2 .. 2 ox + 1 of image row 2 oy + ky - 1 (j = 0 carries a zero weight, j = 1 .. 3 are kx = 0 .. 2).  The B
// operand of v_mfma_f32_16x16x32 wants eight consecutive k of one pixel per lane: lane group g holds rows 2g and 2g + 1 = TWO
// 8-byte loads at a 4-byte aligned address -- no gather, no LDS.  Rows 8 .. 15 (only row 8 exists for three channels) are a
// second k-step whose loads only lane group 0 makes.  The weights are the A operand: 16 VGPRs, built once per wave.
// A wave iteration is 32 neighbouring pixels of one output row as two tiles of the EVEN / ODD pixels, so a lane ends up with two
// neighbouring pixels of four channels per tile pair: one 4-byte store per channel, 64 contiguous bytes per 16 lanes.
// (The VALU form below needs the 864 weights per thread again and again: 460 us through scalar loads, 2.2 ms through LDS
//  broadcast reads of 16 bytes; this one: see DESIGN 6.)
template <int DT>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const StemTrainParams p) {
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const int fr = (int)(lane & 15u), fg = (int)(lane >> 4);
  const int rows = 3 * p.Cin, taps = 9 * p.Cin;
  u32x4 Aw[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = 16 * t + fr;
      u32 h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = ks * 8 + 2 * fg + (e >> 2), j = e & 3;
        const bool ok = j > 0 && r < rows && co < p.Cout;
        const float wv = p.w[ok ? (size_t)co * taps + r * 3 + (j - 1) : 0];
        h[e] = ok ? f32_to_bits16<DT>(wv) : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) Aw[t][ks][q] = h[2 * q] | (h[2 * q + 1] << 16);
    }
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int total_rows = p.N * p.Ho;
  const int wid = (int)(blockIdx.x * 4u + wave);
  const int r0 = wid * p.rows_per_wave, r1 = r0 + p.rows_per_wave < total_rows ? r0 + p.rows_per_wave : total_rows;
  const bool two_steps = rows > 8;                                             // uniform
  const bool pack = (p.Wo & 1) == 0 && (((uintptr_t)p.y) & 3u) == 0;           // uniform: 4-byte stores of pixel pairs
  const int rr[3] = {2 * fg, 2 * fg + 1, 8};                                   // this lane's rows of k-step 0 / of k-step 1 (group 0)
  for (int r = r0; r < r1; ++r) {  // wave-uniform
    const int n = r / p.Ho, oy = r - n * p.Ho;
    const u16* xrow[3];
    bool xok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int ci = rr[q] / 3, ky = rr[q] - 3 * ci, iy = 2 * oy + ky - 1;
      xok[q] = rr[q] < rows && (unsigned)iy < (unsigned)p.H && (q < 2 || fg == 0);
      xrow[q] = p.x + (((size_t)n * p.Cin + (xok[q] ? ci : 0)) * p.H + (xok[q] ? iy : 0)) * p.W;
    }
    u16* yrow = p.y + (size_t)n * p.Cout * plane + (size_t)oy * p.Wo;
    // the loads of a tile: [tile: even / odd pixel][row]; the NEXT tile of the row is requested before this one is multiplied
    auto load_tile = [&](int ox0, uint2 (&raw)[2][3]) {
      int cA = 2 * (ox0 + 2 * fr) - 2, cB = cA + 2;  // first column of the even / odd pixel's window
      cA = cA < 0 ? 0 : (cA > p.W - 4 ? p.W - 4 : cA);
      cB = cB > p.W - 4 ? p.W - 4 : cB;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < 2 || two_steps) {
          raw[0][q] = *reinterpret_cast<const uint2*>(xrow[q] + cA);
          raw[1][q] = *reinterpret_cast<const uint2*>(xrow[q] + cB);
        } else {
          raw[0][q] = raw[1][q] = make_uint2(0u, 0u);
        }
      }
    };
    uint2 raw[2][3], nxt[2][3] = {};
    load_tile(0, raw);
    for (int ox0 = 0; ox0 < p.Wo; ox0 += 32) {
      const bool edge = ox0 == 0 && fr == 0;  // pixel 0: columns -2, -1 are the zero padding
      if (ox0 + 32 < p.Wo) load_tile(ox0 + 32, nxt);  // (uniform)
      f32x4 acc[2][2];  // [tile][co tile]
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        uint2 v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          v[q] = raw[T][q];
          if (T == 0 && edge) v[q] = make_uint2(0u, v[q].x);
          if (!xok[q]) v[q] = make_uint2(0u, 0u);
        }
        const u32x4 B0 = {v[0].x, v[0].y, v[1].x, v[1].y}, B1 = {v[2].x, v[2].y, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[T][t] = mfma16<DT>(Aw[t][0], B0, f32x4{0.f, 0.f, 0.f, 0.f});
          if (two_steps) acc[T][t] = mfma16<DT>(Aw[t][1], B1, acc[T][t]);
        }
      }
      // D[m = 4 fg + i][n = fr] of (tile T, co tile t) = y[co = 16 t + 4 fg + i][pixel ox0 + 2 fr + T]
      const int ox = ox0 + 2 * fr;
      if (ox < p.Wo) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = 16 * t + 4 * fg + i;
            if (co < p.Cout) {
              u16* dst = yrow + (size_t)co * plane + ox;
              if (pack) {
                *reinterpret_cast<u32*>(dst) = pack2_16<DT>(acc[0][t][i], acc[1][t][i]);
              } else {
                dst[0] = (u16)f32_to_bits16<DT>(acc[0][t][i]);
                if (ox + 1 < p.Wo) dst[1] = (u16)f32_to_bits16<DT>(acc[1][t][i]);
              }
            }
          }
      }
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int q = 0; q < 3; ++q) raw[T][q] = nxt[T][q];
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------
// B fragment of one k-step: the lane's tap (ci, ky, kx) at the eight output pixels ox .. ox + 7 of row oy = the columns
// 2 (ox + j) + kx - 1 of image row 2 oy + ky - 1: the even (kx = 1) or odd (kx = 0: from the dword in front, kx = 2) halves of
// the nine dwords that cover columns 2 ox - 2 .. 2 ox + 15.  Loads are unconditional on clamped addresses, masked afterwards.
// FAST (W a multiple of 16, 16-byte aligned tensor): an eight-pixel group lies inside the row or completely outside it.
template <int DT, bool FAST>
__device__ __forceinline__ void stem_patch_load(const u16* row, int ox, int W, int Wo, u32 (&d)[9]) {
  // d[0] = columns (cb - 2, cb - 1), d[1 + i] = columns (cb + 2i, cb + 2i + 1), cb = 2 ox
  if constexpr (FAST) {
    const int oxc = ox < Wo ? ox : Wo - 8, cb = 2 * oxc;
    const u32x4 q0 = *reinterpret_cast<const u32x4*>(row + cb), q1 = *reinterpret_cast<const u32x4*>(row + cb + 8);
    const u32 left = row[cb > 0 ? cb - 1 : 0];
    d[0] = cb > 0 ? left << 16 : 0u;
    d[1] = q0[0];
    d[2] = q0[1];
    d[3] = q0[2];
    d[4] = q0[3];
    d[5] = q1[0];
    d[6] = q1[1];
    d[7] = q1[2];
    d[8] = q1[3];
  } else {
    const int cb = 2 * ox, last = W - 1;
    u32 e[17];  // columns cb - 1 .. cb + 15
#pragma unroll
    for (int i = 0; i < 17; ++i) {
      const int c = cb - 1 + i;
      const u32 t = row[c < 0 ? 0 : (c < last ? c : last)];
      e[i] = (c >= 0 && c < W) ? t : 0u;
    }
    d[0] = e[0] << 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[1 + i] = e[1 + 2 * i] | (e[2 + 2 * i] << 16);
  }
}
__device__ __forceinline__ u32x4 stem_patch_pick(const u32 (&d)[9], bool ok, int kx) {
  u32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // kx = 1: low halves of (d[1 + 2j], d[2 + 2j]); kx = 2: their high halves; kx = 0: high halves of (d[2j], d[1 + 2j])
    const u32 lo = kx == 0 ? d[2 * j] : d[1 + 2 * j], hi = kx == 0 ? d[1 + 2 * j] : d[2 + 2 * j];
    const u32 t = kx == 1 ? __builtin_amdgcn_perm(hi, lo, 0x05040100u) : __builtin_amdgcn_perm(hi, lo, 0x07060302u);
    out[j] = ok ? t : 0u;
  }
  return out;
}

template <int DT, bool FAST>
__device__ __forceinline__ u32x4 stem_dy_load(const u16* row, int ox, int Wo) {  // eight pixels ox .. ox + 7 of one channel row
  u32x4 out;
  if constexpr (FAST) {
    out = *reinterpret_cast<const u32x4*>(row + (ox < Wo ? ox : Wo - 8));
  } else {
    const int last = Wo - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = ox + 2 * i;
      const u32 e0 = row[c < last ? c : last], e1 = row[c + 1 < last ? c + 1 : last];
      out[i] = (c < Wo ? e0 : 0u) | ((c + 1 < Wo ? e1 : 0u) << 16);
    }
  }
  return out;
}

template <int DT, bool FAST>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemTrainParams p) {
  __shared__ float red[4][32 * 32];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  const int taps = p.Cin * 9;
  // this lane's two taps (B fragments 0 / 1) and two channels (A fragments 0 / 1)
  int t_ci[2], t_ky[2], t_kx[2];
  bool t_ok[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int tap = f * 16 + (int)fr;
    t_ok[f] = tap < taps;
    const int tt = t_ok[f] ? tap : 0;
    t_ci[f] = tt / 9;
    t_ky[f] = (tt % 9) / 3;
    t_kx[f] = tt % 3;
  }
  const bool c_ok[2] = {(int)fr < p.Cout, 16 + (int)fr < p.Cout};
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)p.Ho * p.Wo;
  const int total_rows = p.N * p.Ho;
  const int wid = (int)(blockIdx.x * 4u + wave);
  const int r0 = wid * p.rows_per_wave, r1 = r0 + p.rows_per_wave < total_rows ? r0 + p.rows_per_wave : total_rows;
  for (int r = r0; r < r1; ++r) {  // wave-uniform
    const int n = r / p.Ho, oy = r - n * p.Ho;
    const u16* xrow[2];
    bool xok[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int iy = 2 * oy + t_ky[f] - 1;
      xok[f] = t_ok[f] && (unsigned)iy < (unsigned)p.H;
      xrow[f] = p.x + (((size_t)n * p.Cin + t_ci[f]) * p.H + (xok[f] ? iy : 0)) * p.W;
    }
    const u16* grow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) grow[a] = p.dy + ((size_t)n * p.Cout + (c_ok[a] ? a * 16 + (int)fr : 0)) * plane + (size_t)oy * p.Wo;
    // a k-step = 32 output pixels: lane group fg holds pixels ox0 + 8 fg .. + 7; the NEXT k-step's operands are requested before
    // this one's are multiplied
    u32x4 A[2], An[2] = {};
    u32 D[2][9], Dn[2][9] = {};
#pragma unroll
    for (int a = 0; a < 2; ++a) A[a] = stem_dy_load<DT, FAST>(grow[a], 8 * (int)fg, p.Wo);
#pragma unroll
    for (int f = 0; f < 2; ++f) stem_patch_load<DT, FAST>(xrow[f], 8 * (int)fg, p.W, p.Wo, D[f]);
    for (int ox0 = 0; ox0 < p.Wo; ox0 += 32) {
      const int ox = ox0 + 8 * (int)fg;
      if (ox0 + 32 < p.Wo) {  // (uniform)
#pragma unroll
        for (int a = 0; a < 2; ++a) An[a] = stem_dy_load<DT, FAST>(grow[a], ox + 32, p.Wo);
#pragma unroll
        for (int f = 0; f < 2; ++f) stem_patch_load<DT, FAST>(xrow[f], ox + 32, p.W, p.Wo, Dn[f]);
      }
      const bool in = ox < p.Wo;
      u32x4 B[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) B[f] = stem_patch_pick(D[f], xok[f] && in, t_kx[f]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        u32x4 Aa = A[a];
#pragma unroll
        for (int i = 0; i < 4; ++i) Aa[i] = (c_ok[a] && in) ? Aa[i] : 0u;
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[a][f] = mfma16<DT>(Aa, B[f], acc[a][f]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) A[a] = An[a];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 9; ++i) D[f][i] = Dn[f][i];
    }
  }
  // D[m = 4 fg + j][n = fr] of tile (a, f) = dW[co = 16 a + 4 fg + j][tap = 16 f + fr]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][(a * 16 + (int)fg * 4 + j) * 32 + f * 16 + (int)fr] = acc[a][f][j];
  __syncthreads();
  for (u32 i = tid; i < 1024u; i += 256u) p.part[(size_t)blockIdx.x * 1024 + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

// dw[co][tap] = sum of the workgroup partials in index order: 64 row groups x 16 elements per workgroup, each group adds its
// partials g, g + 64, ... in order, the groups are added in order (64 workgroups: with 16 of them the pass took 23 us)
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const StemTrainParams p) {
  __shared__ float red[64][16];
  const u32 el = threadIdx.x & 15u, g = threadIdx.x >> 4;
  const u32 i = blockIdx.x * 16u + el;  // element of the 32 x 32 tile
  float s = 0.f;
  for (int q = (int)g; q < p.partials; q += 64) s += p.part[(size_t)q * 1024 + i];
  red[g][el] = s;
  __syncthreads();
  if (g == 0) {
    float t = red[0][el];
    for (int k = 1; k < 64; ++k) t += red[k][el];
    const int co = (int)(i >> 5), tap = (int)(i & 31u), taps = p.Cin * 9;
    if (co < p.Cout && tap < taps) p.dw[(size_t)co * taps + tap] = t;
  }
}

static int stem_check(const char* what, const void* x, int N, int Cin, int H, int W, int Cout, int dtype) {
  if (!x || N < 1 || Cin < 1 || Cin > 3 || H < 1 || W < 1 || Cout < 1 || Cout > 32 || (dtype != SSDK_BF16 && dtype != SSDK_F16)) {
    set_error("%s: bad arguments (N=%d Cin=%d H=%d W=%d Cout=%d dtype=%d; Cin <= 3, Cout <= 32, 16-bit tensors)", what, N, Cin, H, W, Cout,
              dtype);
    return SSDK_E_BADARG;
  }
  if ((size_t)N * Cout * (((size_t)H - 1) / 2 + 1) * (((size_t)W - 1) / 2 + 1) >= (1ull << 40)) {
    set_error("%s: tensor too large", what);
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}

static int stem_wgrad_plan(int N, int Ho, int* rows_per_wave) {
  const long rows = (long)N * Ho;
  long rpw = (rows + 4096 - 1) / 4096;  // ~1024 workgroups of four waves
  if (rpw < 1) rpw = 1;
  *rows_per_wave = (int)rpw;
  return (int)((rows + rpw * 4 - 1) / (rpw * 4));
}

}  // namespace ssdk

using namespace ssdk;

extern "C" size_t ssdk_stem3x3s2_wgrad_workspace_bytes(int N, int H) {
  if (N < 1 || H < 1) return 0;
  int rpw = 1;
  return (size_t)stem_wgrad_plan(N, (H - 1) / 2 + 1, &rpw) * 1024 * sizeof(float);  // the workgroup partials [partials][32 x 32] fp32
}

extern "C" int ssdk_stem3x3s2_fwd(const void* x, const float* w, void* y, int N, int Cin, int H, int W, int Cout, int dtype, void* stream) {
  if (int rc = stem_check("stem3x3s2_fwd", x, N, Cin, H, W, Cout, dtype)) return rc;
  if (!w || !y) {
    set_error("stem3x3s2_fwd: null pointer");
    return SSDK_E_BADARG;
  }
  StemTrainParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const u16*)x;
  p.w = w;
  p.y = (u16*)y;
  p.N = N;
  p.Cin = Cin;
  p.H = H;
  p.W = W;
  p.Cout = Cout;
  p.Ho = (H - 1) / 2 + 1;
  p.Wo = (W - 1) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if ((W & 1) == 0 && W >= 4 && (((uintptr_t)x) & 3u) == 0) {  // the matrix-core form
    const int parts = stem_wgrad_plan(N, p.Ho, &p.rows_per_wave);  // (the same split of the output rows over waves)
    if (dtype == SSDK_BF16) hipLaunchKernelGGL((stem_fwd_mfma_kernel<SSDK_BF16>), dim3((unsigned)parts), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_fwd_mfma_kernel<SSDK_F16>), dim3((unsigned)parts), dim3(256), 0, st, p);
    return check_launch("stem_fwd_mfma_kernel");
  }
  const int pairs = p.Ho * ((p.Wo + 1) / 2);
  const dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)N);
  const bool fast = (W & 3) == 0 && (((uintptr_t)x) & 7u) == 0;
#define SSDK_STEM_F(DT)                                                                              \
  do {                                                                                               \
    if (fast) {                                                                                      \
      if (Cin == 3) hipLaunchKernelGGL((stem_fwd_kernel<DT, 3, true>), grid, dim3(256), 0, st, p);   \
      else if (Cin == 2) hipLaunchKernelGGL((stem_fwd_kernel<DT, 2, true>), grid, dim3(256), 0, st, p); \
      else hipLaunchKernelGGL((stem_fwd_kernel<DT, 1, true>), grid, dim3(256), 0, st, p);            \
    } else {                                                                                         \
      if (Cin == 3) hipLaunchKernelGGL((stem_fwd_kernel<DT, 3, false>), grid, dim3(256), 0, st, p);  \
      else if (Cin == 2) hipLaunchKernelGGL((stem_fwd_kernel<DT, 2, false>), grid, dim3(256), 0, st, p); \
      else hipLaunchKernelGGL((stem_fwd_kernel<DT, 1, false>), grid, dim3(256), 0, st, p);           \
    }                                                                                                \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_STEM_F(SSDK_BF16);
  else SSDK_STEM_F(SSDK_F16);
#undef SSDK_STEM_F
  return check_launch("stem_fwd_kernel");
}

extern "C" int ssdk_stem3x3s2_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Cin,
                                    int H, int W, int Cout, int dtype, void* stream) {
  if (int rc = stem_check("stem3x3s2_wgrad", x, N, Cin, H, W, Cout, dtype)) return rc;
  if (!dy || !dw || !workspace || workspace_bytes < ssdk_stem3x3s2_wgrad_workspace_bytes(N, H) || ((uintptr_t)workspace & 15u)) {
    set_error("stem3x3s2_wgrad: null pointer, or workspace too small / misaligned");
    return SSDK_E_BADARG;
  }
  StemTrainParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const u16*)x;
  p.dy = (const u16*)dy;
  p.dw = dw;
  p.part = (float*)workspace;
  p.N = N;
  p.Cin = Cin;
  p.H = H;
  p.W = W;
  p.Cout = Cout;
  p.Ho = (H - 1) / 2 + 1;
  p.Wo = (W - 1) / 2 + 1;
  p.partials = stem_wgrad_plan(N, p.Ho, &p.rows_per_wave);
  hipStream_t st = (hipStream_t)stream;
  const bool fast = (W & 15) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15u) == 0;  // (then Wo = W / 2 is a multiple of 8)
  const dim3 grid((unsigned)p.partials);
  if (dtype == SSDK_BF16) {
    if (fast) hipLaunchKernelGGL((stem_wgrad_kernel<SSDK_BF16, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_wgrad_kernel<SSDK_BF16, false>), grid, dim3(256), 0, st, p);
  } else {
    if (fast) hipLaunchKernelGGL((stem_wgrad_kernel<SSDK_F16, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_wgrad_kernel<SSDK_F16, false>), grid, dim3(256), 0, st, p);
  }
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64), dim3(1024), 0, st, p);
  return check_launch("stem_wgrad_kernel");
}
