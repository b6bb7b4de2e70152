// ssdk_stem.hip -- the ResNet stem on gfx950: 7x7 / stride 2 convolution on the 3-channel image (+ folded BN +
// ReLU) on the matrix cores, and the 3x3 / stride 2 max pooling that follows it.
//
// Reference: ssds/modeling/nets/resnet.py:41-46 (torchvision ResNet: conv1 7x7/2 pad 3 -> bn1 -> relu -> maxpool
// 3x3/2 pad 1).  With 3 input channels the conv is a K = 147 contraction per pixel -- too thin for an implicit GEMM
// over NHWC rows, so the kernel builds the im2col operand on the fly from an LDS patch of the image:
//   * a workgroup (4 waves) owns 8 rows x 16 columns of output pixels; the (2*8+5) x (2*16+5) input patch is staged
//     in LDS as [row][col][4] (channel padded 3 -> 4, zeros outside the image);
//   * K is laid out as (ky, kx padded 7 -> 8, ci padded 3 -> 4) = 7 k-steps of 32: the 8 k-values of a lane are 2
//     neighbouring patch pixels = ONE aligned ds_read_b128, no gather;
//   * the weights (64 x 224, zero in the padding slots) live in VGPRs in fragment layout for the whole persistent
//     workgroup (28 x 16 bytes per lane), the image patch is the only LDS traffic;
//   * D = W * X^T (rows = channels, columns = pixels): a lane ends up with 4 consecutive channels of one pixel,
//     stored as 8 bytes NHWC.
// The pooling kernel is a plain HBM-bound NHWC window max (8 channels per lane, -inf padding like torch).
#include "ssdk_conv_common.h"

namespace ssdk {

struct StemParams {
  const u16* x;
  const u16* w;  // [Cout][7][8][4]
  const float* scale;
  const float* bias;
  u16* y;
  int N, H, W, Cout, Ho, Wo, act, in_layout;
  int tiles_x, tiles_y;
  u32 ntiles;
};

constexpr int ST_TH = 8, ST_TW = 16;
constexpr int ST_PR = 2 * ST_TH + 5, ST_PC = 2 * ST_TW + 5 + 1;  // 21 x 38 (one zero pad column: even row stride)
constexpr int ST_PATCH = ST_PR * ST_PC * 4 + 8;                  // u16 elements (+ tail pad read by the kx = 7 slot)

template <int DT, int NJ>  // NJ = Cout / 16
__global__ __launch_bounds__(256) void stem7_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) u16 patch[ST_PATCH];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;

  // weights -> registers, fragment layout: wf[ky][j] = W[j*16 + fr][ky][2*fg .. 2*fg+1][0..3]
  u32x4 wf[7][NJ];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      wf[ky][j] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)(j * 16 + (int)fr) * 7 + ky) * 32 + fg * 8);
  f32x4 sc[NJ], bi[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sc[j] = *reinterpret_cast<const f32x4*>(p.scale + j * 16 + fg * 4);
    bi[j] = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + fg * 4);
  }
  for (u32 i = tid; i < (u32)ST_PATCH; i += 256) patch[i] = 0;  // channel 3, pad column and tail stay zero forever
  const ActSel as = act_sel(p.act);
  const bool any_sig = act_is_sig(p.act), any_clamp = act_is_clamp(p.act);
  const size_t cstride = p.in_layout == LAYOUT_NCHW ? (size_t)p.H * p.W : 1;
  const size_t pstride = p.in_layout == LAYOUT_NCHW ? 1 : 3;

  // The image patch of the NEXT tile is requested (into registers) before the current tile is computed: with ~200 VGPRs
  // only two workgroups share a CU, too few to hide a tile's staging round trip (372 us for the 640x640 stem at batch 32 when
  // every tile waited for its own loads).
  constexpr u32 NEL = (u32)(ST_PR * (ST_PC - 1) * 3), NPT = (NEL + 255u) / 256u;
  u16 pv[NPT];
  auto tile_origin = [&](u32 t, u32* n_, int* oy0_, int* ox0_) {
    const u32 tx = t % (u32)p.tiles_x;
    const u32 q = t / (u32)p.tiles_x;
    *n_ = q / (u32)p.tiles_y;
    *oy0_ = (int)(q % (u32)p.tiles_y) * ST_TH;
    *ox0_ = (int)tx * ST_TW;
  };
  auto fetch = [&](u32 t) {  // (channel, row, column): columns fastest; all of a thread's loads issued together
    u32 n;
    int oy0, ox0;
    tile_origin(t < p.ntiles ? t : p.ntiles - 1u, &n, &oy0, &ox0);  // (past the end: a harmless re-read)
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const u16* img = p.x + (size_t)n * 3 * p.H * p.W;
#pragma unroll
    for (u32 k = 0; k < NPT; ++k) {
      const u32 i = tid + k * 256u;
      const u32 c = i % (u32)(ST_PC - 1);
      const u32 r2 = i / (u32)(ST_PC - 1);
      const u32 r = r2 % (u32)ST_PR, ch = r2 / (u32)ST_PR;
      const int iy = iy0 + (int)r, ix = ix0 + (int)c;
      pv[k] = 0;
      if (i < NEL && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) pv[k] = img[((size_t)iy * p.W + ix) * pstride + ch * cstride];
    }
  };
  if (blockIdx.x < p.ntiles) fetch(blockIdx.x);
  for (u32 t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    u32 n;
    int oy0, ox0;
    tile_origin(t, &n, &oy0, &ox0);
    __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
    for (u32 k = 0; k < NPT; ++k) {
      const u32 i = tid + k * 256u;
      const u32 c = i % (u32)(ST_PC - 1);
      const u32 r2 = i / (u32)(ST_PC - 1);
      const u32 r = r2 % (u32)ST_PR, ch = r2 / (u32)ST_PR;
      if (i < NEL) patch[(r * ST_PC + c) * 4 + ch] = pv[k];
    }
    __syncthreads();
    fetch(t + gridDim.x);  // in flight under this tile's MFMAs and stores
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int f = (int)wave * 2 + mi;  // output row of the tile = m-fragment
      f32x4 e[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) e[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const u32x4 xf = *reinterpret_cast<const u32x4*>(&patch[((f * 2 + ky) * ST_PC + (int)fr * 2 + 2 * (int)fg) * 4]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) e[j] = mfma16<DT>(wf[ky][j], xf, e[j]);  // D[channel fg*4+r][pixel fr]
      }
      const int oy = oy0 + f, ox = ox0 + (int)fr;
      if (oy < p.Ho && ox < p.Wo) {
        u16* dst = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + fg * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = e[j][r] * sc[j][r] + bi[j][r];
          if (any_sig) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * v[r]));
              v[r] = as.mode == 1 ? sg : v[r] * sg;
            }
          }
          if (any_clamp) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_fminf(__builtin_fmaxf(v[r], as.lo), as.hi);
          }
          *reinterpret_cast<uint2*>(dst + j * 16) = make_uint2(pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]));
        }
      }
    }
  }
}

struct PoolParams {
  const u16* x;
  u16* y;
  int N, H, W, C, Ho, Wo;
  long total;
};

template <int DT>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const PoolParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const int cg = p.C / 8;
  const int c0 = (int)(t % cg) * 8;
  long r = t / cg;
  const int ox = (int)(r % p.Wo);
  r /= p.Wo;
  const int oy = (int)(r % p.Ho);
  const int n = (int)(r / p.Ho);
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.H + iy) * p.W + ix) * p.C + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bits16_to_f32<DT>(v[e] & 0xffffu), hi = bits16_to_f32<DT>(v[e] >> 16);
        m[2 * e] = (lo > m[2 * e] || lo != lo) ? lo : m[2 * e];
        m[2 * e + 1] = (hi > m[2 * e + 1] || hi != hi) ? hi : m[2 * e + 1];
      }
    }
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack2_16<DT>(m[2 * e], m[2 * e + 1]);
  *reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.C + c0) = o;
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_conv_stem7(const ssdk_stem_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->scale || !d->bias || !d->y) {
    set_error("conv_stem7: null pointer");
    return SSDK_E_BADARG;
  }
  if ((d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) || d->Cin != 3 || (d->Cout != 32 && d->Cout != 64) || d->N < 1 ||
      d->H < 1 || d->W < 1) {
    set_error("conv_stem7: needs bf16|f16, Cin = 3, Cout in {32, 64} (got Cin=%d Cout=%d)", d->Cin, d->Cout);
    return SSDK_E_BADARG;
  }
  if (((uintptr_t)d->w | (uintptr_t)d->y | (uintptr_t)d->scale | (uintptr_t)d->bias) & 15) {
    set_error("conv_stem7: w, scale, bias and y must be 16-byte aligned");
    return SSDK_E_BADARG;
  }
  StemParams p;
  p.x = (const u16*)d->x;
  p.w = (const u16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.y = (u16*)d->y;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.Cout = d->Cout;
  p.Ho = (d->H + 6 - 7) / 2 + 1;
  p.Wo = (d->W + 6 - 7) / 2 + 1;
  p.act = d->act;
  p.in_layout = d->in_layout;
  p.tiles_x = (p.Wo + ST_TW - 1) / ST_TW;
  p.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
  const long nt = (long)d->N * p.tiles_x * p.tiles_y;
  if (nt >= (1l << 31)) {
    set_error("conv_stem7: too many tiles");
    return SSDK_E_BADARG;
  }
  p.ntiles = (u32)nt;
  const unsigned grid = (unsigned)(nt < 2048 ? nt : 2048);
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == SSDK_BF16) {
    if (d->Cout == 64) hipLaunchKernelGGL((stem7_kernel<SSDK_BF16, 4>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((stem7_kernel<SSDK_BF16, 2>), dim3(grid), dim3(256), 0, s, p);
  } else {
    if (d->Cout == 64) hipLaunchKernelGGL((stem7_kernel<SSDK_F16, 4>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((stem7_kernel<SSDK_F16, 2>), dim3(grid), dim3(256), 0, s, p);
  }
  return check_launch("stem7_kernel");
}

extern "C" int ssdk_maxpool3x3s2(const ssdk_pool_desc* d, void* stream) {
  if (!d || !d->x || !d->y) {
    set_error("maxpool3x3s2: null pointer");
    return SSDK_E_BADARG;
  }
  if ((d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) || d->N < 1 || d->H < 1 || d->W < 1 || d->C < 8 || (d->C % 8) ||
      (((uintptr_t)d->x | (uintptr_t)d->y) & 15)) {
    set_error("maxpool3x3s2: needs bf16|f16, NHWC, C %% 8 == 0, 16-byte aligned tensors");
    return SSDK_E_BADARG;
  }
  PoolParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.C = d->C;
  p.Ho = (d->H + 2 - 3) / 2 + 1;
  p.Wo = (d->W + 2 - 3) / 2 + 1;
  p.total = (long)d->N * p.Ho * p.Wo * (d->C / 8);
  const unsigned grid = (unsigned)((p.total + 255) / 256);
  if (d->dtype == SSDK_BF16) hipLaunchKernelGGL((maxpool3x3s2_kernel<SSDK_BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((maxpool3x3s2_kernel<SSDK_F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("maxpool3x3s2_kernel");
}
