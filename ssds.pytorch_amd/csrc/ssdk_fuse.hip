// ssdk_fuse.hip -- weighted feature fusion of the BiFPN on gfx950 (HBM-bound, NHWC, 16-byte vectors).
//
// Replaces the element-wise chains of the reference's BiFPNModule.forward (bifpn.py:41-62):
//     top-down   w0 * x_i      + w1 * upsample2x_nearest(x_{i+1})
//     bottom-up  w0 * x_{i+1}  + w1 * max_pool2d(x_i, 2)  (+ w2 * skip_{i+1})
// which the reference runs as 4-6 separate ATen launches (interpolate / max_pool2d / mul / add) with a full-size
// temporary each.  One launch here: every lane owns 8 channels of one output pixel, reads its 1-3 sources (one
// pixel, the parent pixel, or the 2x2 window), accumulates in fp32 and rounds once.  The fusion weights are the
// fast-normalised relu(w) / (sum relu(w) + 1e-6) scalars, evaluated on the host when the plan is recorded.
// Algorithmic bytes: every source once + the output once.
#include "ssdk_conv_common.h"

namespace ssdk {

struct FuseParams {
  const u16* a;
  const u16* b;
  const u16* c;
  u16* y;
  float w0, w1, w2;
  int mode_b, mode_c;  // SSDK_FUSE_SAME | SSDK_FUSE_UP2 | SSDK_FUSE_POOL2
  int hb, wb, hc, wc;  // source dims of b / c (POOL2: 2H or 2H+1 rows -- floor mode drops the odd row)
  int N, H, W, C;      // output dims
  long total;          // N*H*W*C/8
};

template <int DT>
__device__ __forceinline__ void acc8(float (&acc)[8], const u32x4 v, float w) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[2 * e] += w * bits16_to_f32<DT>(v[e] & 0xffffu);
    acc[2 * e + 1] += w * bits16_to_f32<DT>(v[e] >> 16);
  }
}

template <int DT>
__device__ __forceinline__ void add_source(float (&acc)[8], const u16* src, int mode, float w, int n, int y, int x,
                                           int c0, int H, int W, int C, int Hs, int Ws) {
  if (mode == SSDK_FUSE_SAME) {
    acc8<DT>(acc, *reinterpret_cast<const u32x4*>(src + (((size_t)n * H + y) * W + x) * C + c0), w);
  } else if (mode == SSDK_FUSE_UP2) {  // source is [N][H/2][W/2][C]
    acc8<DT>(acc, *reinterpret_cast<const u32x4*>(src + (((size_t)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + c0), w);
  } else {  // max_pool2d(kernel 2, stride 2) of a [N][2H(+1)][2W(+1)][C] source (floor mode: the odd row/column is dropped)
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (((size_t)n * Hs + 2 * y + dy) * Ws + 2 * x + dx) * C + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bits16_to_f32<DT>(v[e] & 0xffffu), hi = bits16_to_f32<DT>(v[e] >> 16);
          m[2 * e] = (lo > m[2 * e] || lo != lo) ? lo : m[2 * e];  // NaN propagates like torch's max_pool2d
          m[2 * e + 1] = (hi > m[2 * e + 1] || hi != hi) ? hi : m[2 * e + 1];
        }
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += w * m[e];
  }
}

template <int DT>
__global__ __launch_bounds__(256) void fuse_kernel(const FuseParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const int cg = p.C / 8;
  const int c0 = (int)(t % cg) * 8;
  long r = t / cg;
  const int x = (int)(r % p.W);
  r /= p.W;
  const int y = (int)(r % p.H);
  const int n = (int)(r / p.H);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  add_source<DT>(acc, p.a, SSDK_FUSE_SAME, p.w0, n, y, x, c0, p.H, p.W, p.C, p.H, p.W);
  add_source<DT>(acc, p.b, p.mode_b, p.w1, n, y, x, c0, p.H, p.W, p.C, p.hb, p.wb);
  if (p.c) add_source<DT>(acc, p.c, p.mode_c, p.w2, n, y, x, c0, p.H, p.W, p.C, p.hc, p.wc);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack2_16<DT>(acc[2 * e], acc[2 * e + 1]);
  *reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.H + y) * p.W + x) * p.C + c0) = o;
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_fuse(const ssdk_fuse_desc* d, void* stream) {
  if (!d || !d->a || !d->b || !d->y) {
    set_error("fuse: null pointer (a, b and y are mandatory)");
    return SSDK_E_BADARG;
  }
  if (d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) {
    set_error("fuse: dtype must be bf16 or f16");
    return SSDK_E_BADARG;
  }
  if (d->N < 1 || d->H < 1 || d->W < 1 || d->C < 8 || (d->C % 8) || d->mode_b < 0 || d->mode_b > 2 ||
      (d->c && (d->mode_c < 0 || d->mode_c > 2))) {
    set_error("fuse: bad geometry N=%d H=%d W=%d C=%d (C %% 8 == 0) modes %d/%d", d->N, d->H, d->W, d->C, d->mode_b,
              d->mode_c);
    return SSDK_E_BADARG;
  }
  if (((d->mode_b == SSDK_FUSE_UP2) || (d->c && d->mode_c == SSDK_FUSE_UP2)) && ((d->H | d->W) & 1)) {
    set_error("fuse: an upsampled source needs even output dims (%dx%d)", d->H, d->W);
    return SSDK_E_BADARG;
  }
  if (((uintptr_t)d->a | (uintptr_t)d->b | (uintptr_t)d->c | (uintptr_t)d->y) & 15) {
    set_error("fuse: tensors must be 16-byte aligned");
    return SSDK_E_BADARG;
  }
  FuseParams p;
  p.a = (const u16*)d->a;
  p.b = (const u16*)d->b;
  p.c = (const u16*)d->c;
  p.y = (u16*)d->y;
  p.w0 = d->w0;
  p.w1 = d->w1;
  p.w2 = d->w2;
  p.mode_b = d->mode_b;
  p.mode_c = d->mode_c;
  auto src_dims = [&](int mode, int* hs, int* ws, int given_h, int given_w) {
    if (mode == SSDK_FUSE_POOL2) {  // floor(hs / 2) == H
      *hs = given_h;
      *ws = given_w;
      return given_h / 2 == d->H && given_w / 2 == d->W;
    }
    *hs = mode == SSDK_FUSE_UP2 ? d->H / 2 : d->H;
    *ws = mode == SSDK_FUSE_UP2 ? d->W / 2 : d->W;
    return true;
  };
  if (!src_dims(d->mode_b, &p.hb, &p.wb, d->hb, d->wb) || (d->c && !src_dims(d->mode_c, &p.hc, &p.wc, d->hc, d->wc))) {
    set_error("fuse: pooled source dims (%dx%d / %dx%d) do not reduce to the output %dx%d", d->hb, d->wb, d->hc, d->wc,
              d->H, d->W);
    return SSDK_E_BADARG;
  }
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.C = d->C;
  p.total = (long)d->N * d->H * d->W * (d->C / 8);
  const unsigned grid = (unsigned)((p.total + 255) / 256);
  if (d->dtype == SSDK_BF16) hipLaunchKernelGGL((fuse_kernel<SSDK_BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((fuse_kernel<SSDK_F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("fuse_kernel");
}
