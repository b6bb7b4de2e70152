// ssdk_dwtrain.hip -- depthwise 3x3 convolution for the TRAINING step (forward, input gradient, weight gradient),
// NCHW, fp32 | bf16 | f16, fp32 accumulation, on gfx950.
//
// Why: the DDP training step of SSD-MobileNetV2 (reference pipeline_anchor_apex.py:75-171 on torchvision's
// InvertedResidual blocks) spends more than half of its GPU time in MIOpen's `naive_conv_*_wrw/bwd/fwd` kernels,
// which is what PyTorch-ROCm dispatches bf16 depthwise convolutions to (profiles/README.md).  All three passes are
// HBM-bound window operations on independent (image, channel) planes:
//   forward        y[n,c,oy,ox] = sum_t w[c,t] * x[n,c, oy*s + ky - 1, ox*s + kx - 1]
//   input grad     dx[n,c,iy,ix] = sum_t w[c,t] * dy[n,c,(iy + 1 - ky)/s, (ix + 1 - kx)/s]   (exact divisions only)
//   weight grad    dw[c,t]      = sum_{n,oy,ox} x[n,c, oy*s + ky - 1, ox*s + kx - 1] * dy[n,c,oy,ox]
// A workgroup is a 4-row x 64-column tile of one plane (lanes consecutive in x: coalesced, the 3x3 window re-reads
// hit L1/L2).  The weight gradient is a two-stage reduction in a FIXED order (per-plane-tile partial sums, then one
// thread per (c, tap) adds them up) -- no float atomics, bit-reproducible run to run.
#include "ssdk_conv_common.h"

namespace ssdk {

struct DwtParams {
  const void* a;   // x (fwd, wgrad) | dy (dgrad)
  const void* b;   // w (fwd, dgrad: [C][9] of the activation dtype) | dy (wgrad)
  void* out;       // y | dx | partial sums
  int N, C, H, W, Ho, Wo, stride, dtype;
  int tiles_x, tiles_y;  // tiles of the OUTPUT of the pass (y / dx / y-space for wgrad)
};

template <int DT> __device__ __forceinline__ float ldf(const void* p, size_t i) {
  if constexpr (DT == SSDK_F32) return ((const float*)p)[i];
  else if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(((const u16*)p)[i]);
  else return f16_bits_to_f32(((const u16*)p)[i]);
}
template <int DT> __device__ __forceinline__ void stf(void* p, size_t i, float v) {
  if constexpr (DT == SSDK_F32) ((float*)p)[i] = v;
  else ((u16*)p)[i] = (u16)f32_to_bits16<DT>(v);
}

template <int DT, int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwtParams p) {
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;  // n*C + c
  const int c = plane % p.C;
  const int ox = tx * 64 + (threadIdx.x & 63), oy = ty * 4 + (threadIdx.x >> 6);
  if (ox >= p.Wo || oy >= p.Ho) return;
  const size_t xb = (size_t)plane * p.H * p.W;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * S + ky - 1;
    if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * S + kx - 1;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      acc += ldf<DT>(p.b, (size_t)c * 9 + ky * 3 + kx) * ldf<DT>(p.a, xb + (size_t)iy * p.W + ix);
    }
  }
  stf<DT>(p.out, (size_t)plane * p.Ho * p.Wo + (size_t)oy * p.Wo + ox, acc);
}

template <int DT, int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwtParams p) {  // output = dx [N,C,H,W]
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;
  const int c = plane % p.C;
  const int ix = tx * 64 + (threadIdx.x & 63), iy = ty * 4 + (threadIdx.x >> 6);
  if (ix >= p.W || iy >= p.H) return;
  const size_t yb = (size_t)plane * p.Ho * p.Wo;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ny = iy + 1 - ky;
    if (ny < 0 || (S == 2 && (ny & 1))) continue;
    const int oy = ny / S;
    if (oy >= p.Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int nx = ix + 1 - kx;
      if (nx < 0 || (S == 2 && (nx & 1))) continue;
      const int ox = nx / S;
      if (ox >= p.Wo) continue;
      acc += ldf<DT>(p.b, (size_t)c * 9 + ky * 3 + kx) * ldf<DT>(p.a, yb + (size_t)oy * p.Wo + ox);
    }
  }
  stf<DT>(p.out, (size_t)plane * p.H * p.W + (size_t)iy * p.W + ix, acc);
}

// stage 1: partial[(plane * tiles + tile) * 9 + t] = sum over the tile's output pixels
template <int DT, int S>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwtParams p) {
  __shared__ float red[4][9];
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;
  const int ox = tx * 64 + (threadIdx.x & 63), oy = ty * 4 + (threadIdx.x >> 6);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (ox < p.Wo && oy < p.Ho) {
    const float g = ldf<DT>(p.b, (size_t)plane * p.Ho * p.Wo + (size_t)oy * p.Wo + ox);
    const size_t xb = (size_t)plane * p.H * p.W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * S + ky - 1;
      if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * S + kx - 1;
        if ((unsigned)ix >= (unsigned)p.W) continue;
        acc[ky * 3 + kx] = g * ldf<DT>(p.a, xb + (size_t)iy * p.W + ix);
      }
    }
  }
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = acc[t];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);  // fixed butterfly: same order every run
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    ((float*)p.out)[((size_t)plane * gridDim.x + blockIdx.x) * 9 + threadIdx.x] = v;
  }
}

// stage 2: dw[c][t] = sum over images and tiles, in index order
__global__ __launch_bounds__(64) void dw_wgrad_reduce_kernel(const float* partial, float* dw, int N, int C, int tiles) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= C * 9) return;
  const int c = i / 9, t = i % 9;
  float s = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* q = partial + ((size_t)(n * C + c) * tiles) * 9 + t;
    for (int k = 0; k < tiles; ++k) s += q[(size_t)k * 9];
  }
  dw[i] = s;
}

static int dwt_check(const char* what, const void* a, const void* b, const void* o, int N, int C, int H, int W, int stride,
                     int dtype) {
  if (!a || !b || !o) {
    set_error("%s: null pointer", what);
    return SSDK_E_BADARG;
  }
  if (N < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2) ||
      (dtype != SSDK_F32 && dtype != SSDK_BF16 && dtype != SSDK_F16) || (long)N * C > 2147483647l) {
    set_error("%s: bad arguments N=%d C=%d H=%d W=%d stride=%d dtype=%d", what, N, C, H, W, stride, dtype);
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}

static DwtParams dwt_params(const void* a, const void* b, void* o, int N, int C, int H, int W, int stride, int dtype,
                            bool out_is_input_space) {
  DwtParams p;
  p.a = a;
  p.b = b;
  p.out = o;
  p.N = N;
  p.C = C;
  p.H = H;
  p.W = W;
  p.stride = stride;
  p.dtype = dtype;
  p.Ho = (H + 2 - 3) / stride + 1;
  p.Wo = (W + 2 - 3) / stride + 1;
  const int oh = out_is_input_space ? H : p.Ho, ow = out_is_input_space ? W : p.Wo;
  p.tiles_x = (ow + 63) / 64;
  p.tiles_y = (oh + 3) / 4;
  return p;
}

#define SSDK_DWT_LAUNCH(KERNEL, grid)                                                                     \
  do {                                                                                                    \
    if (dtype == SSDK_F32) {                                                                              \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_F32, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);   \
      else hipLaunchKernelGGL((KERNEL<SSDK_F32, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);               \
    } else if (dtype == SSDK_BF16) {                                                                      \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_BF16, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);  \
      else hipLaunchKernelGGL((KERNEL<SSDK_BF16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);              \
    } else {                                                                                              \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_F16, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);   \
      else hipLaunchKernelGGL((KERNEL<SSDK_F16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);               \
    }                                                                                                     \
  } while (0)

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_dwconv_fwd(const void* x, const void* w, void* y, int N, int C, int H, int W, int stride, int dtype,
                               void* stream) {
  const int rc = dwt_check("dwconv_fwd", x, w, y, N, C, H, W, stride, dtype);
  if (rc) return rc;
  DwtParams p = dwt_params(x, w, y, N, C, H, W, stride, dtype, false);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_fwd_kernel, grid);
  return check_launch("dw_fwd_kernel");
}

extern "C" int ssdk_dwconv_bwd_data(const void* dy, const void* w, void* dx, int N, int C, int H, int W, int stride,
                                    int dtype, void* stream) {
  const int rc = dwt_check("dwconv_bwd_data", dy, w, dx, N, C, H, W, stride, dtype);
  if (rc) return rc;
  DwtParams p = dwt_params(dy, w, dx, N, C, H, W, stride, dtype, true);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_dgrad_kernel, grid);
  return check_launch("dw_dgrad_kernel");
}

extern "C" size_t ssdk_dwconv_bwd_weight_workspace_bytes(int N, int C, int H, int W, int stride) {
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  return (size_t)N * C * ((Wo + 63) / 64) * ((Ho + 3) / 4) * 9 * sizeof(float);
}

extern "C" int ssdk_dwconv_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                      int N, int C, int H, int W, int stride, int dtype, void* stream) {
  const int rc = dwt_check("dwconv_bwd_weight", x, dy, dw, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (!workspace || workspace_bytes < ssdk_dwconv_bwd_weight_workspace_bytes(N, C, H, W, stride)) {
    set_error("dwconv_bwd_weight: workspace too small");
    return SSDK_E_BADARG;
  }
  DwtParams p = dwt_params(x, dy, workspace, N, C, H, W, stride, dtype, false);
  const int tiles = p.tiles_x * p.tiles_y;
  const dim3 grid((unsigned)tiles, (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_wgrad_kernel, grid);
  int rc2 = check_launch("dw_wgrad_kernel");
  if (rc2) return rc2;
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((unsigned)((C * 9 + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     (const float*)workspace, dw, N, C, tiles);
  return check_launch("dw_wgrad_reduce_kernel");
}
