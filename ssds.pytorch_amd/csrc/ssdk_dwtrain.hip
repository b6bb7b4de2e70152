// ssdk_dwtrain.hip -- depthwise 3x3 convolution for the TRAINING step (forward, input gradient, weight gradient),
// NCHW, fp32 | bf16 | f16, fp32 accumulation, on gfx950: the C-ABI entry points, and the TILED kernels (round 1).
// Since round 2 the entry points try the whole-row kernels of ssdk_dwplane.hip first (3-5x faster on the plane sizes
// the backbones have); the tiled kernels below take rows too wide for that kernel's LDS budget and SSDK_DW_PLANE=0.
//
// Why: the DDP training step of SSD-MobileNetV2 (reference pipeline_anchor_apex.py:75-171 on torchvision's
// InvertedResidual blocks) spends more than half of its GPU time in MIOpen's `naive_conv_*_wrw/bwd/fwd` kernels,
// which is what PyTorch-ROCm dispatches bf16 depthwise convolutions to (profiles/README.md).  All three passes are
// HBM-bound window operations on independent (image, channel) planes:
//   forward        y[n,c,oy,ox] = sum_t w[c,t] * x[n,c, oy*s + ky - 1, ox*s + kx - 1]
//   input grad     dx[n,c,iy,ix] = sum_t w[c,t] * dy[n,c,(iy + 1 - ky)/s, (ix + 1 - kx)/s]   (exact divisions only)
//   weight grad    dw[c,t]      = sum_{n,oy,ox} x[n,c, oy*s + ky - 1, ox*s + kx - 1] * dy[n,c,oy,ox]
// A workgroup owns a 32-row x 64-column output tile of one plane: the input tile (+halo) is staged once in LDS as
// fp32 by aligned 16-byte reads, a lane computes 8 consecutive pixels of one row and stores them as 16 bytes.  The weight gradient is a two-stage reduction in a FIXED order (per-plane-tile partial sums, then one
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

constexpr int DT_TH = 32, DT_TW = 64;  // output tile of a workgroup: 256 threads = 32 rows x 8 groups of 8 pixels

// 8 (2-byte types) or 4 (fp32) consecutive elements <-> fp32
template <int DT> struct Vec16 { static constexpr int n = DT == SSDK_F32 ? 4 : 8; };
template <int DT>
__device__ __forceinline__ void load16(const void* src, size_t i, float (&v)[8]) {  // i: element index, 16-byte aligned
  if constexpr (DT == SSDK_F32) {
    const f32x4 q = *reinterpret_cast<const f32x4*>((const float*)src + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = q[e];
  } else {
    const u32x4 q = *reinterpret_cast<const u32x4*>((const u16*)src + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bits16_to_f32<DT>(q[e] & 0xffffu);
      v[2 * e + 1] = bits16_to_f32<DT>(q[e] >> 16);
    }
  }
}

// stage rows [r0, r0+NR) x cols [c0, c0+NC) of one plane into LDS as fp32, zeros outside the plane.  When the rows
// of the plane are 16-byte aligned (Ws a multiple of the vector width) a lane fetches one aligned 16-byte chunk and
// scatters its in-range elements; otherwise element by element.  Either way lanes run along the columns.
template <int DT>
__device__ __forceinline__ void stage_tile(float* lds, int ld, const void* src, size_t plane_base, int Hs, int Ws, int r0,
                                           int c0, int NR, int NC) {
  constexpr int VN = Vec16<DT>::n;
  const bool vec = (Ws % VN) == 0 && (plane_base % VN) == 0 && (((uintptr_t)src) & 15u) == 0;
  if (vec) {
    const int cbeg = c0 >= 0 ? (c0 / VN) * VN : -(((-c0) + VN - 1) / VN) * VN;  // floor to a chunk boundary
    const int nch = (c0 + NC - cbeg + VN - 1) / VN;
    for (int i = threadIdx.x; i < NR * nch; i += 256) {
      const int r = i / nch, ch = i - r * nch;
      const int y = r0 + r, x = cbeg + ch * VN;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if ((unsigned)y < (unsigned)Hs && x >= 0 && x + VN <= Ws) load16<DT>(src, plane_base + (size_t)y * Ws + x, v);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const int c = x + e - c0;
        if (c >= 0 && c < NC) lds[r * ld + c] = v[e];
      }
    }
  } else {
    for (int i = threadIdx.x; i < NR * NC; i += 256) {
      const int r = i / NC, c = i - r * NC;
      const int y = r0 + r, x = c0 + c;
      float v = 0.f;
      if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) v = ldf<DT>(src, plane_base + (size_t)y * Ws + x);
      lds[r * ld + c] = v;
    }
  }
}

// 8 consecutive outputs of one row: 16-byte store(s) when aligned and complete, element stores otherwise
template <int DT>
__device__ __forceinline__ void store8(void* dst, size_t i, const float (&v)[8], int valid) {
  if constexpr (DT == SSDK_F32) {
    float* d = (float*)dst + i;
    if (valid == 8 && (((uintptr_t)d) & 15u) == 0) {
      *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      for (int e = 0; e < valid; ++e) d[e] = v[e];
    }
  } else {
    u16* d = (u16*)dst + i;
    if (valid == 8 && (((uintptr_t)d) & 15u) == 0) {
      *reinterpret_cast<u32x4*>(d) = u32x4{pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]), pack2_16<DT>(v[4], v[5]),
                                           pack2_16<DT>(v[6], v[7])};
    } else {
      for (int e = 0; e < valid; ++e) d[e] = (u16)f32_to_bits16<DT>(v[e]);
    }
  }
}

template <int DT, int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwtParams p) {
  constexpr int NR = (DT_TH - 1) * S + 3, NC = (DT_TW - 1) * S + 3, LD = NC | 1;
  extern __shared__ float xt[];
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;  // n*C + c
  const int c = plane % p.C;
  const int oy0 = ty * DT_TH, ox0 = tx * DT_TW;
  stage_tile<DT>(xt, LD, p.a, (size_t)plane * p.H * p.W, p.H, p.W, oy0 * S - 1, ox0 * S - 1, NR, NC);
  float w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = ldf<DT>(p.b, (size_t)c * 9 + t);
  __syncthreads();
  const int row = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int oy = oy0 + row, ox = ox0 + g * 8;
  if (oy >= p.Ho || ox >= p.Wo) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const float* xr = xt + (row * S + ky) * LD + g * 8 * S;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[e] += w[ky * 3 + kx] * xr[e * S + kx];
  }
  store8<DT>(p.out, (size_t)plane * p.Ho * p.Wo + (size_t)oy * p.Wo + ox, acc, p.Wo - ox >= 8 ? 8 : p.Wo - ox);
}

template <int DT, int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwtParams p) {  // output tile = 32 x 64 of dx [N,C,H,W]
  // dy rows / cols that can reach the tile: (iy + 1 - ky) / S for iy in [iy0, iy0 + 32), ky in 0..2
  constexpr int NR = (DT_TH + 1) / S + 2, NC = (DT_TW + 1) / S + 2, LD = NC | 1;
  extern __shared__ float gt[];
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;
  const int c = plane % p.C;
  const int iy0 = ty * DT_TH, ix0 = tx * DT_TW;
  // first dy row / col staged: floor((iy0 - 1) / S); iy0, ix0 are even
  const int gy0 = S == 1 ? iy0 - 1 : iy0 / 2 - 1, gx0 = S == 1 ? ix0 - 1 : ix0 / 2 - 1;
  stage_tile<DT>(gt, LD, p.a, (size_t)plane * p.Ho * p.Wo, p.Ho, p.Wo, gy0, gx0, NR, NC);
  float w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = ldf<DT>(p.b, (size_t)c * 9 + t);
  __syncthreads();
  const int row = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int iy = iy0 + row, ixb = ix0 + g * 8;
  if (iy >= p.H || ixb >= p.W) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ny = iy + 1 - ky;
    if (S == 2 && (ny & 1)) continue;
    const float* gr = gt + ((S == 1 ? ny : ny >> 1) - gy0) * LD;  // staged rows outside the plane hold zeros
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int nx = ixb + e + 1 - kx;
        if (S == 2 && (nx & 1)) continue;
        acc[e] += w[ky * 3 + kx] * gr[(S == 1 ? nx : nx >> 1) - gx0];
      }
  }
  store8<DT>(p.out, (size_t)plane * p.H * p.W + (size_t)iy * p.W + ixb, acc, p.W - ixb >= 8 ? 8 : p.W - ixb);
}

// stage 1: partial[(plane * tiles + tile) * 9 + t] = sum over the tile's output pixels
template <int DT, int S>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwtParams p) {
  constexpr int NR = (DT_TH - 1) * S + 3, NC = (DT_TW - 1) * S + 3, LD = NC | 1;
  constexpr int VN = Vec16<DT>::n;
  extern __shared__ float xt[];
  __shared__ float red[4][9];
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int plane = blockIdx.y;
  const int oy0 = ty * DT_TH, ox0 = tx * DT_TW;
  stage_tile<DT>(xt, LD, p.a, (size_t)plane * p.H * p.W, p.H, p.W, oy0 * S - 1, ox0 * S - 1, NR, NC);
  __syncthreads();
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const int row = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int oy = oy0 + row, ox = ox0 + g * 8;
  if (oy < p.Ho && ox < p.Wo) {
    const size_t gi = (size_t)plane * p.Ho * p.Wo + (size_t)oy * p.Wo + ox;
    float gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = 0.f;
    if (p.Wo - ox >= 8 && (p.Wo % VN) == 0 && (((size_t)plane * p.Ho * p.Wo) % VN) == 0 && (((uintptr_t)p.b) & 15u) == 0) {
      load16<DT>(p.b, gi, gv);
      if (VN == 4) {
        float hi[8];
        load16<DT>(p.b, gi + 4, hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[4 + e] = hi[e];
      }
    } else {
      for (int e = 0; e < 8 && ox + e < p.Wo; ++e) gv[e] = ldf<DT>(p.b, gi + e);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* xr = xt + (row * S + ky) * LD + g * 8 * S;
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += gv[e] * xr[e * S + kx];
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

// stage 2: dw[c][t] = sum over images and tiles: one wave per channel, lane l adds the (image, tile) partials
// l, l + 64, ... in index order, then a fixed butterfly -- deterministic
__global__ __launch_bounds__(64) void dw_wgrad_reduce_kernel(const float* partial, float* dw, int N, int C, int tiles) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const int total = N * tiles;
  for (int j = lane; j < total; j += 64) {
    const int n = j / tiles, k = j - n * tiles;
    const float* q = partial + ((size_t)(n * C + c) * tiles + k) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += q[t];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = acc[t];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) dw[c * 9 + t] = v;
  }
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
  p.tiles_x = (ow + DT_TW - 1) / DT_TW;
  p.tiles_y = (oh + DT_TH - 1) / DT_TH;
  return p;
}

// lds1 / lds2: dynamic LDS bytes of the stride-1 / stride-2 instantiation (the staged tile)
#define SSDK_DWT_LAUNCH(KERNEL, grid, lds1, lds2)                                                                \
  do {                                                                                                           \
    if (dtype == SSDK_F32) {                                                                                     \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_F32, 1>), grid, dim3(256), lds1, (hipStream_t)stream, p);  \
      else hipLaunchKernelGGL((KERNEL<SSDK_F32, 2>), grid, dim3(256), lds2, (hipStream_t)stream, p);              \
    } else if (dtype == SSDK_BF16) {                                                                             \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_BF16, 1>), grid, dim3(256), lds1, (hipStream_t)stream, p); \
      else hipLaunchKernelGGL((KERNEL<SSDK_BF16, 2>), grid, dim3(256), lds2, (hipStream_t)stream, p);             \
    } else {                                                                                                     \
      if (stride == 1) hipLaunchKernelGGL((KERNEL<SSDK_F16, 1>), grid, dim3(256), lds1, (hipStream_t)stream, p);  \
      else hipLaunchKernelGGL((KERNEL<SSDK_F16, 2>), grid, dim3(256), lds2, (hipStream_t)stream, p);              \
    }                                                                                                            \
  } while (0)

static constexpr size_t dwt_lds_in(int s) {  // tile of the input behind a DT_TH x DT_TW output tile
  return (size_t)((DT_TH - 1) * s + 3) * ((((DT_TW - 1) * s + 3)) | 1) * sizeof(float);
}
static constexpr size_t dwt_lds_dy(int s) {  // tile of dy behind a DT_TH x DT_TW tile of dx
  return (size_t)((DT_TH + 1) / s + 2) * (((DT_TW + 1) / s + 2) | 1) * sizeof(float);
}

// ssdk_dwplane.hip: the whole-row kernels (0: launched, 1: not taken)
int launch_dwp_fwd(const void* x, const void* w, void* y, int N, int C, int H, int W, int stride, int dtype, hipStream_t stream,
                   float* stats = nullptr, int* groups_out = nullptr, const float* coef = nullptr, int act = 0);
int dwp_fwd_groups(int N, int C, int H, int W, int stride, int dtype);
int reduce_rows_fixed_order(const float* src, float* mid, float* dst, unsigned n, unsigned rows, hipStream_t st);  // ssdk_pwtrain.hip
int launch_dwp_dgrad(const void* dy, const void* w, void* dx, int N, int C, int H, int W, int stride, int dtype, hipStream_t stream);
size_t dwp_wgrad_workspace_bytes(int N, int C, int H, int W, int stride);
int dwp_affine_ok(int N, int C, int H, int W, int stride, int dtype);
int launch_dwp_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int C, int H, int W,
                     int stride, int dtype, hipStream_t stream, const float* coef = nullptr, int act = 0);

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_dwconv_fwd(const void* x, const void* w, void* y, int N, int C, int H, int W, int stride, int dtype,
                               void* stream) {
  const int rc = dwt_check("dwconv_fwd", x, w, y, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (launch_dwp_fwd(x, w, y, N, C, H, W, stride, dtype, (hipStream_t)stream) == 0) return check_launch("dwp_fwd_kernel");
  DwtParams p = dwt_params(x, w, y, N, C, H, W, stride, dtype, false);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_fwd_kernel, grid, dwt_lds_in(1), dwt_lds_in(2));
  return check_launch("dw_fwd_kernel");
}

// forward + sums [C][2] = per channel (sum y, sum y^2) over N * Ho * Wo of the outputs (fp32, before the store's rounding): the batch
// statistics of the BatchNorm behind the convolution (ssdk_bn_act_train_fwd_sums).  workspace: ssdk_dwconv_fwd_stats_workspace_bytes
// (0: this geometry runs on the tiled kernels, which do not produce statistics -- call ssdk_dwconv_fwd).
extern "C" size_t ssdk_dwconv_fwd_stats_workspace_bytes(int N, int C, int H, int W, int stride, int dtype) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2)) return 0;
  const int g = dwp_fwd_groups(N, C, H, W, stride, dtype);
  if (g <= 0 || g > 4096) return 0;
  return ((size_t)g + (size_t)((g + 63) / 64)) * (size_t)C * 2 * sizeof(float);
}

extern "C" int ssdk_dwconv_fwd_stats(const void* x, const void* w, void* y, float* sums, void* workspace, size_t workspace_bytes,
                                     int N, int C, int H, int W, int stride, int dtype, void* stream) {
  const int rc = dwt_check("dwconv_fwd_stats", x, w, y, N, C, H, W, stride, dtype);
  if (rc) return rc;
  const size_t need = ssdk_dwconv_fwd_stats_workspace_bytes(N, C, H, W, stride, dtype);
  if (!sums || !workspace || need == 0 || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
    set_error("dwconv_fwd_stats: no workspace / this geometry does not run on the whole-row kernels");
    return SSDK_E_WORKSPACE;
  }
  int groups = 0;
  if (launch_dwp_fwd(x, w, y, N, C, H, W, stride, dtype, (hipStream_t)stream, (float*)workspace, &groups) != 0) {
    set_error("dwconv_fwd_stats: the whole-row kernels declined");
    return SSDK_E_BADARG;
  }
  const unsigned n = (unsigned)C * 2u;
  return reduce_rows_fixed_order((const float*)workspace, (float*)workspace + (size_t)groups * n, sums, n, (unsigned)groups,
                                 (hipStream_t)stream);
}

// The depthwise convolution behind a DEFERRED BatchNorm (round 6): x is the BatchNorm's INPUT, coef [C][4] its per-channel
// (a, b, ., .) from ssdk_bn_act_train_stats, act its folded activation; the kernels stage act(a x + b) rounded to the dtype --
// the tensor bn_apply would have written -- so the BatchNorm's output never exists in memory.  16 bit only.
//   ssdk_dwconv_affine_supported   1 when forward AND weight gradient of this geometry run on the whole-row kernels
//   ssdk_dwconv_fwd_affine         y (+ sums [C][2] for the next BatchNorm when sums != NULL: workspace as ssdk_dwconv_fwd_stats)
//   ssdk_dwconv_bwd_weight_affine  dw from (x, coef, act) and dy
extern "C" int ssdk_dwconv_affine_supported(int N, int C, int H, int W, int stride, int dtype) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2) || (dtype != SSDK_BF16 && dtype != SSDK_F16)) return 0;
  return dwp_affine_ok(N, C, H, W, stride, dtype);
}

extern "C" int ssdk_dwconv_fwd_affine(const void* x, const float* coef, int act, const void* w, void* y, float* sums, void* workspace,
                                      size_t workspace_bytes, int N, int C, int H, int W, int stride, int dtype, void* stream) {
  const int rc = dwt_check("dwconv_fwd_affine", x, w, y, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (!coef || act < 0 || act > 2 || !ssdk_dwconv_affine_supported(N, C, H, W, stride, dtype)) {
    set_error("dwconv_fwd_affine: 16-bit tensors on the whole-row kernels only, act 0 | 1 | 2, coef must not be null");
    return SSDK_E_BADARG;
  }
  int groups = 0;
  if (sums) {
    const size_t need = ssdk_dwconv_fwd_stats_workspace_bytes(N, C, H, W, stride, dtype);
    if (!workspace || need == 0 || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
      set_error("dwconv_fwd_affine: workspace too small or misaligned");
      return SSDK_E_WORKSPACE;
    }
  }
  if (launch_dwp_fwd(x, w, y, N, C, H, W, stride, dtype, (hipStream_t)stream, sums ? (float*)workspace : nullptr, &groups, coef, act) != 0) {
    set_error("dwconv_fwd_affine: the whole-row kernels declined");
    return SSDK_E_BADARG;
  }
  int rc2 = check_launch("dwp_fwd_kernel");
  if (rc2 || !sums) return rc2;
  const unsigned n = (unsigned)C * 2u;
  return reduce_rows_fixed_order((const float*)workspace, (float*)workspace + (size_t)groups * n, sums, n, (unsigned)groups,
                                 (hipStream_t)stream);
}

extern "C" int ssdk_dwconv_bwd_weight_affine(const void* x, const float* coef, int act, const void* dy, float* dw, void* workspace,
                                             size_t workspace_bytes, int N, int C, int H, int W, int stride, int dtype, void* stream) {
  const int rc = dwt_check("dwconv_bwd_weight_affine", x, dy, dw, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (!coef || act < 0 || act > 2 || !workspace || workspace_bytes < ssdk_dwconv_bwd_weight_workspace_bytes(N, C, H, W, stride) ||
      !ssdk_dwconv_affine_supported(N, C, H, W, stride, dtype)) {
    set_error("dwconv_bwd_weight_affine: bad argument / workspace too small / geometry not on the whole-row kernels");
    return SSDK_E_BADARG;
  }
  if (launch_dwp_wgrad(x, dy, dw, workspace, workspace_bytes, N, C, H, W, stride, dtype, (hipStream_t)stream, coef, act) != 0) {
    set_error("dwconv_bwd_weight_affine: the whole-row kernels declined");
    return SSDK_E_BADARG;
  }
  return check_launch("dwp_wgrad_kernel");
}

extern "C" int ssdk_dwconv_bwd_data(const void* dy, const void* w, void* dx, int N, int C, int H, int W, int stride,
                                    int dtype, void* stream) {
  const int rc = dwt_check("dwconv_bwd_data", dy, w, dx, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (launch_dwp_dgrad(dy, w, dx, N, C, H, W, stride, dtype, (hipStream_t)stream) == 0) return check_launch("dwp_dgrad_kernel");
  DwtParams p = dwt_params(dy, w, dx, N, C, H, W, stride, dtype, true);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_dgrad_kernel, grid, dwt_lds_dy(1), dwt_lds_dy(2));
  return check_launch("dw_dgrad_kernel");
}

extern "C" size_t ssdk_dwconv_bwd_weight_workspace_bytes(int N, int C, int H, int W, int stride) {
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const size_t tiled = (size_t)N * C * ((Wo + DT_TW - 1) / DT_TW) * ((Ho + DT_TH - 1) / DT_TH) * 9 * sizeof(float);
  const size_t rows = dwp_wgrad_workspace_bytes(N, C, H, W, stride);  // either kernel family may take the call
  return tiled > rows ? tiled : rows;
}

extern "C" int ssdk_dwconv_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                      int N, int C, int H, int W, int stride, int dtype, void* stream) {
  const int rc = dwt_check("dwconv_bwd_weight", x, dy, dw, N, C, H, W, stride, dtype);
  if (rc) return rc;
  if (!workspace || workspace_bytes < ssdk_dwconv_bwd_weight_workspace_bytes(N, C, H, W, stride)) {
    set_error("dwconv_bwd_weight: workspace too small");
    return SSDK_E_BADARG;
  }
  if (launch_dwp_wgrad(x, dy, dw, workspace, workspace_bytes, N, C, H, W, stride, dtype, (hipStream_t)stream) == 0)
    return check_launch("dwp_wgrad_kernel");
  DwtParams p = dwt_params(x, dy, workspace, N, C, H, W, stride, dtype, false);
  const int tiles = p.tiles_x * p.tiles_y;
  const dim3 grid((unsigned)tiles, (unsigned)(N * C));
  SSDK_DWT_LAUNCH(dw_wgrad_kernel, grid, dwt_lds_in(1), dwt_lds_in(2));
  int rc2 = check_launch("dw_wgrad_kernel");
  if (rc2) return rc2;
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream,
                     (const float*)workspace, dw, N, C, tiles);
  return check_launch("dw_wgrad_reduce_kernel");
}
