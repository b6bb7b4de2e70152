// ssdk_preproc.hip -- image preprocessing of the detector front door on gfx950 (HBM-bound, one pass).
//
// Reference: SSDDetector.__call__ (ssds/ssds.py:47-57): HWC -> CHW transpose on the host, float32 upload,
// `(x - mean) / std` as two full-size tensor ops, then the network's own cast.  Here the RAW image batch is uploaded
// (uint8: a quarter of the PCIe bytes of float32) and one launch does transpose + normalise + cast:
//     y[n][c][h][w] = dtype( (float(x[n][h][w][c] | x[n][c][h][w]) - mean[c]) / std[c] )
// with the reference's fp32 operation order (subtract, then divide: bit-exact with the torch expression followed
// by .to(dtype)).  A lane owns 8 consecutive pixels of one output row and channel -> 16-byte stores (2-byte dtypes).
// Algorithmic bytes: source once + destination once.
#include "ssdk_conv_common.h"

namespace ssdk {

struct PreParams {
  const void* x;
  void* y;
  int N, H, W, C, src_dtype, src_layout, dst_dtype;
  float mean[4], std[4];
  long total;  // N*C*H*ceil(W/8)
};

__device__ __forceinline__ float pre_load(const PreParams& p, size_t i) {
  if (p.src_dtype == SSDK_U8) return (float)((const unsigned char*)p.x)[i];
  if (p.src_dtype == SSDK_F32) return ((const float*)p.x)[i];
  if (p.src_dtype == SSDK_BF16) return bf16_bits_to_f32(((const u16*)p.x)[i]);
  return f16_bits_to_f32(((const u16*)p.x)[i]);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const int wg = (p.W + 7) / 8;
  const int x0 = (int)(t % wg) * 8;
  long r = t / wg;
  const int y = (int)(r % p.H);
  r /= p.H;
  const int c = (int)(r % p.C);
  const int n = (int)(r / p.C);
  const float mean = p.mean[c], sd = p.std[c];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int x = x0 + e;
    v[e] = 0.f;
    if (x < p.W) {
      const size_t i = p.src_layout == LAYOUT_NHWC ? (((size_t)n * p.H + y) * p.W + x) * p.C + c
                                                   : (((size_t)n * p.C + c) * p.H + y) * p.W + x;
      v[e] = (pre_load(p, i) - mean) / sd;  // reference order: subtract, divide (ssds.py:55)
    }
  }
  const size_t o = (((size_t)n * p.C + c) * p.H + y) * p.W + x0;
  if (p.dst_dtype == SSDK_F32) {
    float* dst = (float*)p.y + o;
    for (int e = 0; e < 8 && x0 + e < p.W; ++e) dst[e] = v[e];
    return;
  }
  u16* dst = (u16*)p.y + o;
  u32 h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = p.dst_dtype == SSDK_BF16 ? f32_to_bits16<SSDK_BF16>(v[e]) : f32_to_bits16<SSDK_F16>(v[e]);
  if (x0 + 8 <= p.W && (((uintptr_t)dst) & 15u) == 0) {
    *reinterpret_cast<u32x4*>(dst) = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  } else {
    for (int e = 0; e < 8 && x0 + e < p.W; ++e) dst[e] = (u16)h[e];
  }
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_preprocess(const void* x, int src_dtype, int src_layout, int N, int H, int W, int C,
                               const float* mean, const float* std, void* y, int dst_dtype, void* stream) {
  if (!x || !y || !mean || !std) {
    set_error("preprocess: null pointer");
    return SSDK_E_BADARG;
  }
  if (N < 1 || H < 1 || W < 1 || C < 1 || C > 4 || (src_layout != LAYOUT_NHWC && src_layout != LAYOUT_NCHW) ||
      (src_dtype != SSDK_U8 && src_dtype != SSDK_F32 && src_dtype != SSDK_BF16 && src_dtype != SSDK_F16) ||
      (dst_dtype != SSDK_F32 && dst_dtype != SSDK_BF16 && dst_dtype != SSDK_F16)) {
    set_error("preprocess: bad arguments (N=%d H=%d W=%d C=%d <= 4, src dtype %d layout %d, dst dtype %d)", N, H, W, C,
              src_dtype, src_layout, dst_dtype);
    return SSDK_E_BADARG;
  }
  PreParams p;
  p.x = x;
  p.y = y;
  p.N = N;
  p.H = H;
  p.W = W;
  p.C = C;
  p.src_dtype = src_dtype;
  p.src_layout = src_layout;
  p.dst_dtype = dst_dtype;
  for (int c = 0; c < 4; ++c) {
    p.mean[c] = c < C ? mean[c] : 0.f;
    p.std[c] = c < C ? std[c] : 1.f;
  }
  p.total = (long)N * C * H * ((W + 7) / 8);
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((p.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("preprocess_kernel");
}
