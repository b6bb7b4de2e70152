// ssdk_mbconv.hip -- one MobileNetV2 inverted-residual block as ONE kernel on gfx950.
//
// Reference: the torchvision InvertedResidual blocks the reference's backbone is built from
// (ssds/modeling/nets/mobilenet.py:56, 84-89): 1x1 expand (BN, ReLU6) -> 3x3 depthwise stride s (BN, ReLU6)
// -> 1x1 linear projection (BN) [+ x].  Run layer by layer, the expanded tensor (6x the block input) is
// written to and read from HBM twice and dominates the network's traffic (0.8 GB per layer at 256x256,
// batch 64).  Here a workgroup owns an 8x8 output tile of one image and keeps everything on chip:
//
//   sX   the input tile with its halo ((8s+2)^2 pixels x Cin) in LDS, loaded once
//   per 32-channel chunk of the hidden dimension:
//     P1  expand:   E[p][hc]  = relu6(se * sum_ci X[p][ci] We[hc][ci] + be)     MFMA 16x16x32, -> LDS
//                   (zero outside the image: the depthwise conv pads the EXPANDED activation)
//     P2  depthwise D[q][hc]  = relu6(sd * sum_taps E[..][hc] Wd[tap][hc] + bd)  VALU fp32, -> LDS
//     P3  project   Y[q][co] += sum_hc D[q][hc] Wp[co][hc]                       MFMA, accumulators in VGPRs
//   epilogue: Y * sp + bp (+ x from sX), 8-byte stores, NHWC.
//
// Two barriers per chunk.  HBM traffic = block input (with a 1.13-1.56x halo) + block output + weights
// from L2: the roofline of the fused block is its input+output bytes.
#include "ssdk_common.h"

namespace ssdk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MbParams {
  const u16* x;
  u16* y;
  const u16* we;  // [Chid][Cin]
  const float* se;
  const float* be;
  const u16* wd;  // [3][3][Chid]
  const float* sd;
  const float* bd;
  const u16* wp;  // [Cout][Chid]
  const float* sp;
  const float* bp;
  int N, H, W, Cin, Chid, Cout, Ho, Wo, residual;
  int tiles_x, tiles_y;
  int xs;  // LDS row stride of sX in bytes
};

template <int DT> __device__ __forceinline__ u32 mb_to16(float v) {
  if constexpr (DT == SSDK_BF16) {
    u32 b = __builtin_bit_cast(u32, v);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
  } else {
    _Float16 h = (_Float16)v;
    return (u32)__builtin_bit_cast(u16, h);
  }
}
template <int DT> __device__ __forceinline__ float mb_from16(u32 h) {
  if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
  else return f16_bits_to_f32(h);
}
template <int DT>
__device__ __forceinline__ f32x4 mb_mfma(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (DT == SSDK_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float relu6f(float v) { return v < 0.f ? 0.f : (v > 6.f ? 6.f : v); }

constexpr int kMbThreads = 256;
constexpr int HC = 32;       // hidden channels per chunk
constexpr int ES = 80;       // LDS row stride (bytes) of sE / sD: 32 ch * 2 B + 16 B pad
constexpr int MAX_KS = 5;    // Cin <= 160

template <int DT, int S, int NFO, int KSMAX>
__global__ __launch_bounds__(kMbThreads) void mbconv_kernel(const MbParams p) {
  constexpr int RW = 8 * S + (3 - S);          // 10 (s=1) or 17 (s=2) input columns / rows per tile
  constexpr int P = RW * RW;                   // region pixels
  constexpr int MF = (P + 15) / 16;            // m-frags of the expand GEMM
  constexpr int P16 = MF * 16;
  constexpr int MFW = (MF + 3) / 4;            // m-frags per wave (max)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int XS = p.xs;
  unsigned char* sX = smem;                                  // [P16][XS]
  unsigned char* sE = sX + (size_t)P16 * XS;                 // [P16][ES]
  unsigned char* sD = sE + (size_t)P16 * ES;                 // [64][ES]

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 fr = lane & 15u, fg = lane >> 4;
  u32 bid = blockIdx.x;
  const int tx = (int)(bid % (u32)p.tiles_x);
  bid /= (u32)p.tiles_x;
  const int ty = (int)(bid % (u32)p.tiles_y);
  const int n = (int)(bid / (u32)p.tiles_y);
  const int oy0 = ty * 8, ox0 = tx * 8;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int Cin = p.Cin, Chid = p.Chid, Cout = p.Cout, H = p.H, W = p.W;
  const int KS = (Cin + 31) / 32;

  // ---- phase 0: input tile + halo -> sX (zeros outside the image and in the padding rows) -------------
  {
    const int cpr = Cin / 8;  // 16-byte chunks per pixel
    const int total = P16 * cpr;
    const u16* xin = p.x + (size_t)n * H * W * Cin;
    for (int q = (int)tid; q < total; q += kMbThreads) {
      const int pix = q / cpr, c = q % cpr;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pix < P) {
        const int iy = iy0 + pix / RW, ix = ix0 + pix % RW;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          v = *reinterpret_cast<const u32x4*>(xin + ((size_t)iy * W + ix) * Cin + c * 8);
      }
      *reinterpret_cast<u32x4*>(sX + (size_t)pix * XS + c * 16) = v;
    }
  }
  // validity of the region pixels this lane produces in P1 (bit i: m-frag wave + 4*i)
  u32 pvalid = 0;
#pragma unroll
  for (int i = 0; i < MFW; ++i) {
    const int pix = ((int)wave + 4 * i) * 16 + (int)fr;
    if (pix < P) {
      const int iy = iy0 + pix / RW, ix = ix0 + pix % RW;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) pvalid |= 1u << i;
    }
  }
  f32x4 yacc[NFO];
#pragma unroll
  for (int j = 0; j < NFO; ++j) yacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int nchunks = (Chid + HC - 1) / HC;
  const u32 d_px = tid >> 2, d_cg = tid & 3u;      // P2 role: output pixel, 8-channel group
  const u32 d_oy = d_px >> 3, d_ox = d_px & 7u;

  for (int c = 0; c < nchunks; ++c) {
    const int hc0 = c * HC;
    // weights of this chunk straight from global/L2 into fragment registers (issued first: their latency
    // hides under the LDS reads / MFMAs of P1)
    u32x4 wef[2][KSMAX];
#pragma unroll
    for (int jf = 0; jf < 2; ++jf)
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const int hc = hc0 + jf * 16 + (int)fr, k = ks * 32 + (int)fg * 8;
        if (ks < KS && hc < Chid && k < Cin) v = *reinterpret_cast<const u32x4*>(p.we + (size_t)hc * Cin + k);
        wef[jf][ks] = v;
      }
    u32x4 wpf[NFO];
#pragma unroll
    for (int j = 0; j < NFO; ++j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      const int co = j * 16 + (int)fr, k = hc0 + (int)fg * 8;
      if (co < Cout && k < Chid) v = *reinterpret_cast<const u32x4*>(p.wp + (size_t)co * Chid + k);
      wpf[j] = v;
    }
    // ---- P1: expand the region for channels [hc0, hc0+32) -> sE -----------------------------------
    float se4[2][4], be4[2][4];
#pragma unroll
    for (int jf = 0; jf < 2; ++jf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hc = hc0 + jf * 16 + (int)fg * 4 + r;
        se4[jf][r] = hc < Chid ? p.se[hc] : 0.f;
        be4[jf][r] = hc < Chid ? p.be[hc] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < MFW; ++i) {
      const int mf = (int)wave + 4 * i;
      if (mf < MF) {  // wave-uniform
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* xrow = sX + (size_t)(mf * 16 + (int)fr) * XS;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
          if (ks < KS) {
            u32x4 xf = {0u, 0u, 0u, 0u};
            const int k = ks * 32 + (int)fg * 8;
            if (k < Cin) xf = *reinterpret_cast<const u32x4*>(xrow + k * 2);
            e0 = mb_mfma<DT>(wef[0][ks], xf, e0);  // D[hc = fg*4+r][pixel = fr]
            e1 = mb_mfma<DT>(wef[1][ks], xf, e1);
          }
        }
        const bool ok = (pvalid >> i) & 1u;
        u32 h0[4], h1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h0[r] = ok ? mb_to16<DT>(relu6f(e0[r] * se4[0][r] + be4[0][r])) : 0u;
          h1[r] = ok ? mb_to16<DT>(relu6f(e1[r] * se4[1][r] + be4[1][r])) : 0u;
        }
        unsigned char* erow = sE + (size_t)(mf * 16 + (int)fr) * ES;
        *reinterpret_cast<uint2*>(erow + (fg * 4) * 2) = make_uint2(h0[0] | (h0[1] << 16), h0[2] | (h0[3] << 16));
        *reinterpret_cast<uint2*>(erow + (16 + fg * 4) * 2) = make_uint2(h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16));
      }
    }
    __syncthreads();
    // ---- P2: depthwise 3x3 stride S on the chunk -> sD ------------------------------------------------
    {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      const int ch = hc0 + (int)d_cg * 8;
      const bool chok = ch < Chid;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int rp = ((int)d_oy * S + ky) * RW + (int)d_ox * S + kx;
          const u32x4 ev = *reinterpret_cast<const u32x4*>(sE + (size_t)rp * ES + d_cg * 16);
          u32x4 wv = {0u, 0u, 0u, 0u};
          if (chok) wv = *reinterpret_cast<const u32x4*>(p.wd + (size_t)(ky * 3 + kx) * Chid + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[2 * e] = fmaf(mb_from16<DT>(ev[e] & 0xffffu), mb_from16<DT>(wv[e] & 0xffffu), acc[2 * e]);
            acc[2 * e + 1] = fmaf(mb_from16<DT>(ev[e] >> 16), mb_from16<DT>(wv[e] >> 16), acc[2 * e + 1]);
          }
        }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s0 = 0.f, b0 = 0.f, s1 = 0.f, b1 = 0.f;
        if (chok) {
          s0 = p.sd[ch + 2 * e];
          b0 = p.bd[ch + 2 * e];
          s1 = p.sd[ch + 2 * e + 1];
          b1 = p.bd[ch + 2 * e + 1];
        }
        const u32 lo = chok ? mb_to16<DT>(relu6f(acc[2 * e] * s0 + b0)) : 0u;
        const u32 hi = chok ? mb_to16<DT>(relu6f(acc[2 * e + 1] * s1 + b1)) : 0u;
        o[e] = lo | (hi << 16);
      }
      *reinterpret_cast<u32x4*>(sD + (size_t)d_px * ES + d_cg * 16) = o;
    }
    __syncthreads();
    // ---- P3: project: wave w owns output pixels [16w, 16w+16) x all Cout ----------------------------
    {
      const u32x4 df = *reinterpret_cast<const u32x4*>(sD + (size_t)(wave * 16 + fr) * ES + fg * 16);
#pragma unroll
      for (int j = 0; j < NFO; ++j) yacc[j] = mb_mfma<DT>(wpf[j], df, yacc[j]);  // D[co = fg*4+r][px = fr]
    }
    // (no barrier here: the next chunk's P1 writes sE, which P2 of this chunk finished reading before the
    //  barrier above; its P2 writes sD only after the barrier that follows its P1, i.e. after every wave
    //  has passed this P3)
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  const int q = (int)wave * 16 + (int)fr;  // output pixel inside the tile
  const int oy = oy0 + (q >> 3), ox = ox0 + (q & 7);
  if (oy < p.Ho && ox < p.Wo) {
    u16* yrow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * Cout;
    const unsigned char* xres = sX + (size_t)(((q >> 3) * S + 1) * RW + (q & 7) * S + 1) * XS;
#pragma unroll
    for (int j = 0; j < NFO; ++j) {
      const int co = j * 16 + (int)fg * 4;
      if (co < Cout) {  // Cout is a multiple of 8, so 4-channel groups are all-or-nothing
        u32 h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = yacc[j][r] * p.sp[co + r] + p.bp[co + r];
          h[r] = mb_to16<DT>(v);
        }
        if (p.residual) {
          const uint2 xv = *reinterpret_cast<const uint2*>(xres + co * 2);
          const u32 xr[4] = {xv.x & 0xffffu, xv.x >> 16, xv.y & 0xffffu, xv.y >> 16};
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = mb_to16<DT>(mb_from16<DT>(h[r]) + mb_from16<DT>(xr[r]));
        }
        *reinterpret_cast<uint2*>(yrow + co) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      }
    }
  }
}

template <int DT, int S, int NFO, int KSMAX>
static void launch_one(const MbParams& p, size_t lds, unsigned grid, hipStream_t stream) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_kernel<DT, S, NFO, KSMAX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mbconv_kernel<DT, S, NFO, KSMAX>), dim3(grid), dim3(kMbThreads), lds, stream, p);
}

// (k-steps of the expand GEMM, n-frags of the projection) pairs of the MobileNetV2 family get their own
// instantiation (register budget = occupancy); everything else runs on the most general one.
template <int DT, int S>
static int launch_mb(const MbParams& p, int ks, int nfo, size_t lds, unsigned grid, hipStream_t stream) {
  if (ks <= 1 && nfo <= 2) launch_one<DT, S, 2, 1>(p, lds, grid, stream);
  else if (ks <= 1 && nfo <= 4) launch_one<DT, S, 4, 1>(p, lds, grid, stream);
  else if (ks <= 2 && nfo <= 4) launch_one<DT, S, 4, 2>(p, lds, grid, stream);
  else if (ks <= 2 && nfo <= 6) launch_one<DT, S, 6, 2>(p, lds, grid, stream);
  else if (ks <= 3 && nfo <= 6) launch_one<DT, S, 6, 3>(p, lds, grid, stream);
  else if (ks <= 3 && nfo <= 10) launch_one<DT, S, 10, 3>(p, lds, grid, stream);
  else if (nfo <= 10) launch_one<DT, S, 10, 5>(p, lds, grid, stream);
  else launch_one<DT, S, 20, 5>(p, lds, grid, stream);
  return check_launch("mbconv_kernel");
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_mbconv(const ssdk_mbconv_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !d->x || !d->y || !d->w_expand || !d->w_dw || !d->w_project || !d->scale_expand || !d->bias_expand ||
      !d->scale_dw || !d->bias_dw || !d->scale_project || !d->bias_project) {
    set_error("mbconv: null pointer");
    return SSDK_E_BADARG;
  }
  if (d->dtype != SSDK_BF16 && d->dtype != SSDK_F16) {
    set_error("mbconv: dtype must be bf16 or f16");
    return SSDK_E_BADARG;
  }
  if (d->N < 1 || d->H < 1 || d->W < 1 || (d->stride != 1 && d->stride != 2) || d->Cin < 8 || (d->Cin % 8) ||
      d->Cin > 32 * MAX_KS || (d->Chid % 8) || d->Chid < 8 || (d->Cout % 8) || d->Cout < 8 || d->Cout > 320 ||
      (d->residual && (d->stride != 1 || d->Cin != d->Cout))) {
    set_error("mbconv: unsupported geometry Cin=%d Chid=%d Cout=%d stride=%d residual=%d (Cin<=160, Cout<=320, "
              "channels %% 8 == 0)", d->Cin, d->Chid, d->Cout, d->stride, d->residual);
    return SSDK_E_BADARG;
  }
  MbParams p;
  p.x = (const u16*)d->x;
  p.y = (u16*)d->y;
  p.we = (const u16*)d->w_expand;
  p.se = d->scale_expand;
  p.be = d->bias_expand;
  p.wd = (const u16*)d->w_dw;
  p.sd = d->scale_dw;
  p.bd = d->bias_dw;
  p.wp = (const u16*)d->w_project;
  p.sp = d->scale_project;
  p.bp = d->bias_project;
  p.N = d->N;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.Chid = d->Chid;
  p.Cout = d->Cout;
  p.Ho = (d->H + 2 - 3) / d->stride + 1;
  p.Wo = (d->W + 2 - 3) / d->stride + 1;
  p.residual = d->residual;
  p.tiles_x = (p.Wo + 7) / 8;
  p.tiles_y = (p.Ho + 7) / 8;
  const int cpr = d->Cin / 8;
  p.xs = d->Cin * 2 + ((cpr % 2 == 0) ? 16 : 0);  // odd number of 16-byte slots per row
  const int rw = d->stride == 1 ? 10 : 17;
  const int p16 = ((rw * rw + 15) / 16) * 16;
  const size_t lds = (size_t)p16 * p.xs + (size_t)p16 * ES + 64 * ES;
  if (lds > 160 * 1024) {
    set_error("mbconv: tile needs %zu bytes of LDS", lds);
    return SSDK_E_BADARG;
  }
  const unsigned grid = (unsigned)((long)d->N * p.tiles_x * p.tiles_y);
  const int nfo = (d->Cout + 15) / 16, ks = (d->Cin + 31) / 32;
  if (d->dtype == SSDK_BF16)
    return d->stride == 1 ? launch_mb<SSDK_BF16, 1>(p, ks, nfo, lds, grid, stream)
                          : launch_mb<SSDK_BF16, 2>(p, ks, nfo, lds, grid, stream);
  return d->stride == 1 ? launch_mb<SSDK_F16, 1>(p, ks, nfo, lds, grid, stream)
                        : launch_mb<SSDK_F16, 2>(p, ks, nfo, lds, grid, stream);
}
