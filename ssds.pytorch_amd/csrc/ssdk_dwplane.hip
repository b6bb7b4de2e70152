// ssdk_dwplane.hip -- the depthwise 3x3 training passes (forward, input gradient, weight gradient; NCHW) as WHOLE-ROW
// bands of the (image, channel) planes -- the kernels behind ssdk_dwconv_fwd / _bwd_data / _bwd_weight for the plane
// sizes the detection backbones have (150, 75, 38, 19, 10 and 5 pixels a side: none of them a multiple of 8).
//
// Why (profiles/r02_train_kernel_split.txt): the 32 x 64 tiles of ssdk_dwtrain.hip stage their input with aligned
// 16-byte reads only when the plane width is a multiple of 8 (never, here: every layer took the element-by-element
// path), and on the 19 x 19 and 10 x 10 maps a 256-thread tile has 57 and 20 busy threads.  The three passes ran at
// 1/5 .. 1/11 of their HBM time (6.0 of the 25.7 ms of the training step).
//
// Here a workgroup owns one CHANNEL c and either G whole planes (images n0 .. n0+G-1 of that channel: small maps) or
// one band of rows of one plane (large maps):
//   * its rows are whole rows, so every global access is a run of consecutive elements: 16-byte loads / stores at
//     2-byte alignment (the hardware's unaligned access mode, which the compiler already relies on), no per-row
//     alignment case;
//   * each piece (plane or band) is staged in LDS with a zero column either side and zero rows above / below the plane,
//     so the window arithmetic has no boundary case at all.  16-bit tensors stay 16-bit in LDS, pixel x at column x + 8
//     of a row whose stride is a multiple of 8: a staged chunk is ONE 16-byte load and ONE ds_write_b128, a window row is
//     one (stride 2: two) aligned ds_read_b128 plus its edge columns (as fp32, one word per pixel, the staging and the
//     window cost 60 instructions per pixel and the kernels were instruction-bound at 2.3 TB/s); fp32 tensors keep the
//     one-word layout (pixel x at column x + 1);
//   * a thread computes 8 consecutive pixels of one row from a 3 x (8S+2) window read once from LDS; threads run over
//     the (piece, row, segment) units of the workgroup, ~1000 units per workgroup whatever the plane size;
//   * one channel per workgroup: the nine weights are uniform, and the weight gradient accumulates in registers over all
//     the units of a thread, then one fixed-order reduction per workgroup -- partial sums [group][C][9], added up per
//     channel in index order by a second kernel.  No float atomics: bit-reproducible.
//   * logical workgroups are dealt to the XCDs in runs (consecutive channels = adjacent memory stay behind one L2).
#include "ssdk_conv_common.h"

namespace ssdk {

struct DwpParams {
  const void* a;   // staged tensor: x (forward, weight gradient) | dy (input gradient)
  const void* b;   // w [C][9] in the activation dtype (forward, input gradient) | dy (weight gradient)
  void* out;       // y | dx | partial sums [groups][C][9] fp32
  float* stats;    // forward only, optional: [groups][C][2] per-workgroup (sum y, sum y^2) of its outputs (fp32, before the store)
  const float* coef;  // AFF instances (forward, weight gradient; 16 bit): [C][4] = (a, b, ., .) of the BatchNorm whose OUTPUT is the
  int act;            // staged tensor: the kernel stages act(a x + b) rounded to the dtype from the BatchNorm's INPUT x (0 none | 1 ReLU6 | 2 ReLU)
  int N, C;
  int Hs, Ws;      // plane of the staged tensor
  int Ht, Wt;      // plane of the thread space: y (forward), dx (input gradient), dy (weight gradient)
  int G, T, TR;    // images per workgroup; bands per plane; rows of the thread space per band
  int seg;         // 8-pixel segments per row
  int LD, SR, CH;  // LDS row stride (floats); staged rows per piece; 8-column chunks per LDS row
  int UP;          // units per piece = TR * seg
  int nwg, nwg8;   // logical workgroups; ceil(nwg / 8)
  float rcpCH, rcpSR, rcpUP, rcpSEG;
  size_t total_a, total_b;  // elements of a / of the dy tensor behind b (weight gradient)
  size_t img_a;             // elements of one image of a = C * Hs * Ws
};

constexpr int kDwpThreads = 256;
constexpr int kDwpMaxPass = 4;  // units per workgroup <= 4 * 256

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

// v / d for 0 <= v < 2^20, rcp = 1.0f / d: exact (the +0.5 keeps the product half a step away from every integer)
__device__ __forceinline__ int dwp_div(int v, float rcp) { return (int)(((float)v + 0.5f) * rcp); }

template <int DT> __device__ __forceinline__ float dwp_ld(const void* p, size_t i) {
  if constexpr (DT == SSDK_F32) return ((const float*)p)[i];
  else return bits16_to_f32<DT>(((const u16*)p)[i]);
}

// 8 consecutive elements from element index i (any alignment the element type has) -> fp32; the caller guarantees
// i + 8 <= elements of the tensor
template <int DT> struct DwpRaw { u32x4 q[DT == SSDK_F32 ? 2 : 1]; };
template <int DT> __device__ __forceinline__ DwpRaw<DT> dwp_load8(const void* src, size_t i) {
  DwpRaw<DT> r;
  if constexpr (DT == SSDK_F32) {
    r.q[0] = *reinterpret_cast<const u32x4_a4*>((const u32*)src + i);
    r.q[1] = *reinterpret_cast<const u32x4_a4*>((const u32*)src + i + 4);
  } else {
    r.q[0] = *reinterpret_cast<const u32x4_a2*>((const u16*)src + i);
  }
  return r;
}
template <int DT> __device__ __forceinline__ void dwp_unpack(const DwpRaw<DT>& r, float (&v)[8]) {
  if constexpr (DT == SSDK_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32 lo = r.q[0][e], hi = r.q[1][e];  // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0)
      v[e] = __builtin_bit_cast(float, lo);
      v[4 + e] = __builtin_bit_cast(float, hi);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bits16_to_f32<DT>(r.q[0][e] & 0xffffu);
      v[2 * e + 1] = bits16_to_f32<DT>(r.q[0][e] >> 16);
    }
  }
}
// 8 elements of one row starting at element i, of which the first `valid` exist in the row; elements past `valid` read
// as zero.  Reads 16 bytes whenever they lie inside the tensor (past the row end that is the next row), element by
// element at the very end of the tensor.
template <int DT> __device__ __forceinline__ void dwp_load_row8(const void* src, size_t i, int valid, size_t total, float (&v)[8]) {
  if (i + 8 <= total) {
    dwp_unpack<DT>(dwp_load8<DT>(src, i), v);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = i + e < total ? dwp_ld<DT>(src, i + e) : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = e < valid ? v[e] : 0.f;
}

template <int DT> __device__ __forceinline__ void dwp_store8(void* dst, size_t i, const float (&v)[8], int valid) {
  if constexpr (DT == SSDK_F32) {
    float* d = (float*)dst + i;
    if (valid >= 8) {
      *reinterpret_cast<f32x4_a4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4_a4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < valid) d[e] = v[e];
    }
  } else {
    u16* d = (u16*)dst + i;
    if (valid >= 8) {
      *reinterpret_cast<u32x4_a2*>(d) =
          u32x4{pack2_16<DT>(v[0], v[1]), pack2_16<DT>(v[2], v[3]), pack2_16<DT>(v[4], v[5]), pack2_16<DT>(v[6], v[7])};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < valid) d[e] = (u16)f32_to_bits16<DT>(v[e]);
    }
  }
}

// where a logical workgroup works: channel, first image, first row of its band
struct DwpWhere { int c, grp, n0, r0; bool live; };
__device__ __forceinline__ DwpWhere dwp_where(const DwpParams& p) {
  DwpWhere w;
  const int bid = (int)blockIdx.x;
  const int L = (bid & 7) * p.nwg8 + (bid >> 3);  // XCD x of 8 gets the logical run [x * nwg8, (x + 1) * nwg8)
  w.live = L < p.nwg;
  w.grp = L / p.C;
  w.c = L - w.grp * p.C;
  const int nb = w.grp / p.T;
  w.n0 = nb * p.G;
  w.r0 = (w.grp - nb * p.T) * p.TR;
  return w;
}

// stage the G pieces of the workgroup: LDS row i * SR + sr = row ys0 + sr of plane (n0 + i, c); the column left of pixel
// 0 and the columns past Ws are zero, rows outside the plane (or of images past N) are zero.
// fp32: one word per pixel, pixel x at column x + 1.
__device__ __forceinline__ void dwp_stage_f32(float* lds, const DwpParams& p, int n0, int c, int ys0) {
  const int items = p.G * p.SR * p.CH;
  const size_t plane = (size_t)p.Hs * p.Ws;
  for (int it0 = (int)threadIdx.x; it0 < items; it0 += 4 * kDwpThreads) {
    float v[4][8];
    int lr[4], x0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j * kDwpThreads;
      lr[j] = dwp_div(it, p.rcpCH);
      x0[j] = (it - lr[j] * p.CH) * 8;
      const int i = dwp_div(lr[j], p.rcpSR), sr = lr[j] - i * p.SR;
      const int ys = ys0 + sr;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
      if (it < items && n0 + i < p.N && (unsigned)ys < (unsigned)p.Hs && x0[j] < p.Ws)
        dwp_load_row8<SSDK_F32>(p.a, (size_t)(n0 + i) * p.img_a + c * plane + (size_t)(ys * p.Ws + x0[j]), p.Ws - x0[j], p.total_a, v[j]);
    }
    __builtin_amdgcn_sched_barrier(0);  // all four loads in flight before the first LDS write waits for one
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (it0 + j * kDwpThreads >= items) continue;
      float* d = lds + lr[j] * p.LD + x0[j] + 1;  // (LD - 1 is a multiple of 8: every chunk lies inside the row)
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = v[j][e];
      if (x0[j] == 0) d[-1] = 0.f;
    }
  }
}
// 16-bit: the tensor's own bits, pixel x at column x + 8, LD a multiple of 8: chunk k of a row = pixels 8k .. 8k + 7 =
// the aligned 16 bytes at column 8k + 8; columns 0 .. 7 (of which the windows read column 7 = pixel -1) are zero
// AFF (round 6): the staged tensor is the OUTPUT of a training-mode BatchNorm (+ ReLU6 / ReLU) that was never written: p.a is the
// BatchNorm's input x, and a chunk becomes act(x a + b) rounded to the dtype -- bit for bit what bn_apply_kernel would have
// stored -- before the padding masks (the padding is zero in the BatchNorm's output, not in its input).
template <int DT, bool AFF>
__device__ __forceinline__ void dwp_stage_h16(u16* lds, const DwpParams& p, int n0, int c, int ys0) {
  const int items = p.G * p.SR * p.CH;
  const size_t plane = (size_t)p.Hs * p.Ws;
  const u16* src = (const u16*)p.a;
  float ca = 1.f, cb = 0.f;
  if constexpr (AFF) {
    ca = p.coef[c * 4 + 0];
    cb = p.coef[c * 4 + 1];
  }
  for (int it0 = (int)threadIdx.x; it0 < items; it0 += 4 * kDwpThreads) {
    u32x4 v[4];
    int lr[4], x0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j * kDwpThreads;
      lr[j] = dwp_div(it, p.rcpCH);
      x0[j] = (it - lr[j] * p.CH) * 8;
      const int i = dwp_div(lr[j], p.rcpSR), sr = lr[j] - i * p.SR;
      const int ys = ys0 + sr;
      v[j] = u32x4{0u, 0u, 0u, 0u};
      if (it < items && n0 + i < p.N && (unsigned)ys < (unsigned)p.Hs && x0[j] < p.Ws) {
        const size_t gi = (size_t)(n0 + i) * p.img_a + c * plane + (size_t)(ys * p.Ws + x0[j]);
        if (gi + 8 <= p.total_a) {  // 16 bytes (past the row end: the next row, masked below)
          v[j] = *reinterpret_cast<const u32x4_a2*>(src + gi);
        } else {  // the last pixels of the tensor
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (gi + e < p.total_a) v[j][e >> 1] |= (u32)src[gi + e] << (16 * (e & 1));
        }
        if constexpr (AFF) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            float lo = bits16_to_f32<DT>(v[j][d] & 0xffffu) * ca + cb, hi = bits16_to_f32<DT>(v[j][d] >> 16) * ca + cb;
            if (p.act) {  // (the same operations, in the same order, as bn_apply_kernel's forward lambda)
              lo = fmaxf(lo, 0.f);
              hi = fmaxf(hi, 0.f);
            }
            if (p.act == 1) {
              lo = fminf(lo, 6.f);
              hi = fminf(hi, 6.f);
            }
            v[j][d] = pack2_16<DT>(lo, hi);
          }
        }
        const int valid = p.Ws - x0[j];  // pixels of this chunk inside the row
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int rem = valid - 2 * d;
          v[j][d] &= rem >= 2 ? 0xffffffffu : rem == 1 ? 0x0000ffffu : 0u;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // all four loads in flight before the first LDS write waits for one
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (it0 + j * kDwpThreads >= items) continue;
      u16* d = lds + lr[j] * p.LD + x0[j] + 8;
      *reinterpret_cast<u32x4*>(d) = v[j];
      if (x0[j] == 0) *reinterpret_cast<u32x4*>(d - 8) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}
template <int DT, bool AFF = false>
__device__ __forceinline__ void dwp_stage(float* lds, const DwpParams& p, int n0, int c, int ys0) {
  static_assert(!AFF || DT != SSDK_F32, "the deferred BatchNorm is a 16-bit path");
  if constexpr (DT == SSDK_F32) dwp_stage_f32(lds, p, n0, c, ys0);
  else dwp_stage_h16<DT, AFF>(reinterpret_cast<u16*>(lds), p, n0, c, ys0);
}

// one window row as fp32: a[j] = pixel x = first - 1 + j of staged row `row`, j = 0 .. NC-1, first = S * 8 * g
// (NC = 10 at stride 1, 17 at stride 2)
template <int DT, int S>
__device__ __forceinline__ void dwp_window(const float* lds, const DwpParams& p, int row, int g, float (&a)[7 * S + 3]) {
  constexpr int NC = 7 * S + 3;
  if constexpr (DT == SSDK_F32) {
    const float* rp = lds + row * p.LD + S * 8 * g;
#pragma unroll
    for (int j = 0; j < NC; ++j) a[j] = rp[j];
  } else {
    const u16* rp = reinterpret_cast<const u16*>(lds) + row * p.LD + S * 8 * g;  // column of pixel first - 8
    a[0] = bits16_to_f32<DT>(rp[7]);
#pragma unroll
    for (int h = 0; h < S; ++h) {
      const u32x4 m = *reinterpret_cast<const u32x4*>(rp + 8 + 8 * h);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32 wd = m[d];
        a[1 + 8 * h + 2 * d] = bits16_to_f32<DT>(wd & 0xffffu);
        a[2 + 8 * h + 2 * d] = bits16_to_f32<DT>(wd >> 16);
      }
    }
    if constexpr (S == 1) a[9] = bits16_to_f32<DT>(rp[16]);
  }
}

// unit u of the workgroup -> (piece i, row r of the band, segment g); false: nothing there
struct DwpUnit { int i, r, g; };
__device__ __forceinline__ bool dwp_unit(const DwpParams& p, const DwpWhere& w, int u, DwpUnit& q) {
  q.i = dwp_div(u, p.rcpUP);
  const int ru = u - q.i * p.UP;
  q.r = dwp_div(ru, p.rcpSEG);
  q.g = ru - q.r * p.seg;
  return u < p.G * p.UP && w.n0 + q.i < p.N && w.r0 + q.r < p.Ht;
}

// forward (FLIP = false) and the stride-1 input gradient (FLIP = true: the same window with the taps reversed, staged
// tensor = dy):  out[r][8g + e] = sum w[ky][kx] * a[S r + ky - 1][S (8g + e) + kx - 1]
// STATS (round 6, forward only): the workgroup owns ONE channel -- it also leaves (sum y, sum y^2) of its outputs for the
// BatchNorm behind the convolution (ssdk_bn_act_train_fwd_sums): per thread in registers, then a fixed-order block sum.
template <int DT, int S, bool FLIP, bool STATS = false, bool AFF = false>
__global__ __launch_bounds__(kDwpThreads) void dwp_fwd_kernel(const DwpParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sred[2][kDwpThreads / 64];
  const DwpWhere wh = dwp_where(p);
  if (!wh.live) return;
  float st1 = 0.f, st2 = 0.f;
  float w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = dwp_ld<DT>(p.b, (size_t)wh.c * 9 + (FLIP ? 8 - t : t));
  dwp_stage<DT, AFF>(lds, p, wh.n0, wh.c, S * wh.r0 - 1);
  __syncthreads();
  // packed fp32 math: the two halves of a segment (pixels e and e + 4) share an instruction -- one v_pk_fma_f32 updates
  // the pair of sums from the pair of window columns j and j + 4S
  constexpr int HP = 4 * S, NP = 3 * S + 3;
  for (int u = (int)threadIdx.x; u < p.G * p.UP; u += kDwpThreads) {
    DwpUnit q;
    if (!dwp_unit(p, wh, u, q)) continue;
    f32x2 acc2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc2[e] = f32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float af[7 * S + 3];
      dwp_window<DT, S>(lds, p, q.i * p.SR + S * q.r + ky, q.g, af);
      f32x2 a[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) a[j] = f32x2{af[j], af[j + HP]};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc2[e] = __builtin_elementwise_fma(f32x2{w[ky * 3 + kx], w[ky * 3 + kx]}, a[S * e + kx], acc2[e]);
    }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] = acc2[e].x;
      acc[4 + e] = acc2[e].y;
    }
    const int oy = wh.r0 + q.r, ox = q.g * 8;
    dwp_store8<DT>(p.out, ((size_t)(wh.n0 + q.i) * p.C + wh.c) * p.Ht * p.Wt + (size_t)oy * p.Wt + ox, acc, p.Wt - ox);
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = e < p.Wt - ox ? acc[e] : 0.f;
        st1 += v;
        st2 += v * v;
      }
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      st1 += __shfl_xor(st1, o, 64);
      st2 += __shfl_xor(st2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      sred[0][threadIdx.x >> 6] = st1;
      sred[1][threadIdx.x >> 6] = st2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f, b = 0.f;
      for (int w = 0; w < kDwpThreads / 64; ++w) {  // wave order: bit-reproducible
        a += sred[0][w];
        b += sred[1][w];
      }
      p.stats[((size_t)wh.grp * p.C + wh.c) * 2 + 0] = a;
      p.stats[((size_t)wh.grp * p.C + wh.c) * 2 + 1] = b;
    }
  }
}

// stride-2 input gradient: dx[iy][ix] = sum over the taps with (iy + 1 - ky), (ix + 1 - kx) even of
// w[ky][kx] * dy[(iy + 1 - ky) / 2][(ix + 1 - kx) / 2].  Row iy reads dy rows nh = (iy + 1) >> 1 (tap row 1 when iy is
// even, 0 when odd) and nh - 1 (tap row 2, odd iy only); pixel 8g + e reads column 4g + e/2 (tap 1) when e is even,
// columns 4g + (e+1)/2 (tap 0) and 4g + (e-1)/2 (tap 2) when odd.  Staged column = dy column + 1.
template <int DT>
__global__ __launch_bounds__(kDwpThreads) void dwp_dgrad2_kernel(const DwpParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwpWhere wh = dwp_where(p);
  if (!wh.live) return;
  float w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = dwp_ld<DT>(p.b, (size_t)wh.c * 9 + t);
  dwp_stage<DT>(lds, p, wh.n0, wh.c, (wh.r0 >> 1) - 1);  // r0 is even
  __syncthreads();
  for (int u = (int)threadIdx.x; u < p.G * p.UP; u += kDwpThreads) {
    DwpUnit q;
    if (!dwp_unit(p, wh, u, q)) continue;
    const int iy = wh.r0 + q.r;
    const int lrow = ((iy + 1) >> 1) - (wh.r0 >> 1) + 1;
    float hv[5], lv[5];  // dy columns 4g .. 4g + 4 of rows nh and nh - 1
    if constexpr (DT == SSDK_F32) {
      const float* hi = lds + (q.i * p.SR + lrow) * p.LD + 4 * q.g + 1;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        hv[j] = hi[j];
        lv[j] = hi[j - p.LD];
      }
    } else {
      const u16* hi = reinterpret_cast<const u16*>(lds) + (q.i * p.SR + lrow) * p.LD + 4 * q.g + 8;
      const uint2 mh = *reinterpret_cast<const uint2*>(hi), ml = *reinterpret_cast<const uint2*>(hi - p.LD);
      hv[0] = bits16_to_f32<DT>(mh.x & 0xffffu);
      hv[1] = bits16_to_f32<DT>(mh.x >> 16);
      hv[2] = bits16_to_f32<DT>(mh.y & 0xffffu);
      hv[3] = bits16_to_f32<DT>(mh.y >> 16);
      hv[4] = bits16_to_f32<DT>(hi[4]);
      lv[0] = bits16_to_f32<DT>(ml.x & 0xffffu);
      lv[1] = bits16_to_f32<DT>(ml.x >> 16);
      lv[2] = bits16_to_f32<DT>(ml.y & 0xffffu);
      lv[3] = bits16_to_f32<DT>(ml.y >> 16);
      lv[4] = bits16_to_f32<DT>(hi[4 - p.LD]);
    }
    const bool odd = iy & 1;
    float whi[3], wlo[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      whi[kx] = odd ? w[kx] : w[3 + kx];
      wlo[kx] = odd ? w[6 + kx] : 0.f;
    }
    f32x2 ah[3], al[3];  // pixel pairs (e, e + 4): dy columns j and j + 2
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ah[j] = f32x2{hv[j], hv[j + 2]};
      al[j] = f32x2{lv[j], lv[j + 2]};
    }
    auto bc = [](float v) { return f32x2{v, v}; };
    float acc[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2 v;
      if (e & 1) {
        v = bc(wlo[2]) * al[(e - 1) / 2];
        v = __builtin_elementwise_fma(bc(wlo[0]), al[(e + 1) / 2], v);
        v = __builtin_elementwise_fma(bc(whi[2]), ah[(e - 1) / 2], v);
        v = __builtin_elementwise_fma(bc(whi[0]), ah[(e + 1) / 2], v);
      } else {
        v = bc(wlo[1]) * al[e / 2];
        v = __builtin_elementwise_fma(bc(whi[1]), ah[e / 2], v);
      }
      acc[e] = v.x;
      acc[4 + e] = v.y;
    }
    const int ix = q.g * 8;
    dwp_store8<DT>(p.out, ((size_t)(wh.n0 + q.i) * p.C + wh.c) * p.Ht * p.Wt + (size_t)iy * p.Wt + ix, acc, p.Wt - ix);
  }
}

// weight gradient, stage 1: partial[grp][c][t] = sum over the workgroup's units of dy[r][8g + e] * x[S r + ky - 1][S (8g + e) + kx - 1]
template <int DT, int S, bool AFF = false>
__global__ __launch_bounds__(kDwpThreads) void dwp_wgrad_kernel(const DwpParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4][9];
  const DwpWhere wh = dwp_where(p);
  if (!wh.live) return;
  // the dy segments of all the units of this thread first: they do not depend on the staged tile
  DwpRaw<DT> graw[kDwpMaxPass];
  const size_t plane_t = (size_t)p.Ht * p.Wt;
#pragma unroll
  for (int ps = 0; ps < kDwpMaxPass; ++ps) {
    DwpUnit q;
    const int u = (int)threadIdx.x + ps * kDwpThreads;
#pragma unroll
    for (int k = 0; k < (DT == SSDK_F32 ? 2 : 1); ++k) graw[ps].q[k] = u32x4{0u, 0u, 0u, 0u};
    if (dwp_unit(p, wh, u, q)) {
      const size_t gi = ((size_t)(wh.n0 + q.i) * p.C + wh.c) * plane_t + (size_t)(wh.r0 + q.r) * p.Wt + q.g * 8;
      if (gi + 8 <= p.total_b) {
        graw[ps] = dwp_load8<DT>(p.b, gi);
      } else {  // the last elements of the tensor: pack them one by one (fp32: as is; 16-bit: two per word)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (gi + e >= p.total_b) continue;
          if constexpr (DT == SSDK_F32) graw[ps].q[e >> 2][e & 3] = ((const u32*)p.b)[gi + e];
          else graw[ps].q[0][e >> 1] |= (u32)((const u16*)p.b)[gi + e] << (16 * (e & 1));
        }
      }
    }
  }
  dwp_stage<DT, AFF>(lds, p, wh.n0, wh.c, S * wh.r0 - 1);
  __syncthreads();
  constexpr int HP = 4 * S, NP = 3 * S + 3;  // packed fp32 pairs (pixels e, e + 4) as in the forward kernel
  f32x2 acc2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc2[t] = f32x2{0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < kDwpMaxPass; ++ps) {
    DwpUnit q;
    const int u = (int)threadIdx.x + ps * kDwpThreads;
    if (!dwp_unit(p, wh, u, q)) continue;
    float gv[8];
    dwp_unpack<DT>(graw[ps], gv);
    const int valid = p.Wt - q.g * 8;
    f32x2 g2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g2[e] = f32x2{e < valid ? gv[e] : 0.f, e + 4 < valid ? gv[e + 4] : 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float af[7 * S + 3];
      dwp_window<DT, S>(lds, p, q.i * p.SR + S * q.r + ky, q.g, af);
      f32x2 a[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) a[j] = f32x2{af[j], af[j + HP]};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc2[ky * 3 + kx] = __builtin_elementwise_fma(g2[e], a[S * e + kx], acc2[ky * 3 + kx]);
    }
  }
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = acc2[t].x + acc2[t].y;
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = acc[t];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);  // fixed butterfly: the same order every run
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    ((float*)p.out)[((size_t)wh.grp * p.C + wh.c) * 9 + threadIdx.x] = v;
  }
}

// weight gradient, stage 2: dw[c][t] = sum over the groups, lane l adds groups l, l + 64, ... in index order, then a
// fixed butterfly
__global__ __launch_bounds__(64) void dwp_wgrad_reduce_kernel(const float* partial, float* dw, int groups, int C) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int j = lane; j < groups; j += 64) {
    const float* q = partial + ((size_t)j * C + c) * 9;
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

// ---- host: how a pass is cut into workgroups -------------------------------------------------------------------------
enum { DWP_FWD = 0, DWP_DGRAD = 1, DWP_WGRAD = 2 };
constexpr int kDwpUnits = kDwpMaxPass * kDwpThreads;  // unit budget of a workgroup
constexpr int kDwpLdsBytes = 32768;                   // LDS budget of the staged pieces: five workgroups per CU

struct DwpPlan {
  DwpParams p;
  size_t lds;
  int groups;  // partial-sum groups of the weight gradient
  bool ok;
};

static DwpPlan dwp_plan(int kind, int N, int C, int H, int W, int stride, bool h16) {  // h16: the 16-bit LDS layout
  DwpPlan pl;
  DwpParams& p = pl.p;
  pl.ok = false;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const bool dgrad = kind == DWP_DGRAD, dgrad2 = dgrad && stride == 2;
  const int S = dgrad ? 1 : stride;  // window stride of the fwd-shaped kernels
  p.N = N;
  p.C = C;
  p.Ht = dgrad ? H : Ho;
  p.Wt = dgrad ? W : Wo;
  p.Hs = dgrad ? Ho : H;
  p.Ws = dgrad ? Wo : W;
  p.seg = (p.Wt + 7) / 8;
  if (h16) {  // pixel x at column x + 8; the highest column a thread reads, + 1; a multiple of 8
    int need = dgrad2 ? 4 * p.seg + 9 : S == 1 ? 8 * p.seg + 9 : 16 * p.seg + 8;
    if (need < p.Ws + 9) need = p.Ws + 9;
    p.LD = (need + 7) & ~7;
    p.CH = (p.LD - 8) / 8;
  } else {  // pixel x at column x + 1
    int need = dgrad2 ? 4 * p.seg + 2 : S * (8 * p.seg - 1) + 3;
    if (need < p.Ws + 2) need = p.Ws + 2;
    p.LD = ((need + 7) & ~7) + 1;  // = 1 mod 8: the rows of a wave start in different banks
    p.CH = (p.LD - 1) / 8;
  }
  const int lds_elems = kDwpLdsBytes / (h16 ? 2 : 4);
  const int rows_lds = lds_elems / p.LD;  // staged rows that fit
  // rows of the thread space whose staged rows fit / whose units fit
  const int tr_lds = dgrad2 ? 2 * (rows_lds - 2) : (rows_lds - 3) / S + 1;
  constexpr int env_units = 0;  // (round 6: the SSDK_DW_UNITS switch is gone, its A/B is settled)  // (tuning: units per workgroup)
  const int units = env_units >= 64 && env_units <= kDwpUnits ? env_units : kDwpUnits;
  int tr = units / p.seg;
  if (tr > tr_lds) tr = tr_lds;
  if (tr > p.Ht) tr = p.Ht;
  if (tr < p.Ht && dgrad2) tr &= ~1;
  // (32-bit offsets inside a plane, 32-bit workgroup counts)
  if (tr < 1 || rows_lds < 3 || (long)N * C * p.Ht > 1000000000l || (long)H * W >= (1l << 31)) return pl;
  auto staged_rows = [&](int t) { return dgrad2 ? (t >> 1) + 2 : S * (t - 1) + 3; };
  if (tr >= p.Ht) {  // whole planes: as many images per workgroup as fit
    p.T = 1;
    p.TR = p.Ht;
    p.SR = staged_rows(p.TR);
    // forward / input gradient: half the unit budget -- more, shorter workgroups (measured on the 19 x 19 and 10 x 10
    // maps at batch 64: 18.6 -> 16.2 and 20.0 -> 16.1 us); the weight gradient keeps its registers busy with the full one
    int g = (kind == DWP_WGRAD || env_units ? units : units / 2) / (p.TR * p.seg);
    if (g > rows_lds / p.SR) g = rows_lds / p.SR;
    if (g > N) g = N;
    p.G = g < 1 ? 1 : g;
  } else {
    p.G = 1;
    int t = (p.Ht + tr - 1) / tr;
    p.TR = (p.Ht + t - 1) / t;
    if (dgrad2) p.TR = (p.TR + 1) & ~1;
    p.T = (p.Ht + p.TR - 1) / p.TR;
    p.SR = staged_rows(p.TR);
  }
  if (p.G * p.SR * p.LD > lds_elems || p.G * p.TR * p.seg > kDwpUnits) return pl;
  p.UP = p.TR * p.seg;
  pl.groups = ((N + p.G - 1) / p.G) * p.T;
  const long nwg = (long)pl.groups * C;
  if (nwg > 2000000000l) return pl;
  p.nwg = (int)nwg;
  p.nwg8 = (p.nwg + 7) / 8;
  p.rcpCH = 1.0f / (float)p.CH;
  p.rcpSR = 1.0f / (float)p.SR;
  p.rcpUP = 1.0f / (float)p.UP;
  p.rcpSEG = 1.0f / (float)p.seg;
  p.total_a = (size_t)N * C * p.Hs * p.Ws;
  p.total_b = (size_t)N * C * p.Ht * p.Wt;
  p.img_a = (size_t)C * p.Hs * p.Ws;
  pl.lds = (size_t)p.G * p.SR * p.LD * (h16 ? 2 : 4);
  pl.ok = true;
  return pl;
}

static bool dwp_enabled() {
  static const int env = getenv("SSDK_DW_PLANE") ? atoi(getenv("SSDK_DW_PLANE")) : 1;
  return env != 0;
}

#define SSDK_DWP_BY_DTYPE(CALL)             \
  do {                                      \
    if (dtype == SSDK_F32) CALL(SSDK_F32);  \
    else if (dtype == SSDK_BF16) CALL(SSDK_BF16); \
    else CALL(SSDK_F16);                    \
  } while (0)

// 0: launched; 1: not taken (the caller uses the tiled kernels of ssdk_dwtrain.hip)
int launch_dwp_fwd(const void* x, const void* w, void* y, int N, int C, int H, int W, int stride, int dtype, hipStream_t stream,
                   float* stats, int* groups_out, const float* coef, int act) {
  if (!dwp_enabled()) return 1;
  DwpPlan pl = dwp_plan(DWP_FWD, N, C, H, W, stride, dtype != SSDK_F32);
  if (!pl.ok) return 1;
  pl.p.a = x;
  pl.p.b = w;
  pl.p.out = y;
  pl.p.stats = stats;
  pl.p.coef = coef;
  pl.p.act = act;
  if (groups_out) *groups_out = pl.groups;
  const dim3 grid((unsigned)(pl.p.nwg8 * 8));
  if (coef) {  // the staged tensor is a deferred BatchNorm's output (16 bit only; statistics on or off)
    if (dtype == SSDK_F32) return 1;
#define SSDK_DWP_FWA(DT)                                                                                                     \
  do {                                                                                                                       \
    if (stats) {                                                                                                             \
      if (stride == 1) hipLaunchKernelGGL((dwp_fwd_kernel<DT, 1, false, true, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
      else hipLaunchKernelGGL((dwp_fwd_kernel<DT, 2, false, true, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);     \
    } else if (stride == 1) hipLaunchKernelGGL((dwp_fwd_kernel<DT, 1, false, false, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
    else hipLaunchKernelGGL((dwp_fwd_kernel<DT, 2, false, false, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);      \
  } while (0)
    if (dtype == SSDK_BF16) SSDK_DWP_FWA(SSDK_BF16);
    else SSDK_DWP_FWA(SSDK_F16);
#undef SSDK_DWP_FWA
    return 0;
  }
#define SSDK_DWP_FWD(DT)                                                                                                 \
  do {                                                                                                                   \
    if (stats) {                                                                                                         \
      if (stride == 1) hipLaunchKernelGGL((dwp_fwd_kernel<DT, 1, false, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
      else hipLaunchKernelGGL((dwp_fwd_kernel<DT, 2, false, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);       \
    } else if (stride == 1) hipLaunchKernelGGL((dwp_fwd_kernel<DT, 1, false>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
    else hipLaunchKernelGGL((dwp_fwd_kernel<DT, 2, false>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);               \
  } while (0)
  SSDK_DWP_BY_DTYPE(SSDK_DWP_FWD);
#undef SSDK_DWP_FWD
  return 0;
}

// 1 when the forward AND the weight-gradient pass of this geometry run on the whole-row kernels (what a deferred BatchNorm needs)
int dwp_affine_ok(int N, int C, int H, int W, int stride, int dtype) {
  if (!dwp_enabled() || dtype == SSDK_F32) return 0;
  return dwp_plan(DWP_FWD, N, C, H, W, stride, true).ok && dwp_plan(DWP_WGRAD, N, C, H, W, stride, true).ok ? 1 : 0;
}

// partial-sum groups of the forward statistics (0: the whole-row kernels do not take the pass)
int dwp_fwd_groups(int N, int C, int H, int W, int stride, int dtype) {
  if (!dwp_enabled()) return 0;
  const DwpPlan pl = dwp_plan(DWP_FWD, N, C, H, W, stride, dtype != SSDK_F32);
  return pl.ok ? pl.groups : 0;
}

int launch_dwp_dgrad(const void* dy, const void* w, void* dx, int N, int C, int H, int W, int stride, int dtype, hipStream_t stream) {
  if (!dwp_enabled()) return 1;
  DwpPlan pl = dwp_plan(DWP_DGRAD, N, C, H, W, stride, dtype != SSDK_F32);
  if (!pl.ok) return 1;
  pl.p.a = dy;
  pl.p.b = w;
  pl.p.out = dx;
  const dim3 grid((unsigned)(pl.p.nwg8 * 8));
#define SSDK_DWP_DG(DT)                                                                                                  \
  do {                                                                                                                   \
    if (stride == 1) hipLaunchKernelGGL((dwp_fwd_kernel<DT, 1, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);    \
    else hipLaunchKernelGGL((dwp_dgrad2_kernel<DT>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);                      \
  } while (0)
  SSDK_DWP_BY_DTYPE(SSDK_DWP_DG);
#undef SSDK_DWP_DG
  return 0;
}

// bytes of the partial sums (0: the pass is not taken)
size_t dwp_wgrad_workspace_bytes(int N, int C, int H, int W, int stride) {
  size_t need = 0;  // (the entry point has no dtype: whichever LDS layout wants more)
  for (int h16 = 0; h16 < 2; ++h16) {
    const DwpPlan pl = dwp_plan(DWP_WGRAD, N, C, H, W, stride, h16 != 0);
    const size_t b = pl.ok ? (size_t)pl.groups * C * 9 * sizeof(float) : 0;
    if (b > need) need = b;
  }
  return need;
}

int launch_dwp_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int C, int H, int W,
                     int stride, int dtype, hipStream_t stream, const float* coef, int act) {
  if (!dwp_enabled()) return 1;
  DwpPlan pl = dwp_plan(DWP_WGRAD, N, C, H, W, stride, dtype != SSDK_F32);
  if (!pl.ok || workspace_bytes < (size_t)pl.groups * C * 9 * sizeof(float)) return 1;
  pl.p.a = x;
  pl.p.b = dy;
  pl.p.out = workspace;
  pl.p.coef = coef;
  pl.p.act = act;
  const dim3 grid((unsigned)(pl.p.nwg8 * 8));
  if (coef) {
    if (dtype == SSDK_F32) return 1;
#define SSDK_DWP_WGA(DT)                                                                                               \
  do {                                                                                                                 \
    if (stride == 1) hipLaunchKernelGGL((dwp_wgrad_kernel<DT, 1, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
    else hipLaunchKernelGGL((dwp_wgrad_kernel<DT, 2, true>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);            \
  } while (0)
    if (dtype == SSDK_BF16) SSDK_DWP_WGA(SSDK_BF16);
    else SSDK_DWP_WGA(SSDK_F16);
#undef SSDK_DWP_WGA
    hipLaunchKernelGGL(dwp_wgrad_reduce_kernel, dim3((unsigned)C), dim3(64), 0, stream, (const float*)workspace, dw, pl.groups, C);
    return 0;
  }
#define SSDK_DWP_WG(DT)                                                                                       \
  do {                                                                                                        \
    if (stride == 1) hipLaunchKernelGGL((dwp_wgrad_kernel<DT, 1>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p); \
    else hipLaunchKernelGGL((dwp_wgrad_kernel<DT, 2>), grid, dim3(kDwpThreads), pl.lds, stream, pl.p);        \
  } while (0)
  SSDK_DWP_BY_DTYPE(SSDK_DWP_WG);
#undef SSDK_DWP_WG
  hipLaunchKernelGGL(dwp_wgrad_reduce_kernel, dim3((unsigned)C), dim3(64), 0, stream, (const float*)workspace, dw, pl.groups, C);
  return 0;
}

}  // namespace ssdk

// How the whole-row kernels would cut one pass (host only: no launch, no device needed).  pass 0 forward, 1 input
// gradient, 2 weight gradient.  out[12] = G, T, TR, seg, LD, SR, CH, UP, workgroups, groups (partial sums of the weight
// gradient), LDS bytes, rows of the thread space.  Returns 0, or 1 when the tiled kernels would take the pass.
extern "C" int ssdk_dwconv_plan(int pass, int N, int C, int H, int W, int stride, int dtype, int* out) {
  if (!out || pass < 0 || pass > 2 || N < 1 || C < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2) ||
      (dtype != SSDK_F32 && dtype != SSDK_BF16 && dtype != SSDK_F16))
    return SSDK_E_BADARG;
  const ssdk::DwpPlan pl = ssdk::dwp_plan(pass, N, C, H, W, stride, dtype != SSDK_F32);
  if (!pl.ok) return 1;
  const ssdk::DwpParams& p = pl.p;
  const int v[12] = {p.G, p.T, p.TR, p.seg, p.LD, p.SR, p.CH, p.UP, p.nwg, pl.groups, (int)pl.lds, p.Ht};
  for (int i = 0; i < 12; ++i) out[i] = v[i];
  return 0;
}

