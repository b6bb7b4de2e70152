// ssdk_tail.hip -- everything of Decoder.__call__ (reference ssds/modeling/layers/decoder.py:25-49) behind the scan, as TWO
// launches that never leave the device:
//
//   levelsel_kernel   one workgroup of 4 waves per (image, LEVEL) -- B x L workgroups: every CU of the chip works.
//     B1. selects the top K of the level from the <= K unordered keys each scan unit left in the workspace (box.py:446
//         topk over the whole level) and puts them in order -- by COUNTING, not by sorting or merging: a 1024-bin histogram
//         of the score (the same window the scan uses: one bin per bf16 value between the threshold and 1), the bin in
//         which the count from the top reaches K, a counting-sort scatter of the bins above it (their start offsets are
//         the histogram's suffix sums), a rank count inside the boundary bin and inside every bin that holds more than
//         one key.  With tie-free scores a bin holds one to three keys, so the "sort" is one LDS atomic per key;
//     B2. decodes the winners: gather of the 4 deltas, delta2box (box.py:74-87), centre rescoring (box.py:464-471), and
//         writes (score, box, class) at position l*K + r of the image's [L*K] arrays -- the slot torch.cat gives the
//         r-th output of level l (decoder.py:48), zero-padded like box.py:430-432.
//   nmswalk_kernel    one workgroup of 16 waves per image:
//     C.  orders the walk of box.nms lazily (box.py:505): the candidates with score > 0 (box.py:496) of the highest
//         score bins that together hold >= 256 keys, in order by the same counting sort; more rounds only while fewer
//         than `ndetections` boxes survived;
//     D.  walks them 64 at a time like nms_kernel (ssdk_nms.hip): kept-list test split over the 16 waves, suppression
//         rows of 4 pivots per wave, in-order resolve on bitmasks by wave 0 (box.py:512-544);
//     E.  writes the zero-padded [ndetections] outputs (box.py:489-491).
//
// History.  Round 2 ran B-E as ONE workgroup of 1024 threads per image (64 workgroups on 256 CUs) that merged SORTED unit
// lists by binary-search ranks and sorted all L*K candidates as 128-key register runs: 83 k cycles.  Round 3 first removed
// every sort (the scan emits unordered winners, ssdk_scan16.hip) inside the same one-workgroup shape: 63 k cycles, of which
// the stamps showed ~25 phases of 16-wave barriers and a VALU-bound decode of 1800 boxes on one CU.  Splitting at the
// [L*K] arrays -- which the reference materialises anyway and `mid_*` callers read -- puts the select + decode on all CUs with
// 4-wave barriers and leaves the sequential walk alone on its CU.
// Arithmetic and tie order are unchanged: same helpers, same fp32 sequences, keys are unique (score bits | ~index), so
// "top K" and "rank" are well defined.
#include "ssdk_common.h"
#include "ssdk_select.h"
#include "ssdk_decode.h"

namespace ssdk {

constexpr int kSelThreads = 256;   // levelsel_kernel
constexpr int kWalkThreads = 1024; // nmswalk_kernel
constexpr u32 kRound = 256;        // candidates ordered + walked per NMS round (target; a round takes whole score bins)
constexpr u32 kRoundMax = 512;     // ... and at most this many
constexpr u32 kCBins = 2048;       // score bins of the walk order (C): one per bf16 value from 2^-11 up to 32
constexpr u32 kCBase = 0xba00u;    // ord(score) >> 16 of 2^-11
constexpr u32 kTailBins = 1024;    // score bins per level (B1)
constexpr u32 kSelReg = 10;        // keys per thread of a level kept in registers (levels of <= 2560 key slots)
constexpr u32 kSelUnits = 64;      // units of a level whose minima give the level's lower bound (more: no bound)

struct TailLevel {
  const void* box;
  int A, C, H, W, stride;
  u32 units, unit_base, pad;
  u32 mW, mH, mC, pad2;  // floor(2^32 / d): quotient by multiply-high + one correction (tail_divmod)
  float anchors[SSDK_MAX_ANCHORS * 4];
};
struct SelParams {
  TailLevel lv[SSDK_MAX_LEVELS];
  int L, dtype, rescore;
  u32 units_per_image, K;
  u32 hist_base, hist_sh;  // score window of B1: bin = (ord(score) - hist_base) >> hist_sh, clamped (ssdk_decode.hip)
  const u64* cand;
  const u32* cand_cnt;
  float* mid_scores;  // [B, L*K], [B, L*K, 4], [B, L*K]
  float* mid_boxes;
  float* mid_classes;
  unsigned long long* stamps;  // optional (debug): shader-clock stamps of workgroup 0 at the phase boundaries
};
struct WalkParams {
  u32 N;  // L*K candidates per image
  u32 M;  // power of two >= N (>= 128): length of the NMS key array
  const float* scores;
  const float* boxes;
  const float* classes;
  float thr;
  int ndet, diou;
  float* out_scores;
  float* out_boxes;
  float* out_classes;
  unsigned long long* stamps;
};

// shared with level_kernel / nms_kernel (ssdk_decode.hip / ssdk_nms.hip); restated here because those live in other
// translation units as static inline device code
// q = n / d, r = n % d with m = floor(2^32 / d) (d >= 1; d = 1 is passed as m = 0xffffffff): the multiply-high
// under-estimates the quotient by at most one
__device__ __forceinline__ u32 tail_divmod(u32 n, u32 d, u32 m, u32* r) {
  u32 q = __umulhi(n, m);
  u32 rem = n - q * d;
  if (rem >= d) {
    ++q;
    rem -= d;
  }
  *r = rem;
  return q;
}

struct TailGather {  // a winner's position inside its level and its four raw deltas (loaded, not yet used)
  float d0, d1, d2, d3;
  u32 x, y, c, a;
};

// DT is a template parameter on purpose: with a run-time dtype every load sat in its own branch and the compiler waited for
// it there (s_waitcnt vmcnt(0) behind each of the 8 loads of a thread's two winners: 7.5 k of the 11 k cycles of B2)
template <int DT>
__device__ __forceinline__ float tail_load(const void* p, size_t i) {
  if constexpr (DT == SSDK_F32) return ((const float*)p)[i];
  else if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(((const u16*)p)[i]);
  else return f16_bits_to_f32(((const u16*)p)[i]);
}

template <int DT>
__device__ __forceinline__ TailGather tail_gather(const TailLevel& d, u32 b, u32 idx) {
  const u32 W = d.W, H = d.H, C = d.C;
  TailGather g;
  const u32 t1 = tail_divmod(idx, W, d.mW, &g.x);   // x = idx % W
  const u32 t2 = tail_divmod(t1, H, d.mH, &g.y);    // y = (idx / W) % H
  g.a = tail_divmod(t2, C, d.mC, &g.c);             // c = (idx / W / H) % C (box.py:448), a = idx / C / H / W (box.py:454)
  const size_t hw = (size_t)H * W;
  const size_t boff = ((size_t)b * d.A * 4 + (size_t)g.a * 4) * hw + (size_t)g.y * W + g.x;
  g.d0 = tail_load<DT>(d.box, boff);
  g.d1 = tail_load<DT>(d.box, boff + hw);
  g.d2 = tail_load<DT>(d.box, boff + 2 * hw);
  g.d3 = tail_load<DT>(d.box, boff + 3 * hw);
  return g;
}

// box.py:74-87 delta2box + box.py:459-471 for one winner; fp32, reference operation order (as level_kernel)
__device__ __forceinline__ void tail_decode(const TailLevel& d, const TailGather& g, int rescore, float score, float* o_score,
                                            float4* o_box, float* o_cls) {
  const u32 W = d.W, H = d.H;
  const float d0 = g.d0, d1 = g.d1, d2 = g.d2, d3 = g.d3;
  const u32 x = g.x, y = g.y, a = g.a, c = g.c;
  const float fs = (float)d.stride;
  const float g0 = (float)x * fs + d.anchors[a * 4 + 0];  // box.py:459-462
  const float g1 = (float)y * fs + d.anchors[a * 4 + 1];
  const float g2 = (float)x * fs + d.anchors[a * 4 + 2];
  const float g3 = (float)y * fs + d.anchors[a * 4 + 3];
  const float aw = g2 - g0 + 1.0f, ah = g3 - g1 + 1.0f;  // box.py:77
  const float cx = g0 + 0.5f * aw, cy = g1 + 0.5f * ah;  // box.py:78
  const float pcx = d0 * aw + cx, pcy = d1 * ah + cy;    // box.py:79
  const float pw = expf(d2) * aw;                       // box.py:80
  const float ph = expf(d3) * ah;
  const float Mx = (float)W * fs - 1.0f, My = (float)H * fs - 1.0f;  // box.py:83
  // torch.max(m, torch.min(t, M)) propagates a NaN t (box.py:84-87); v_min / v_max drop it, so it is put back explicitly
  // (two instructions instead of the NaN-masking min / max of ssdk_common.h, 14 of which made this function VALU-bound)
  auto clamp = [](float t, float M) {
    const float r = __builtin_fmaxf(0.0f, __builtin_fminf(t, M));
    return t != t ? t : r;
  };
  const float x1 = clamp(pcx - 0.5f * pw, Mx);
  const float y1 = clamp(pcy - 0.5f * ph, My);
  const float x2 = clamp(pcx + 0.5f * pw - 1.0f, Mx);
  const float y2 = clamp(pcy + 0.5f * ph - 1.0f, My);
  float s = score;
  if (rescore) {  // box.py:464-471
    const float gcx = (g0 + g2) / 2.0f, gcy = (g1 + g3) / 2.0f;
    const float ltx = fabsf(gcx - x1), lty = fabsf(gcy - y1);
    const float rbx = fabsf(x2 - gcx), rby = fabsf(y2 - gcy);
    // torch.min / torch.max of a NaN operand is NaN -> the score is NaN; the operands are NaN exactly when a corner is
    const bool bad = (x1 != x1) | (y1 != y1) | (x2 != x2) | (y2 != y2);
    const float rx = __builtin_fminf(ltx, rbx) / __builtin_fmaxf(ltx, rbx);
    const float ry = __builtin_fminf(lty, rby) / __builtin_fmaxf(lty, rby);
    s = s * sqrtf(rx * ry);
    s = bad ? __builtin_nanf("") : s;
  }
  *o_score = s;
  *o_box = make_float4(x1, y1, x2, y2);
  *o_cls = (float)c;
}

// FAST: no box of the image holds a NaN (the usual case, known per image: TailLds.hasnan) -- v_min / v_max are then exactly
// torch.min / torch.max; otherwise the NaN-propagating forms of ssdk_common.h (8 of them, ~9 instructions each)
template <bool FAST>
__device__ __forceinline__ bool tail_suppressed_by(const float4 b, float ab, const float4 p, float ap, float thr,
                                                   int diou) {  // box.py:518-533, as nms_kernel
  auto mx = [](float a, float c) { return FAST ? __builtin_fmaxf(a, c) : tmax(a, c); };
  auto mn = [](float a, float c) { return FAST ? __builtin_fminf(a, c) : tmin(a, c); };
  const float ix1 = mx(b.x, p.x), iy1 = mx(b.y, p.y);
  const float ix2 = mn(b.z, p.z), iy2 = mn(b.w, p.w);
  float w = ix2 - ix1 + 1.0f, h = iy2 - iy1 + 1.0f;
  w = mx(w, 0.0f);
  h = mx(h, 0.0f);
  const float inter = w * h;
  const float iou = inter / (ab + ap - inter + 1e-7f);
  bool over = !(iou <= thr);
  if (over && diou) {
    const float ox1 = mn(b.x, p.x), oy1 = mn(b.y, p.y);
    const float ox2 = mx(b.z, p.z), oy2 = mx(b.w, p.w);
    const float dx = b.x - p.x, dy = b.y - p.y;
    const float inter_diag = dx * dx + dy * dy;
    const float ow = ox2 - ox1, oh = oy2 - oy1;
    const float outer_diag = (ow * ow + oh * oh) + 1e-7f;
    float v = iou - inter_diag / outer_diag;
    v = (v < -1.0f) ? -1.0f : ((v > 1.0f) ? 1.0f : v);
    over = !(v <= thr);
  }
  return over;
}

// COH (tail2_kernel): the [L*K] arrays travel between workgroups of ONE launch, possibly on different XCDs (one L2 each).
// Agent-scope relaxed atomics are the cheap way: a store goes through to memory (sc1), a load does not take a line from the
// reader's L2 -- no cache-wide write-back / invalidate (measured: with an agent-scope release fence per workgroup, i.e. 384
// buffer_wbl2 of L2s full of the head convolutions' dirty lines, the walk's first 1800 loads took 16 us).
template <bool COH> __device__ __forceinline__ void tail_st(float* p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH> __device__ __forceinline__ void tail_st4(float4* p, const float4& v) {
  if constexpr (COH) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    const unsigned long long lo = (unsigned long long)__builtin_bit_cast(u32, v.x) | ((unsigned long long)__builtin_bit_cast(u32, v.y) << 32);
    const unsigned long long hi = (unsigned long long)__builtin_bit_cast(u32, v.z) | ((unsigned long long)__builtin_bit_cast(u32, v.w) << 32);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
template <bool COH> __device__ __forceinline__ float tail_ld(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH> __device__ __forceinline__ float4 tail_ld4(const float4* p) {
  if constexpr (COH) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__builtin_bit_cast(float, (u32)lo), __builtin_bit_cast(float, (u32)(lo >> 32)),
                       __builtin_bit_cast(float, (u32)hi), __builtin_bit_cast(float, (u32)(hi >> 32)));
  } else {
    return *p;
  }
}

__device__ __forceinline__ float tail_bcast(float v, u32 j) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)j));
}

__device__ __forceinline__ u32 tail_bin(u64 key, u32 base, u32 sh) {
  const u32 o = (u32)(key >> 32);
  const u32 bin = o < base ? 0u : (o - base) >> sh;  // (every candidate is at or above the threshold, i.e. >= base)
  return bin < kTailBins - 1 ? bin : kTailBins - 1;
}


// ------------------------------------------------------------------------------------------------------------------------
// levelsel_kernel
// ------------------------------------------------------------------------------------------------------------------------
struct alignas(16) SelLds {
  TailLevel lv;             // this workgroup's level (a copy: per-lane indexed anchors)
  u32 hist[kTailBins];      // score bins; after the cut: next free slot of every bin's group
  u16 gstart[kTailBins];    // first slot of a bin's group
  u64 umin[kSelUnits];      // smallest key of every unit of the level
  u32 ufull[kSelUnits];     // ... and whether its list is full (K keys)
  u64 lb;                   // lower bound of the level's K-th key (largest minimum of its FULL unit lists), when needed
  int cutbin;               // -1: the level offers <= K keys, every one is a winner
  u32 ln, above, nw, bcnt, generic, smax, pad1;  // smax: largest score (ordered bits) of the level's keys (ties fast path)
  SelScratch ss;
};
__host__ __device__ inline size_t sel_lds_bytes(u32 K) {
  return sizeof(SelLds) + 2 * ((((size_t)K * 8) + 15) & ~(size_t)15);  // wkeys, bkeys
}

// NT: 256 threads, or 1024 where a level has many scan units (the 112x112 / 80x80 levels of the FPN / BiFPN heads: 55 / 28
// units x K keys per image -- with 256 threads the two passes over the keys were 80 k of the workgroup's 94 k cycles)
template <int DT, int NT, bool COH = false>
__device__ __forceinline__ void levelsel_body(const SelParams& p, const u32 l, const u32 b, unsigned char* smem) {
  SelLds* S = reinterpret_cast<SelLds*>(smem);
  const u32 K = p.K;
  u64* wkeys = reinterpret_cast<u64*>(smem + sizeof(SelLds));  // [K] winners grouped by bin (descending), boundary winners in order
  u64* bkeys = wkeys + ((K + 1u) & ~1u);                       // [K] boundary list
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 L = (u32)p.L;
  const bool stamp = p.stamps != nullptr && l == 0 && b == 0 && tid == 0;
  if (stamp) p.stamps[0] = clock64();
  const u32 hbase = p.hist_base, hsh = p.hist_sh;
  {
    const u32* src = reinterpret_cast<const u32*>(&p.lv[l]);
    u32* dst = reinterpret_cast<u32*>(&S->lv);
    for (u32 i = tid; i < (u32)(sizeof(TailLevel) / 4); i += NT) dst[i] = src[i];
  }
  const TailLevel& lv = S->lv;  // (complete behind the first barrier below; the two fields used before it come from p)
  const u32 nu = p.lv[l].units, nslots = nu * K;
  // the level's unit lists are one contiguous array of nu * K key slots (zero-padded behind every unit's count)
  const u64* keys = p.cand + ((size_t)b * p.units_per_image + p.lv[l].unit_base) * K;

  // the first kSelReg slots of every thread stay in registers for the passes below (SSD level 0: 8 units x 300 = 2400 slots)
  u64 kreg[kSelReg];
#pragma unroll
  for (u32 t = 0; t < kSelReg; ++t) {
    const u32 i = tid + t * NT;
    kreg[t] = i < nslots ? keys[i] : 0ull;
  }
  for (u32 i = tid; i < kTailBins; i += NT) S->hist[i] = 0;
  if (tid < kSelUnits) {
    S->umin[tid] = ~0ull;
    S->ufull[tid] = 0;
  }
  if (tid == 0) {
    S->lb = 0ull;
    S->cutbin = -1;
    S->ln = 0;
    S->above = 0;
    S->nw = 0;
    S->bcnt = 0;
    S->generic = 0;
    S->smax = 0;
  }
  // f(key, i) for every non-zero key slot of the level (all lanes of a wave run the same trips)
  auto for_each_key = [&](auto f) {
#pragma unroll
    for (u32 t = 0; t < kSelReg; ++t)
      if (t * NT < nslots) f(kreg[t], tid + t * NT);  // workgroup-uniform predicate
    for (u32 i0 = kSelReg * NT; i0 < nslots; i0 += NT) {
      const u32 i = i0 + tid;
      f(i < nslots ? keys[i] : 0ull, i);
    }
  };
  __syncthreads();
  u64 lb = 0ull;  // lower bound of the level's K-th key (0: none needed so far)
  if (stamp) p.stamps[1] = clock64();

  // ---- ties fast path (round 5).  scan16_kernel flags a unit whose list is nothing but its FIRST K ties, in index order (bit
  // 31 of its count): every key of it carries one score s.  If that is the level's first unit, its list is full and no key of
  // the level has a larger score, the list IS the level's top K under the (score desc, index asc) contract -- every other key
  // with score s has a larger index -- already in final order: no histogram, no scatter, no rank.  This is the reference-init
  // network (box.py:446 on an all-equal level), where the generic route cost two histogram passes with every key on ONE LDS
  // counter, a 64-bit atomicMin pass and K^2 rank compares (26.7 vs 12.9 us for the launch).
  bool fast = false;
  {
    const u32 c0 = p.cand_cnt[(size_t)b * p.units_per_image + p.lv[l].unit_base];  // workgroup-uniform
    if ((c0 & 0x80000000u) && (c0 & 0x7fffffffu) == K) {
      u32 mx = 0;
      for_each_key([&](u64 key, u32) { mx = (u32)(key >> 32) > mx ? (u32)(key >> 32) : mx; });
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const u32 y = (u32)__shfl_xor((int)mx, d);
        mx = y > mx ? y : mx;
      }
      if (lane == 0) atomicMax(&S->smax, mx);
      __syncthreads();
      fast = S->smax == (u32)(keys[0] >> 32);  // (slot 0 = the flagged unit's first key; a uniform load)
      if (fast) {
#pragma unroll
        for (u32 t = 0; t < kSelReg; ++t) {
          const u32 i = tid + t * NT;
          if (i < K) wkeys[i] = kreg[t];  // (K <= 512 <= kSelReg * NT: the first unit's list sits in the register keys)
        }
        if (tid == 0) {
          S->nw = K;
          S->above = 0;  // every winner is "boundary": its slot is its rank
        }
      }
    }
  }
  if (!fast) {

  // ---- B1 pass 1: histogram of the keys, the boundary bin, the groups' first slots.  Trip 0 takes every key.  If its
  // boundary bin holds more than K keys and the level has several units, trip 1 first derives a LOWER BOUND of the level's
  // K-th key: a full list proves that K keys are >= its smallest one, so the largest such minimum over the level's units
  // bounds the K-th key from below.  (An all-equal image -- the reference-init network -- leaves the first K indices in EVERY
  // unit: only the first unit's keys survive the bound and no selection is left.)
  for (int trip = 0; trip < 2; ++trip) {  // workgroup-uniform
    if (trip == 1) {
      const bool again = S->generic && nu > 1u && nu <= kSelUnits;
      __syncthreads();  // every thread has read the flag before thread 0 clears it below (a late wave must not see the reset)
      if (!again) break;
      const u32 mK = K <= 1u ? 0xffffffffu : (u32)((1ull << 32) / K);
      for_each_key([&](u64 key, u32 i) {
        if (key == 0ull) return;
        u32 j;
        const u32 u = tail_divmod(i, K, mK, &j);
        atomicMin(reinterpret_cast<unsigned long long*>(&S->umin[u]), key);
        if (j == K - 1u) S->ufull[u] = 1u;  // the list's last slot is used: a full list
      });
      for (u32 i = tid; i < kTailBins; i += NT) S->hist[i] = 0;
      if (tid == 0) {
        S->cutbin = -1;
        S->ln = 0;
        S->above = 0;
        S->nw = 0;
        S->generic = 0;
      }
      __syncthreads();
      if (tid < nu && S->ufull[tid]) atomicMax(reinterpret_cast<unsigned long long*>(&S->lb), S->umin[tid]);
      __syncthreads();
      lb = S->lb;
    }
    {
      u32 mine = 0;
      for_each_key([&](u64 key, u32) {
        const bool in = key != 0ull && key >= lb;
        if (in) atomicAdd(&S->hist[tail_bin(key, hbase, hsh)], 1u);
        mine += in ? 1u : 0u;
      });
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
      if (lane == 0 && mine) atomicAdd(&S->ln, mine);
    }
    __syncthreads();
    if (stamp && trip == 0) p.stamps[21] = clock64();
    // wave 0: the bin in which the count from the top reaches K, and every bin's first slot (suffix sums)
    if (wave == 0) {
      constexpr u32 BPL = kTailBins / 64;
      u32 c[BPL], local = 0;
#pragma unroll
      for (u32 j = 0; j < BPL; j += 4) {
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(&S->hist[lane * BPL + j]);
        c[j] = q4[0];
        c[j + 1] = q4[1];
        c[j + 2] = q4[2];
        c[j + 3] = q4[3];
        local += q4[0] + q4[1] + q4[2] + q4[3];
      }
      u32 incl = local;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_down(incl, d);
        if (lane + d < 64) incl += y;
      }
      const u32 excl = incl - local;  // keys in the bins above this lane's
      const u32 nl = (u32)__builtin_amdgcn_readfirstlane((int)incl);
      if (nl <= K) {
        if (lane == 0) {
          S->cutbin = -1;
          S->above = nl;
          S->nw = nl;
        }
      } else if (excl < K && K <= incl) {  // exactly one lane
        u32 acc = excl;
        for (int j = (int)BPL - 1; j >= 0; --j) {
          if (acc + c[j] >= K) {
            S->cutbin = (int)(lane * BPL + (u32)j);
            S->above = acc;
            S->nw = K;
            S->generic = c[j] > K ? 1u : 0u;
            break;
          }
          acc += c[j];
        }
      }
      if (excl < K) {  // (only the bins at or above the boundary bin are ever scattered into)
        u32 run = excl;
#pragma unroll
        for (int j = (int)BPL - 1; j >= 0; --j) {
          S->hist[lane * BPL + j] = run;  // cursor of the counting-sort scatter
          S->gstart[lane * BPL + j] = (u16)(run < 0xffffu ? run : 0xffffu);
          run += c[j];
        }
      }
    }
    __syncthreads();
  }
  // pass 2: scatter.  Bins above the boundary bin: next free slot of the bin's group; boundary bin: the boundary list.
  // (the returning LDS atomics of a thread's register keys are issued together, the stores follow: one LDS round trip
  // instead of one per key)
  {
    const int cb = S->cutbin;
    const bool generic = S->generic != 0u;
    u32 slot[kSelReg];
#pragma unroll
    for (u32 t = 0; t < kSelReg; ++t) {
      slot[t] = ~0u;
      const u64 key = kreg[t];
      if (t * NT < nslots && key != 0ull && key >= lb) {
        const int bin = (int)tail_bin(key, hbase, hsh);
        if (bin > cb) slot[t] = atomicAdd(&S->hist[(u32)bin], 1u);
        else if (bin == cb && !generic) slot[t] = 0x80000000u | atomicAdd(&S->bcnt, 1u);
      }
    }
#pragma unroll
    for (u32 t = 0; t < kSelReg; ++t)
      if (slot[t] != ~0u) {
        if (slot[t] & 0x80000000u) bkeys[slot[t] & 0x7fffffffu] = kreg[t];
        else wkeys[slot[t]] = kreg[t];
      }
    for (u32 i0 = kSelReg * NT; i0 < nslots; i0 += NT) {
      const u32 i = i0 + tid;
      const u64 key = i < nslots ? keys[i] : 0ull;
      if (!(key != 0ull && key >= lb)) continue;
      const int bin = (int)tail_bin(key, hbase, hsh);
      if (bin > cb) wkeys[atomicAdd(&S->hist[(u32)bin], 1u)] = key;
      else if (bin == cb && !generic) bkeys[atomicAdd(&S->bcnt, 1u)] = key;
    }
  }
  __syncthreads();
  if (stamp) p.stamps[22] = clock64();
  // a boundary bin with more than K keys (heavy ties that the bound above did not remove, coarse bins): the
  // (K - above)-th largest of ITS keys by adaptive radix select straight over the units' lists, then those >= it
  if (S->generic) {  // workgroup-uniform
    const u32 need = K - S->above;
    const int cb = S->cutbin;
    auto fetch = [&](u32 i) -> u64 {
      const u64 key = keys[i];
      return (key != 0ull && key >= lb && (int)tail_bin(key, hbase, hsh) == cb) ? key : 0ull;
    };
    const u64 T = wg_select_kth_f<NT>(fetch, nslots, need, &S->ss);
    for (u32 i = tid; i < nslots; i += NT) {
      const u64 key = fetch(i);
      if (key != 0ull && key >= T) bkeys[atomicAdd(&S->bcnt, 1u)] = key;
    }
    __syncthreads();
  }
  // boundary list: a key whose rank inside the list is below the level's remaining need is a winner -- at its FINAL slot
  {
    const u32 bc = S->bcnt, ab = S->above, nw = S->nw;
    for (u32 i = tid; i < bc; i += NT) {
      const u64 me = bkeys[i];
      const u32 r = lds_count_greater(bkeys, 0, bc, me);
      if (ab + r < nw) wkeys[ab + r] = me;
    }
  }
  }  // !fast
  __syncthreads();
  if (stamp) p.stamps[2] = clock64();
  // ---- order + B2: every winner finds its rank (a key alone in its bin's group is in place; more -- ties in score, coarse
  // bins -- rank inside the group), is decoded and written at l*K + rank of the image's [L*K] arrays; the other slots are
  // zeroed (box.py:430-432)
  {
    const size_t o = ((size_t)b * L + l) * K;
    float* ms = p.mid_scores + o;
    float4* mb = reinterpret_cast<float4*>(p.mid_boxes) + o;
    float* mc = p.mid_classes + o;
    const u32 nw = S->nw, ab = S->above;
    for (u32 s0 = tid; s0 < K; s0 += 2 * NT) {  // two winners per thread and trip: their 8 delta loads fly together
      const u32 s1 = s0 + NT;
      const u64 key0 = s0 < nw ? wkeys[s0] : 0ull, key1 = s1 < nw ? wkeys[s1] : 0ull;
      // branch-free: an empty slot gathers the deltas of index 0 (a valid address) and throws them away -- all eight
      // loads of the thread are issued before the first wait
      const TailGather g0 = tail_gather<DT>(lv, b, key0 != 0ull ? key_index(key0) : 0u);
      const TailGather g1 = tail_gather<DT>(lv, b, key1 != 0ull ? key_index(key1) : 0u);
      if (stamp) p.stamps[13] = clock64();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32 sl = h ? s1 : s0;
        const u64 key = h ? key1 : key0;
        if (sl >= K) continue;
        u32 dst = sl;
        if (key != 0ull && sl < ab) {
          const u32 bin = tail_bin(key, hbase, hsh);
          const u32 q0 = S->gstart[bin], q1 = S->hist[bin];
          if (q1 - q0 > 1u) dst = q0 + lds_count_greater(wkeys, q0, q1, key);
        }
        float sc_ = 0.f, c = 0.f;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key != 0ull) tail_decode(lv, h ? g1 : g0, p.rescore, key_score(key), &sc_, &bx, &c);
        // (the slots below nw are a permutation of themselves; the slots from nw on hold no winner)
        tail_st<COH>(ms + dst, sc_);
        tail_st4<COH>(mb + dst, bx);
        tail_st<COH>(mc + dst, c);
        if (stamp && h == 0) p.stamps[15] = clock64();
      }
    }
  }
  if (stamp) p.stamps[16] = clock64();
}

template <int DT, int NT>
__global__ __launch_bounds__(NT) void levelsel_kernel(const SelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  levelsel_body<DT, NT>(p, blockIdx.x, blockIdx.y, smem);
}

// ------------------------------------------------------------------------------------------------------------------------
// nmswalk_kernel
// ------------------------------------------------------------------------------------------------------------------------
struct alignas(16) WalkLds {
  u64 rows[64];
  u64 blk_dead;
  u32 nvalid, nk, topcnt, hasnan;  // hasnan: a candidate of the walk has a NaN coordinate (rescoring off, NaN deltas)
  u32 ccut, chead, cpad0, cpad1;   // C: lowest bin of the round, keys in it and above
  u32 chist[kCBins];               // score bins of the walk order; after the cut: next free slot of every bin's group
  u16 cstart[kCBins];              // first slot of a bin's group
  u64 top_u[kRoundMax];            // the round's keys grouped by bin (descending)
  u64 sorted[kRoundMax];           // ... in walk order
  float4 rbox[kRoundMax];          // the round's boxes / classes, in walk order
  float rcls[kRoundMax];
  SelScratch ss;
};
// [WalkLds][nkeys: Mp x 8][kbox: ndet x 16][karea | kcls: ndet x 8][abox: Mp x 16][acls: Mp x 4]
__host__ __device__ inline size_t walk_lds_bytes(u32 M, u32 ndet) {
  const size_t Mp = M < 128u ? 128u : M;
  return sizeof(WalkLds) + Mp * 8 + (size_t)ndet * 16 + ((((size_t)ndet * 8) + 15) & ~(size_t)15) + Mp * 16 + Mp * 4 + 64;
}

template <bool COH = false>
__device__ __forceinline__ void nmswalk_body(const WalkParams& p, const u32 b, unsigned char* smem) {
  constexpr int NT = kWalkThreads;
  constexpr u32 NW = NT / 64;
  WalkLds* S = reinterpret_cast<WalkLds*>(smem);
  const u32 N = p.N, M = p.M, ndet = (u32)p.ndet;
  const u32 Mp = M < 128u ? 128u : M;
  unsigned char* q = smem + sizeof(WalkLds);
  u64* nkeys = reinterpret_cast<u64*>(q);  // [Mp] (rescored score | ~position) of the candidates not yet walked
  q += (size_t)Mp * 8;
  float4* kbox = reinterpret_cast<float4*>(q);
  q += (size_t)ndet * 16;
  float* karea = reinterpret_cast<float*>(q);
  float* kcls = karea + ndet;
  q += (((size_t)ndet * 8) + 15) & ~(size_t)15;
  // every candidate's box and class, fetched together with its score (round 5: the round's gather was a second, dependent
  // memory round trip per round; 20 bytes x L*K = 36 KB of LDS on a kernel that owns its CU anyway)
  float4* abox = reinterpret_cast<float4*>(q);
  q += (size_t)Mp * 16;
  float* acls = reinterpret_cast<float*>(q);
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool stamp = p.stamps != nullptr && b == 0 && tid == 0;
  if (stamp) p.stamps[3] = clock64();
  const float* sc = p.scores + (size_t)b * N;
  const float4* bx = reinterpret_cast<const float4*>(p.boxes) + (size_t)b * N;
  const float* cl = p.classes + (size_t)b * N;
  if (tid == 0) {
    S->blk_dead = 0ull;
    S->nvalid = 0;
    S->nk = 0;
    S->topcnt = 0;
    S->hasnan = 0;
  }
  __syncthreads();
  for (u32 pos = tid; pos < Mp; pos += NT) {  // (Mp is a multiple of 64: whole waves)
    u64 nkey = 0ull;
    if (pos < N) {
      const float s = tail_ld<COH>(sc + pos);
      abox[pos] = tail_ld4<COH>(bx + pos);
      acls[pos] = tail_ld<COH>(cl + pos);
      nkey = (s > 0.0f) ? make_key(s, pos) : 0ull;  // box.py:496 (NaN drops out too)
    }
    nkeys[pos] = nkey;
    const u64 m = __ballot(nkey != 0ull);
    if (lane == 0 && m) atomicAdd(&S->nvalid, (u32)__popcll(m));
  }
  __syncthreads();
  if (stamp) p.stamps[6] = clock64();

  // ---- C + D: lazy order of the walk, greedy walk ---------------------------------------------------------------------------
  float* os = p.out_scores + (size_t)b * ndet;
  float4* ob = reinterpret_cast<float4*>(p.out_boxes) + (size_t)b * ndet;
  float* oc = p.out_classes + (size_t)b * ndet;
  const float thr = p.thr;
  const int diou = p.diou;
  u32 left = S->nvalid;
  u32 nk = 0;
  bool first = true;
  auto cbin_of = [&](u64 k) -> u32 {
    const u32 o = (u32)(k >> 48);
    const u32 bin = o < kCBase ? 0u : o - kCBase;
    return bin < kCBins - 1 ? bin : kCBins - 1;
  };
  while (left > 0 && nk < ndet) {  // workgroup-uniform
    // The round = the candidates of the highest score bins that together hold >= min(256, left) keys: one histogram pass
    // (one bin per bf16 value: with tie-free scores the boundary bin adds a handful of keys), put in order by the same
    // counting sort as the level select.  More than 512 that way (massive ties in the rescored scores), or a round that
    // reaches the catch-all bin 0 -> the exact 256 by adaptive radix select, ranked by counting.
    for (u32 i = tid; i < kCBins; i += NT) S->chist[i] = 0;
    __syncthreads();
    for (u32 i = tid; i < Mp; i += NT) {
      const u64 k = nkeys[i];
      if (k != 0ull) atomicAdd(&S->chist[cbin_of(k)], 1u);
    }
    __syncthreads();
    if (stamp && first) p.stamps[7] = clock64();
    const u32 want = left < kRound ? left : kRound;
    if (wave == 0) {
      constexpr u32 BPL = kCBins / 64;
      u32 local = 0;
      u32 c[BPL];
#pragma unroll
      for (u32 j = 0; j < BPL; j += 4) {
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(&S->chist[lane * BPL + j]);
        c[j] = q4[0];
        c[j + 1] = q4[1];
        c[j + 2] = q4[2];
        c[j + 3] = q4[3];
        local += q4[0] + q4[1] + q4[2] + q4[3];
      }
      u32 incl = local;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_down(incl, d);
        if (lane + d < 64) incl += y;
      }
      const u32 excl = incl - local;
      if (excl < want && want <= incl) {  // exactly one lane (left = number of non-zero keys >= want)
        u32 acc = excl;
        for (int j = (int)BPL - 1; j >= 0; --j) {
          acc += c[j];
          if (acc >= want) {
            S->ccut = lane * BPL + (u32)j;
            S->chead = acc;
            break;
          }
        }
      }
      if (excl < want) {  // (only the bins of the round are ever scattered into)
        u32 run = excl;
#pragma unroll
        for (int j = (int)BPL - 1; j >= 0; --j) {
          S->chist[lane * BPL + j] = run;  // cursor of the scatter
          S->cstart[lane * BPL + j] = (u16)(run < 0xffffu ? run : 0xffffu);
          run += c[j];
        }
      }
    }
    if (tid == 0) S->topcnt = 0;
    __syncthreads();
    if (stamp && first) p.stamps[8] = clock64();
    u32 r = S->chead;
    if (r > kRoundMax || (S->ccut == 0u && r > want)) {
      r = want;
      // Ties fast path (round 5): the round's boundary bin holds too many keys.  If every key of that bin carries the SAME
      // score (the reference-init network: all rescored scores of an image are one value), the keys order by position, and
      // nkeys[] is indexed by position: the round = every key of the bins above + the first `need` keys of the boundary bin
      // in array order -- one min / max pass and one prefix count instead of the adaptive radix select (31.5 vs 14.9 us for
      // the launch on the bench's own input).
      const u32 cc = S->ccut;
      u32 lo = 0xffffffffu, hi2 = 0u, inb = 0u;
      for (u32 i = tid; i < Mp; i += NT) {
        const u64 k = nkeys[i];
        if (k != 0ull && cbin_of(k) == cc) {
          const u32 sbits = (u32)(k >> 32);
          lo = sbits < lo ? sbits : lo;
          hi2 = sbits > hi2 ? sbits : hi2;
          ++inb;
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const u32 a = (u32)__shfl_xor((int)lo, d), b2 = (u32)__shfl_xor((int)hi2, d);
        lo = a < lo ? a : lo;
        hi2 = b2 > hi2 ? b2 : hi2;
        inb += (u32)__shfl_xor((int)inb, d);
      }
      if (tid == 0) {
        S->ss.bin = 0xffffffffu;  // min
        S->ss.above = 0u;         // max
        S->ss.cb = 0u;            // keys in the boundary bin
      }
      __syncthreads();
      if (lane == 0 && inb) {
        atomicMin(&S->ss.bin, lo);
        atomicMax(&S->ss.above, hi2);
        atomicAdd(&S->ss.cb, inb);
      }
      __syncthreads();
      const bool uniform = S->ss.bin == S->ss.above && S->ss.cb >= 1u;
      const u32 higher = S->chead - S->ss.cb;  // keys in the bins above the boundary bin (all of them belong to the round)
      if (uniform && higher < want) {  // workgroup-uniform
        const u32 need = want - higher;
        u32 run = 0;  // boundary-bin keys in front of this trip's positions
        for (u32 i0 = 0; i0 < Mp; i0 += NT) {  // uniform trip count (Mp is a multiple of 64; whole waves)
          const u32 i = i0 + tid;
          const u64 k = i < Mp ? nkeys[i] : 0ull;
          const bool have = k != 0ull;
          const u32 bin = have ? cbin_of(k) : 0u;
          const bool edge = have && bin == cc, up = have && bin > cc;
          const u64 me = __ballot(edge);
          if (lane == 0) S->ss.wsum[wave] = (u32)__popcll(me);
          __syncthreads();
          u32 before = run, all = 0;
#pragma unroll
          for (u32 w = 0; w < NW; ++w) {
            const u32 cw = S->ss.wsum[w];
            before += w < wave ? cw : 0u;
            all += cw;
          }
          const bool take = up || (edge && before + mbcnt(me) < need);
          if (take) {
            S->top_u[atomicAdd(&S->topcnt, 1u)] = k;
            nkeys[i] = 0ull;
          }
          run += all;
          __syncthreads();
        }
      } else {
      u64 T = 1ull;  // every remaining (non-zero) key
      if (left > r) T = wg_select_kth<NT>(nkeys, Mp, r, &S->ss);  // r-th largest of the remaining keys (box.py:505)
      for (u32 i = tid; i < Mp; i += NT) {
        const u64 k = nkeys[i];
        if (k != 0ull && k >= T) {
          S->top_u[atomicAdd(&S->topcnt, 1u)] = k;
          nkeys[i] = 0ull;
        }
      }
      }
      __syncthreads();
      for (u32 i = tid; i < r; i += NT) {
        const u64 me = S->top_u[i];
        S->sorted[lds_count_greater(S->top_u, 0, r, me)] = me;
      }
    } else {
      const u32 cc = S->ccut;
      for (u32 i = tid; i < Mp; i += NT) {
        const u64 k = nkeys[i];
        if (k == 0ull) continue;
        const u32 bin = cbin_of(k);
        if (bin >= cc) {
          S->top_u[atomicAdd(&S->chist[bin], 1u)] = k;
          nkeys[i] = 0ull;
        }
      }
      __syncthreads();
      if (stamp && first) p.stamps[17] = clock64();
      for (u32 s = tid; s < r; s += NT) {
        const u64 k = S->top_u[s];
        const u32 bin = cbin_of(k);
        const u32 g0 = S->cstart[bin], g1 = S->chist[bin];
        S->sorted[g1 - g0 > 1u ? g0 + lds_count_greater(S->top_u, g0, g1, k) : s] = k;
      }
    }
    __syncthreads();
    // the round's boxes and classes, in walk order (one gather for the whole round)
    for (u32 i = tid; i < r; i += NT) {
      const u32 pos = key_index(S->sorted[i]);
      const float4 bb = abox[pos];
      S->rbox[i] = bb;
      S->rcls[i] = acls[pos];
      if ((bb.x != bb.x) | (bb.y != bb.y) | (bb.z != bb.z) | (bb.w != bb.w)) S->hasnan = 1u;
    }
    __syncthreads();
    const bool fastnms = S->hasnan == 0u;  // workgroup-uniform (sticky: kept boxes of earlier rounds stay in play)
    if (stamp && first) {
      p.stamps[9] = clock64();
      p.stamps[11] = r;
    }
    first = false;
    for (u32 base = 0; base < r && nk < ndet; base += 64) {  // workgroup-uniform
      const u32 i = base + lane;
      const bool valid = i < r;
      float score = 0.f, cls = -1.f, area = 0.f;
      float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        score = key_score(S->sorted[i]);
        box = S->rbox[i];
        cls = S->rcls[i];
        area = (box.z - box.x + 1.0f) * (box.w - box.y + 1.0f);  // box.py:507
      }
      bool alive = valid;
      for (u32 k = wave; k < nk; k += NW) {
        const float ck = kcls[k];
        if (__ballot(alive && cls == ck) == 0ull) continue;
        const bool sup = (cls == ck) && (fastnms ? tail_suppressed_by<true>(box, area, kbox[k], karea[k], thr, diou)
                                                 : tail_suppressed_by<false>(box, area, kbox[k], karea[k], thr, diou));
        alive = alive && !sup;
      }
      const u64 dead = __ballot(valid && !alive);
      if (lane == 0 && dead) atomicOr(&S->blk_dead, dead);
      constexpr u32 PPW = 64 / NW;
      for (u32 jj = 0; jj < PPW; ++jj) {
        const u32 j = wave * PPW + jj;
        const float cj = tail_bcast(cls, j);
        const u64 m = __ballot(valid && lane > j && cls == cj);
        u64 row = 0;
        if (m != 0ull) {
          float4 pj;
          pj.x = tail_bcast(box.x, j);
          pj.y = tail_bcast(box.y, j);
          pj.z = tail_bcast(box.z, j);
          pj.w = tail_bcast(box.w, j);
          const float aj = tail_bcast(area, j);
          row = __ballot(((m >> lane) & 1ull) && (fastnms ? tail_suppressed_by<true>(box, area, pj, aj, thr, diou)
                                                          : tail_suppressed_by<false>(box, area, pj, aj, thr, diou)));
        }
        if (lane == 0) S->rows[j] = row;
      }
      __syncthreads();
      if (wave == 0) {
        u64 am = __ballot(valid) & ~S->blk_dead;
        const u64 myrow = S->rows[lane];
        const u32 row_lo = (u32)myrow, row_hi = (u32)(myrow >> 32);
        // In-order resolve on bitmasks: only pivots that are still alive AND suppress somebody need a step (a pivot
        // without a row changes nothing).  The truncation to `ndet` survivors (box.py:512) commutes with it: whatever a
        // pivot beyond the cut suppresses lies behind it, i.e. beyond the cut as well.
        const u64 has_row = __ballot(myrow != 0ull);
        u64 todo = am & has_row;
        while (todo) {
          const u32 j = (u32)__ffsll((long long)todo) - 1u;
          const u64 rj = (u64)(u32)__builtin_amdgcn_readlane((int)row_lo, (int)j) |
                         ((u64)(u32)__builtin_amdgcn_readlane((int)row_hi, (int)j) << 32);
          am &= ~rj;
          todo = am & has_row & ~((2ull << j) - 1ull);  // alive pivots with a row after j
        }
        if (nk + (u32)__popcll(am) > ndet) {  // keep the first ndet - nk alive candidates
          const u32 room = ndet - nk;
          const u64 over = __ballot(((am >> lane) & 1ull) && mbcnt(am) >= room);
          am &= ~over;
        }
        const bool keep = (am >> lane) & 1ull;
        const u32 slot = nk + mbcnt(am);
        if (keep && slot < ndet) {
          kbox[slot] = box;
          karea[slot] = area;
          kcls[slot] = cls;
          os[slot] = score;
          ob[slot] = box;
          oc[slot] = cls;
        }
        u32 nk2 = nk + (u32)__popcll(am);
        if (nk2 > ndet) nk2 = ndet;
        if (lane == 0) {
          S->nk = nk2;
          S->blk_dead = 0ull;
        }
      }
      __syncthreads();
      nk = S->nk;
    }
    left -= r;
  }
  if (stamp) {
    p.stamps[4] = clock64();
    p.stamps[12] = nk;
  }

  // ---- E: zero padding (box.py:489-491) ------------------------------------------------------------------------------------
  for (u32 i = nk + tid; i < ndet; i += NT) {
    os[i] = 0.f;
    ob[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    oc[i] = 0.f;
  }
  if (stamp) p.stamps[5] = clock64();
}

__global__ __launch_bounds__(kWalkThreads) void nmswalk_kernel(const WalkParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  nmswalk_body<false>(p, blockIdx.x, smem);
}

// (Round 4 built levelsel + nmswalk as ONE launch -- tail2_kernel: the last workgroup of an image to arrive on a per-image ticket
//  walked its NMS -- and measured it SLOWER than the two launches (55.8 / 57.5 vs 50.9 us for the stage, profiles/r04_tail2_ab.txt:
//  the walk's 1 800 score loads per image had to come through the coherent L2 path).  It rode along behind SSDK_TAIL2 for two
//  rounds; round 6 removed the kernel, its launcher and the switch.  The COH template parameter of the two bodies is what is
//  left of it: always false.)
// ------------------------------------------------------------------------------------------------------------------
// host side (called from ssdk_decode_nms, ssdk_ctx.cpp)
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kTailLdsMax = 160 * 1024;

static u32 tail_pow2(u32 n) {
  u32 m = 2;
  while (m < n) m <<= 1;
  return m;
}

// non-zero when the two kernels of this file can take the geometry (otherwise level_kernel + nms_kernel run).  Independent
// of the number of scan units: their lists are read from the workspace, not staged.
size_t tail_fits(int K, int L, int ndet) {
  if (K < 1 || K > 2 * kSelThreads || L < 1 || L > SSDK_MAX_LEVELS || ndet < 1) return 0;
  const u32 M = tail_pow2((u32)(L * K));
  if (M > 4096) return 0;
  const size_t need = walk_lds_bytes(M, (u32)ndet);
  return need <= kTailLdsMax ? need : 0;
}

static void fill_sel_params(SelParams& p, const ssdk_level* lv, int L, int dtype, int K, int rescore, const u32* units,
                            const u32* unit_base, u32 units_per_image, const void* cand, const void* cand_cnt, u32 hist_base,
                            u32 hist_sh, float* ms, float* mb, float* mc, unsigned long long* stamps) {
  memset(&p, 0, sizeof(p));
  for (int l = 0; l < L; ++l) {
    p.lv[l].box = lv[l].box;
    p.lv[l].A = lv[l].A;
    p.lv[l].C = lv[l].C;
    p.lv[l].H = lv[l].H;
    p.lv[l].W = lv[l].W;
    p.lv[l].stride = lv[l].stride;
    p.lv[l].units = units[l];
    p.lv[l].unit_base = unit_base[l];
    auto magic = [](int d) -> u32 { return d <= 1 ? 0xffffffffu : (u32)((1ull << 32) / (unsigned)d); };
    p.lv[l].mW = magic(lv[l].W);
    p.lv[l].mH = magic(lv[l].H);
    p.lv[l].mC = magic(lv[l].C);
    memcpy(p.lv[l].anchors, lv[l].anchors, sizeof(float) * 4 * lv[l].A);
  }
  p.L = L;
  p.dtype = dtype;
  p.rescore = rescore;
  p.units_per_image = units_per_image;
  p.K = (u32)K;
  p.hist_base = hist_base;
  p.hist_sh = hist_sh;
  p.cand = (const u64*)cand;
  p.cand_cnt = (const u32*)cand_cnt;
  p.mid_scores = ms;
  p.mid_boxes = mb;
  p.mid_classes = mc;
  p.stamps = stamps;
}

int launch_levelsel(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const u32* units, const u32* unit_base,
                    u32 units_per_image, const void* cand, const void* cand_cnt, u32 hist_base, u32 hist_sh, float* ms,
                    float* mb, float* mc, unsigned long long* stamps, hipStream_t stream) {
  if (!ms || !mb || !mc || ((uintptr_t)mb & 15)) {
    set_error("decode_nms: null or misaligned per-level output pointer (boxes need 16-byte alignment)");
    return SSDK_E_BADARG;
  }
  SelParams p;
  fill_sel_params(p, lv, L, dtype, K, rescore, units, unit_base, units_per_image, cand, cand_cnt, hist_base, hist_sh, ms, mb, mc,
                  stamps);
  lds_poison(stream);
  const dim3 grid((unsigned)L, (unsigned)B);
  u32 most = 0;
  for (int l = 0; l < L; ++l) most = units[l] > most ? units[l] : most;
  constexpr int env_wide = 1;  // (round 6: the SSDK_LEVELSEL_WIDE switch is gone, its A/B is settled)
  const bool wide = env_wide == 2 || (env_wide && (size_t)most * (size_t)K > 4096);  // (2: always -- A/B switch)
#define SSDK_SEL(DT)                                                                                                        \
  do {                                                                                                                     \
    if (wide) hipLaunchKernelGGL((levelsel_kernel<DT, 1024>), grid, dim3(1024), sel_lds_bytes((u32)K), stream, p);         \
    else hipLaunchKernelGGL((levelsel_kernel<DT, kSelThreads>), grid, dim3(kSelThreads), sel_lds_bytes((u32)K), stream, p); \
  } while (0)
  if (dtype == SSDK_F32) SSDK_SEL(SSDK_F32);
  else if (dtype == SSDK_BF16) SSDK_SEL(SSDK_BF16);
  else SSDK_SEL(SSDK_F16);
#undef SSDK_SEL
  return check_launch("levelsel_kernel");
}

int launch_nmswalk(const float* ms, const float* mb, const float* mc, int B, int N, float nms_thr, int ndet, int diou, float* os,
                   float* ob, float* oc, unsigned long long* stamps, hipStream_t stream) {
  if (!os || !ob || !oc || ((uintptr_t)ob & 15)) {
    set_error("decode_nms: null or misaligned output pointer (boxes need 16-byte alignment)");
    return SSDK_E_BADARG;
  }
  WalkParams p;
  memset(&p, 0, sizeof(p));
  p.N = (u32)N;
  p.M = tail_pow2((u32)N);
  p.scores = ms;
  p.boxes = mb;
  p.classes = mc;
  p.thr = nms_thr;
  p.ndet = ndet;
  p.diou = diou;
  p.out_scores = os;
  p.out_boxes = ob;
  p.out_classes = oc;
  p.stamps = stamps;
  const size_t lds = walk_lds_bytes(p.M, (u32)ndet);
  // (the attribute belongs to the current device's function object: set on every launch, like the other kernels do)
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(nmswalk_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTailLdsMax) != hipSuccess) {
    (void)hipGetLastError();
    set_error("decode_nms: cannot raise the dynamic LDS limit of nmswalk_kernel");
    return SSDK_E_LAUNCH;
  }
  lds_poison(stream);
  hipLaunchKernelGGL(nmswalk_kernel, dim3((unsigned)B), dim3(kWalkThreads), lds, stream, p);
  return check_launch("nmswalk_kernel");
}

}  // namespace ssdk
