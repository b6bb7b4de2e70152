// ssdk_tail.hip -- everything of Decoder.__call__ (reference ssds/modeling/layers/decoder.py:25-49) behind the scan, as ONE
// launch: per image one workgroup of 16 waves that
//
//   B1. selects the top K of every LEVEL from the <= K unordered keys each scan unit left in the workspace (box.py:446
//       topk over the whole level) and puts them in order -- by COUNTING, not by sorting or merging: a 1024-bin histogram
//       of the score per level (the same window the scan uses: one bin per bf16 value between the threshold and 1), the
//       bin in which the count from the top reaches K, a counting-sort scatter of the bins above it (their start offsets
//       are the histogram's suffix sums), a rank count inside the boundary bin and inside every bin that holds more than
//       one key.  With tie-free scores a bin holds one to three keys, so the "sort" is one LDS atomic per key;
//   B2. decodes the winners: gather of the 4 deltas, delta2box (box.py:74-87), centre rescoring (box.py:464-471), into
//       LDS at position l*K + r -- the slot torch.cat gives the r-th output of level l (decoder.py:48);
//   C.  orders the walk of box.nms lazily (box.py:505): the exact top 256 of the candidates with score > 0 (box.py:496)
//       by adaptive radix select on (rescored score, position) keys, ranked by counting; more rounds only while fewer
//       than `ndetections` boxes survived;
//   D.  walks them 64 at a time like nms_kernel (ssdk_nms.hip): kept-list test split over the 16 waves, suppression rows
//       of 4 pivots per wave, in-order resolve on bitmasks by wave 0 (box.py:512-544);
//   E.  writes the zero-padded [ndetections] outputs (box.py:489-491).
//
// Round 2's kernel merged SORTED unit lists by binary-search ranks (10 k cycles), sorted all L*K candidates as 128-key
// register runs and ranked a 512-key head among them (41 k cycles of 83 k).  Nothing is sorted here: the scan emits its
// winners unordered (ssdk_scan16.hip), B1 is ~5 k cycles and C ~5 k.  Arithmetic and tie order are unchanged: same
// helpers, same fp32 sequences, keys are unique (score bits | ~index), so "top K" and "rank" are well defined.
#include "ssdk_common.h"
#include "ssdk_select.h"
#include "ssdk_decode.h"

namespace ssdk {

constexpr int kTailThreads = 1024;
constexpr u32 kRound = 256;      // candidates ordered + walked per NMS round (target; a round takes whole score bins)
constexpr u32 kRoundMax = 512;   // ... and at most this many
constexpr u32 kCBins = 2048;     // score bins of the walk order (C): one per bf16 value from 2^-11 up to 32
constexpr u32 kCBase = 0xba00u;  // ord(score) >> 16 of 2^-11
constexpr u32 kTailBins = 1024;  // score bins per level (B1)
constexpr u32 kRegKeys = 8;      // keys per lane of a wave's first unit kept in registers (K <= 512)

struct TailLevel {
  const void* box;
  int A, C, H, W, stride;
  u32 units, unit_base, pad;
  u32 mW, mH, mC, pad2;  // floor(2^32 / d): quotient by multiply-high + one correction (tail_divmod)
  float anchors[SSDK_MAX_ANCHORS * 4];
};
struct TailParams {
  TailLevel lv[SSDK_MAX_LEVELS];
  int L, dtype, rescore;
  u32 units_per_image, K;
  u32 M;  // power of two >= L*K: length of the NMS key array
  u32 hist_base, hist_sh;  // score window of B1: bin = (ord(score) - hist_base) >> hist_sh, clamped (ssdk_decode.hip)
  const u64* cand;
  const u32* cand_cnt;
  float thr;
  int ndet, diou;
  float* out_scores;
  float* out_boxes;
  float* out_classes;
  float* mid_scores;  // optional [B, L*K], [B, L*K, 4], [B, L*K]
  float* mid_boxes;
  float* mid_classes;
  unsigned long long* stamps;  // optional (debug): shader-clock stamps of workgroup 0 at the phase boundaries
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

__device__ __forceinline__ TailGather tail_gather(const TailLevel& d, int dtype, u32 b, u32 idx) {
  const u32 W = d.W, H = d.H, C = d.C;
  TailGather g;
  const u32 t1 = tail_divmod(idx, W, d.mW, &g.x);   // x = idx % W
  const u32 t2 = tail_divmod(t1, H, d.mH, &g.y);    // y = (idx / W) % H
  g.a = tail_divmod(t2, C, d.mC, &g.c);             // c = (idx / W / H) % C (box.py:448), a = idx / C / H / W (box.py:454)
  const size_t hw = (size_t)H * W;
  const size_t boff = ((size_t)b * d.A * 4 + (size_t)g.a * 4) * hw + (size_t)g.y * W + g.x;
  g.d0 = load_as_f32(d.box, boff, dtype);
  g.d1 = load_as_f32(d.box, boff + hw, dtype);
  g.d2 = load_as_f32(d.box, boff + 2 * hw, dtype);
  g.d3 = load_as_f32(d.box, boff + 3 * hw, dtype);
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

__device__ __forceinline__ float tail_bcast(float v, u32 j) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)j));
}

__device__ __forceinline__ u32 tail_bin(u64 key, u32 base, u32 sh) {
  const u32 o = (u32)(key >> 32);
  const u32 bin = o < base ? 0u : (o - base) >> sh;  // (every candidate is at or above the threshold, i.e. >= base)
  return bin < kTailBins - 1 ? bin : kTailBins - 1;
}

struct alignas(16) TailLds {  // fixed-size part of the LDS image (the arrays follow, see tail_lds_bytes)
  TailLevel lv[SSDK_MAX_LEVELS];
  u64 rows[64];
  u64 blk_dead;
  u64 lb[SSDK_MAX_LEVELS];      // per level: a lower bound of its K-th key (largest minimum of its FULL unit lists)
  u32 nvalid, nk, topcnt, hasnan;  // hasnan: a candidate of the walk has a NaN coordinate (rescoring off, NaN deltas)
  u32 ccut, chead, cpad0, cpad1;  // C: lowest bin of the round, keys in it and above
  int cutbin[SSDK_MAX_LEVELS];  // -1: the level offers <= K keys, every one is a winner
  u32 ln[SSDK_MAX_LEVELS];      // keys the level's units offer (at or above lb)
  u32 above[SSDK_MAX_LEVELS];   // winners in the bins above the boundary bin
  u32 nw[SSDK_MAX_LEVELS];      // winners of the level: min(K, ln)
  u32 bcnt[SSDK_MAX_LEVELS];    // keys in the boundary list
  u32 generic[SSDK_MAX_LEVELS]; // the boundary bin holds more than K keys: adaptive radix select over the units' keys
  SelScratch ss;
};

__host__ __device__ inline size_t tail_r1_bytes(u32 L, u32 M) {  // histograms of B1, later the NMS keys + one round
  const size_t a = (size_t)L * kTailBins * 4, b = (size_t)(M < 128u ? 128u : M) * 8 + (size_t)kCBins * 4 + 2 * (size_t)kRoundMax * 8;
  return ((a > b ? a : b) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t tail_lds_bytes(u32 K, u32 L, u32 M, u32 ndet) {
  size_t n = sizeof(TailLds);
  n += tail_r1_bytes(L, M);
  n += (size_t)L * kTailBins * 2;                 // gstart
  n += 2 * ((((size_t)L * K * 8) + 15) & ~(size_t)15);  // wkeys, wl (the boundary lists live on wl until it is written)
  n += (size_t)L * K * 16;                        // rec_box
  n += (((size_t)L * K * 8) + 15) & ~(size_t)15;  // rec_score, rec_cls
  n += (size_t)ndet * 16 + ((((size_t)ndet * 8) + 15) & ~(size_t)15);  // kbox, karea, kcls
  return n + 64;
}

__global__ __launch_bounds__(kTailThreads) void tail_kernel(const TailParams p) {
  constexpr int NT = kTailThreads;
  constexpr u32 NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TailLds* S = reinterpret_cast<TailLds*>(smem);
  const u32 K = p.K, L = (u32)p.L, upi = p.units_per_image, M = p.M, ndet = (u32)p.ndet;
  const u32 LK = L * K;
  const u32 Mp = M < 128u ? 128u : M;
  unsigned char* q = smem + sizeof(TailLds);
  u32* hist = reinterpret_cast<u32*>(q);                    // [L][kTailBins]   (B1)
  u64* nkeys = reinterpret_cast<u64*>(q);                   // [Mp]             (B2 .. D, on top of the histograms)
  u32* chist = reinterpret_cast<u32*>(nkeys + Mp);          // [kCBins] score bins of the walk order (C)
  u64* top_u = reinterpret_cast<u64*>(chist + kCBins);      // [kRoundMax] the round's keys, unordered
  u64* sorted = top_u + kRoundMax;                          // [kRoundMax] ... in walk order
  q += tail_r1_bytes(L, M);
  u16* gstart = reinterpret_cast<u16*>(q);                  // [L][kTailBins] first slot of a bin's group
  q += (size_t)L * kTailBins * 2;
  u64* wkeys = reinterpret_cast<u64*>(q);                   // [L][K] winners grouped by bin (descending), boundary winners in order
  q += (((size_t)LK * 8) + 15) & ~(size_t)15;
  u64* wl = reinterpret_cast<u64*>(q);                      // [L][K] winners in order (rank r of level l at l*K + r)
  u64* bkeys = wl;                                          // [L][K] boundary lists (dead before wl is written)
  q += (((size_t)LK * 8) + 15) & ~(size_t)15;
  float4* rec_box = reinterpret_cast<float4*>(q);
  q += (size_t)LK * 16;
  float* rec_score = reinterpret_cast<float*>(q);
  float* rec_cls = rec_score + (size_t)LK;
  q += (((size_t)LK * 8) + 15) & ~(size_t)15;
  float4* kbox = reinterpret_cast<float4*>(q);
  q += (size_t)ndet * 16;
  float* karea = reinterpret_cast<float*>(q);
  float* kcls = karea + ndet;

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.x;
  const bool stamp = p.stamps != nullptr && b == 0 && tid == 0;
  if (stamp) p.stamps[0] = clock64();
  const u32 hbase = p.hist_base, hsh = p.hist_sh;

  // ---- A: geometry, counters, histograms ---------------------------------------------------------------------------------
  {
    const u32* src = reinterpret_cast<const u32*>(&p.lv[0]);
    u32* dst = reinterpret_cast<u32*>(&S->lv[0]);
    for (u32 i = tid; i < (u32)(sizeof(TailLevel) / 4) * L; i += NT) dst[i] = src[i];
  }
  for (u32 i = tid; i < L * kTailBins; i += NT) hist[i] = 0;
  if (tid < SSDK_MAX_LEVELS) {
    S->lb[tid] = 0ull;
    S->ln[tid] = 0;
    S->bcnt[tid] = 0;
    S->generic[tid] = 0;
    S->above[tid] = 0;
    S->nw[tid] = 0;
    S->cutbin[tid] = -1;
  }
  if (tid == 0) {
    S->blk_dead = 0ull;
    S->nvalid = 0;
    S->nk = 0;
    S->topcnt = 0;
    S->hasnan = 0;
  }
  // the unit lists: wave w owns units w, w + 16, ...; the keys of its FIRST unit stay in registers for the passes below
  const u64* cand = p.cand + (size_t)b * upi * K;
  const u32* ccnt = p.cand_cnt + (size_t)b * upi;
  u64 kreg[kRegKeys];
  u32 cnt0 = 0;
  if (wave < upi) {
    // (a unit's list is zero-padded to K by the scan: the keys are requested without waiting for the count)
#pragma unroll
    for (u32 c = 0; c < kRegKeys; ++c) {
      const u32 i = c * 64 + lane;
      kreg[c] = i < K ? cand[(size_t)wave * K + i] : 0ull;
    }
    cnt0 = ccnt[wave];
  } else {
#pragma unroll
    for (u32 c = 0; c < kRegKeys; ++c) kreg[c] = 0ull;
  }
  auto level_of = [&](u32 u) -> u32 {
    u32 l = 0;
    for (u32 v = 1; v < L; ++v)
      if (u >= p.lv[v].unit_base) l = v;
    return l;
  };
  // f(level, key) for every key of this wave's units (every lane of the wave runs the same trips)
  auto for_each_key = [&](auto f) {
    if (wave < upi) {
      const u32 l = level_of(wave);
#pragma unroll
      for (u32 c = 0; c < kRegKeys; ++c)
        if (c * 64 < cnt0) f(l, kreg[c], c * 64 + lane < cnt0);
    }
    for (u32 u = wave + NW; u < upi; u += NW) {
      const u32 l = level_of(u), cnt = ccnt[u];
      for (u32 i0 = 0; i0 < cnt; i0 += 64) {
        const u32 i = i0 + lane;
        const u64 key = i < cnt ? cand[(size_t)u * K + i] : 0ull;
        f(l, key, i < cnt);
      }
    }
  };
  __syncthreads();
  // A full list proves that K keys are >= its smallest one: the largest such minimum over a level's units is a lower bound
  // of the level's K-th key.  (An all-equal image -- the reference-init network -- leaves the first K indices in EVERY
  // unit: only the first unit's keys survive this bound and no selection is needed at all.)
  {
    auto unit_min = [&](u32 u, u32 cnt, auto get) {
      if (cnt < K) return;  // wave-uniform
      u64 mn = ~0ull;
      for (u32 i0 = 0; i0 < cnt; i0 += 64) {
        const u64 k = get(i0);
        mn = (i0 + lane < cnt && k < mn) ? k : mn;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const u64 o = shfl_xor_u64(mn, d);
        mn = o < mn ? o : mn;
      }
      if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(&S->lb[level_of(u)]), mn);
    };
    if (wave < upi) {
      u64 mn = ~0ull;
#pragma unroll
      for (u32 c = 0; c < kRegKeys; ++c) mn = (c * 64 + lane < cnt0 && kreg[c] < mn) ? kreg[c] : mn;
      if (cnt0 >= K) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const u64 o = shfl_xor_u64(mn, d);
          mn = o < mn ? o : mn;
        }
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(&S->lb[level_of(wave)]), mn);
      }
    }
    for (u32 u = wave + NW; u < upi; u += NW) {
      const u32 cnt = ccnt[u];
      unit_min(u, cnt, [&](u32 i0) { return i0 + lane < cnt ? cand[(size_t)u * K + i0 + lane] : ~0ull; });
    }
  }
  __syncthreads();
  if (stamp) p.stamps[1] = clock64();

  // ---- B1: per level, the top K of its units' keys, in order ---------------------------------------------------------------
  // pass 1: histogram of the keys at or above the level's bound
  for_each_key([&](u32 l, u64 key, bool have) {
    const bool in = have && key >= S->lb[l];
    if (in) atomicAdd(&hist[l * kTailBins + tail_bin(key, hbase, hsh)], 1u);
    const u64 m = __ballot(in);
    if (lane == 0 && m) atomicAdd(&S->ln[l], (u32)__popcll(m));
  });
  __syncthreads();
  if (stamp) p.stamps[21] = clock64();
  // one wave per level: the bin in which the count from the top reaches K, and every bin's first slot (suffix sums)
  if (wave < L) {
    const u32 l = wave;
    u32* h = hist + l * kTailBins;
    u16* gs = gstart + l * kTailBins;
    constexpr u32 BPL = kTailBins / 64;
    u32 c[BPL], local = 0;
#pragma unroll
    for (u32 j = 0; j < BPL; ++j) {
      c[j] = h[lane * BPL + j];
      local += c[j];
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
        S->cutbin[l] = -1;
        S->above[l] = nl;
        S->nw[l] = nl;
      }
    } else if (excl < K && K <= incl) {  // exactly one lane
      u32 acc = excl;
      for (int j = (int)BPL - 1; j >= 0; --j) {
        if (acc + c[j] >= K) {
          S->cutbin[l] = (int)(lane * BPL + (u32)j);
          S->above[l] = acc;
          S->nw[l] = K;
          S->generic[l] = c[j] > K ? 1u : 0u;
          break;
        }
        acc += c[j];
      }
    }
    u32 run = excl;
#pragma unroll
    for (int j = (int)BPL - 1; j >= 0; --j) {
      h[lane * BPL + j] = run;  // cursor of the counting-sort scatter (meaningful for the bins above the boundary bin)
      gs[lane * BPL + j] = (u16)(run < 0xffffu ? run : 0xffffu);
      run += c[j];
    }
  }
  __syncthreads();
  // pass 2: scatter.  Bins above the boundary bin: next free slot of the bin's group; boundary bin: the boundary list.
  for_each_key([&](u32 l, u64 key, bool have) {
    if (!(have && key >= S->lb[l])) return;
    const int bin = (int)tail_bin(key, hbase, hsh), cb = S->cutbin[l];
    if (bin > cb) wkeys[l * K + atomicAdd(&hist[l * kTailBins + (u32)bin], 1u)] = key;
    else if (bin == cb && !S->generic[l]) bkeys[l * K + atomicAdd(&S->bcnt[l], 1u)] = key;
  });
  __syncthreads();
  if (stamp) p.stamps[22] = clock64();
  // a boundary bin with more than K keys (heavy ties that the bound above did not remove, coarse bins): the
  // (K - above)-th largest of ITS keys by adaptive radix select straight over the units' lists, then those >= it
  for (u32 l = 0; l < L; ++l) {
    if (!S->generic[l]) continue;  // workgroup-uniform
    const u32 u0 = S->lv[l].unit_base, nu = S->lv[l].units, need = K - S->above[l];
    const int cb = S->cutbin[l];
    const u64 lbl = S->lb[l];
    auto fetch = [&](u32 i) -> u64 {
      const u32 u = u0 + i / K, j = i % K;
      if (j >= ccnt[u]) return 0ull;
      const u64 key = cand[(size_t)u * K + j];
      return (key >= lbl && (int)tail_bin(key, hbase, hsh) == cb) ? key : 0ull;
    };
    const u64 T = wg_select_kth_f<NT>(fetch, nu * K, need, &S->ss);
    for (u32 i = tid; i < nu * K; i += NT) {
      const u64 key = fetch(i);
      if (key != 0ull && key >= T) bkeys[l * K + atomicAdd(&S->bcnt[l], 1u)] = key;
    }
    __syncthreads();
  }
  // boundary lists: a key whose rank inside its list is below the level's remaining need is a winner -- at its FINAL slot
  for (u32 f = tid; f < LK; f += NT) {
    const u32 l = f / K, i = f - l * K, bc = S->bcnt[l];
    if (i >= bc) continue;
    const u64 me = bkeys[l * K + i];
    const u32 r = lds_count_greater(bkeys + l * K, 0, bc, me);
    const u32 ab = S->above[l];
    if (ab + r < S->nw[l]) wkeys[l * K + ab + r] = me;
  }
  __syncthreads();
  // groups of the bins above: one key -> it is in place; more (ties in score, coarse bins) -> rank inside the group.
  // (wl shares its storage with the boundary lists: their last reader is behind the barrier above)
  for (u32 f = tid; f < LK; f += NT) {
    const u32 l = f / K, s = f - l * K;
    if (s >= S->nw[l]) {  // a slot without a winner: fewer than K candidates in the level
      wl[f] = 0ull;
      continue;
    }
    const u64 key = wkeys[f];
    u32 dst = s;
    if (s < S->above[l]) {
      const u32 bin = tail_bin(key, hbase, hsh);
      const u32 g0 = gstart[l * kTailBins + bin], g1 = hist[l * kTailBins + bin];
      if (g1 - g0 > 1u) dst = g0 + lds_count_greater(wkeys + l * K, g0, g1, key);
    }
    wl[l * K + dst] = key;  // (the slots s < nw of a level are a permutation of themselves)
  }
  __syncthreads();
  if (stamp) p.stamps[2] = clock64();

  // ---- B2: decode of the winners -------------------------------------------------------------------------------------------
  float* mid_s = p.mid_scores ? p.mid_scores + (size_t)b * LK : nullptr;
  float4* mid_b = p.mid_boxes ? reinterpret_cast<float4*>(p.mid_boxes) + (size_t)b * LK : nullptr;
  float* mid_c = p.mid_classes ? p.mid_classes + (size_t)b * LK : nullptr;
  for (u32 pos0 = tid; pos0 < Mp; pos0 += 2 * NT) {  // two winners per thread and trip: their 8 delta loads fly together
    const u32 pos1 = pos0 + NT;
    const u64 key0 = pos0 < LK ? wl[pos0] : 0ull, key1 = pos1 < LK ? wl[pos1] : 0ull;
    TailGather g0{}, g1{};
    if (key0 != 0ull) g0 = tail_gather(S->lv[pos0 / K], p.dtype, b, key_index(key0));
    if (key1 != 0ull) g1 = tail_gather(S->lv[pos1 / K], p.dtype, b, key_index(key1));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 pos = h ? pos1 : pos0;
      const u64 key = h ? key1 : key0;
      if (pos >= Mp) continue;  // (whole waves: Mp is a multiple of 64 and NT)
      u64 nkey = 0ull;
      if (pos < LK) {
        float s = 0.f, c = 0.f;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key != 0ull) {
          tail_decode(S->lv[pos / K], h ? g1 : g0, p.rescore, key_score(key), &s, &bx, &c);
          rec_score[pos] = s;
          rec_box[pos] = bx;
          rec_cls[pos] = c;
          nkey = (s > 0.0f) ? make_key(s, pos) : 0ull;  // box.py:496 (NaN drops out too)
          if (nkey != 0ull && ((bx.x != bx.x) | (bx.y != bx.y) | (bx.z != bx.z) | (bx.w != bx.w))) S->hasnan = 1u;
        }
        if (mid_s) {
          mid_s[pos] = s;
          mid_b[pos] = bx;
          mid_c[pos] = c;
        }
      }
      nkeys[pos] = nkey;
      const u64 m = __ballot(nkey != 0ull);
      if (lane == 0 && m) atomicAdd(&S->nvalid, (u32)__popcll(m));
    }
  }
  __syncthreads();
  if (stamp) p.stamps[16] = clock64();

  // ---- C + D: lazy order of the walk, greedy walk ---------------------------------------------------------------------------
  float* os = p.out_scores + (size_t)b * ndet;
  float4* ob = reinterpret_cast<float4*>(p.out_boxes) + (size_t)b * ndet;
  float* oc = p.out_classes + (size_t)b * ndet;
  const float thr = p.thr;
  const int diou = p.diou;
  u32 left = S->nvalid;
  const bool fastnms = S->hasnan == 0u;  // workgroup-uniform
  u32 nk = 0;
  bool first = true;
  while (left > 0 && nk < ndet) {  // workgroup-uniform
    // The round = the candidates of the highest score bins that together hold >= min(256, left) keys: one histogram pass
    // (one bin per bf16 value: with tie-free scores the boundary bin adds a handful of keys) instead of an exact select.
    // More than 512 that way (massive ties in the rescored scores) -> the exact 256 by adaptive radix select.
    for (u32 i = tid; i < kCBins; i += NT) chist[i] = 0;
    __syncthreads();
    if (stamp && first) p.stamps[6] = clock64();
    auto cbin_of = [&](u64 k) -> u32 {
      const u32 o = (u32)(k >> 48);
      const u32 bin = o < kCBase ? 0u : o - kCBase;
      return bin < kCBins - 1 ? bin : kCBins - 1;
    };
    for (u32 i = tid; i < Mp; i += NT) {
      const u64 k = nkeys[i];
      if (k != 0ull) atomicAdd(&chist[cbin_of(k)], 1u);
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
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(&chist[lane * BPL + j]);
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
    }
    if (tid == 0) S->topcnt = 0;
    __syncthreads();
    if (stamp && first) p.stamps[8] = clock64();
    u32 r = S->chead;
    if (r > kRoundMax || (S->ccut == 0u && r > want)) {
      // (bin 0 collects everything below 2^-11: a round that reaches it is not ordered by bins any more)
      r = want;
      u64 T = 1ull;  // every remaining (non-zero) key
      if (left > r) T = wg_select_kth<NT>(nkeys, Mp, r, &S->ss);  // r-th largest of the remaining keys (box.py:505)
      for (u32 i = tid; i < Mp; i += NT) {
        const u64 k = nkeys[i];
        if (k != 0ull && k >= T) {
          top_u[atomicAdd(&S->topcnt, 1u)] = k;
          nkeys[i] = 0ull;
        }
      }
    } else {
      // per-wave counts -> one barrier -> every wave knows its first slot (16 waves adding to one LDS word one after the
      // other cost more than the whole rest of the round)
      const u32 cc = S->ccut;
      constexpr u32 TRIPS = 4096 / NT;  // Mp <= 4096 (tail_fits)
      u64 held[TRIPS];
      u32 mine = 0;
#pragma unroll
      for (u32 t = 0; t < TRIPS; ++t) {
        const u32 i = tid + t * NT;
        held[t] = 0ull;
        if (t * NT < Mp) {  // workgroup-uniform (Mp is a multiple of 64: whole waves)
          const u64 k = i < Mp ? nkeys[i] : 0ull;
          const bool take = k != 0ull && cbin_of(k) >= cc;
          if (take) {
            held[t] = k;
            nkeys[i] = 0ull;
          }
          mine += (u32)__popcll(__ballot(take));
        }
      }
      if (lane == 0) S->rows[wave] = mine;  // (rows[] is free until the walk)
      __syncthreads();
      u32 base = 0;
      for (u32 w = 0; w < wave; ++w) base += (u32)S->rows[w];
#pragma unroll
      for (u32 t = 0; t < TRIPS; ++t) {
        const u64 m = __ballot(held[t] != 0ull);
        if (held[t] != 0ull) top_u[base + mbcnt(m)] = held[t];
        base += (u32)__popcll(m);
      }
    }
    if (stamp && first) p.stamps[17] = clock64();
    __syncthreads();
    if (stamp && first) p.stamps[9] = clock64();
    {  // rank by counting: 4 threads per key (2 above 256 keys), a part of the round each
      const u32 parts = r <= 256u ? 4u : 2u, sh = r <= 256u ? 2u : 1u;
      const u32 i = tid >> sh, part = tid & (parts - 1u);
      const u32 span = (r + parts - 1u) / parts;
      const u64 me = i < r ? top_u[i] : ~0ull;
      const u32 j1 = (part + 1u) * span < r ? (part + 1u) * span : r;
      u32 g = lds_count_greater(top_u, part * span < j1 ? part * span : j1, j1, me);
      g += __shfl_xor(g, 1);
      if (parts == 4u) g += __shfl_xor(g, 2);
      if (part == 0 && i < r) sorted[g] = me;
    }
    if (stamp && first) {
      p.stamps[10] = clock64();
      p.stamps[11] = r;
    }
    __syncthreads();
    if (stamp && first) p.stamps[3] = clock64();
    first = false;
    for (u32 base = 0; base < r && nk < ndet; base += 64) {  // workgroup-uniform
      const u32 i = base + lane;
      const bool valid = i < r;
      float score = 0.f, cls = -1.f, area = 0.f;
      float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const u64 k = sorted[i];
        const u32 pos = key_index(k);
        score = key_score(k);
        box = rec_box[pos];
        cls = rec_cls[pos];
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

// ------------------------------------------------------------------------------------------------------------------
// host side (called from ssdk_decode_nms, ssdk_ctx.cpp)
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kTailLdsMax = 160 * 1024;

static u32 tail_pow2(u32 n) {
  u32 m = 2;
  while (m < n) m <<= 1;
  return m;
}

// LDS bytes the fused tail needs for this geometry, or 0 when it cannot take it (then level_kernel + nms_kernel run).
// Independent of the number of scan units: their lists are read from the workspace, not staged.
size_t tail_fits(int K, int L, int ndet) {
  if (K < 1 || K > (int)(kRegKeys * 64) || L < 1 || L > SSDK_MAX_LEVELS || ndet < 1) return 0;
  const u32 M = tail_pow2((u32)(L * K));
  if (M > 4096) return 0;
  const size_t need = tail_lds_bytes((u32)K, (u32)L, M, (u32)ndet);
  return need <= kTailLdsMax ? need : 0;
}

int launch_tail(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const u32* units, const u32* unit_base,
                u32 units_per_image, const void* cand, const void* cand_cnt, u32 hist_base, u32 hist_sh, float nms_thr,
                int ndet, int diou, float* os, float* ob, float* oc, float* ms, float* mb, float* mc,
                unsigned long long* stamps, hipStream_t stream) {
  const size_t lds = tail_fits(K, L, ndet);
  if (!lds) {
    set_error("decode_nms: geometry does not fit the fused tail kernel");
    return SSDK_E_BADARG;
  }
  if (!os || !ob || !oc || ((uintptr_t)ob & 15) || (mb && ((uintptr_t)mb & 15))) {
    set_error("decode_nms: null or misaligned output pointer (boxes need 16-byte alignment)");
    return SSDK_E_BADARG;
  }
  TailParams p;
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
  p.M = tail_pow2((u32)(L * K));
  p.hist_base = hist_base;
  p.hist_sh = hist_sh;
  p.cand = (const u64*)cand;
  p.cand_cnt = (const u32*)cand_cnt;
  p.thr = nms_thr;
  p.ndet = ndet;
  p.diou = diou;
  p.out_scores = os;
  p.out_boxes = ob;
  p.out_classes = oc;
  p.mid_scores = ms;
  p.mid_boxes = mb;
  p.mid_classes = mc;
  p.stamps = stamps;
  // (the attribute belongs to the current device's function object: set on every launch, like the other kernels do)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kTailLdsMax) != hipSuccess) {
    (void)hipGetLastError();
    set_error("decode_nms: cannot raise the dynamic LDS limit of tail_kernel");
    return SSDK_E_LAUNCH;
  }
  lds_poison(stream);
  hipLaunchKernelGGL(tail_kernel, dim3((unsigned)B), dim3(kTailThreads), lds, stream, p);
  return check_launch("tail_kernel");
}

}  // namespace ssdk
