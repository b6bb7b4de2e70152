// ssdk_tail.hip -- everything of Decoder.__call__ (reference ssds/modeling/layers/decoder.py:25-49) behind the scan,
// as ONE launch: per image one workgroup of 16 waves
//
//   A. stages the SORTED per-unit top-K lists that scan_kernel left in the workspace (ssdk_decode.hip) in LDS;
//   B. merges the units of every level WITHOUT selecting or sorting again: the rank of a key inside its level is its
//      index in its own list plus, for every sibling list, the number of larger keys there (binary search; keys are
//      unique).  A key of rank r < K is the r-th output of box.decode for that level (box.py:446 topk, sorted): its
//      thread gathers the 4 deltas, applies delta2box (box.py:74-87) and the centre rescoring (box.py:464-471) and
//      leaves (score, box, class) in LDS at position l*K + r -- the slot torch.cat gives it (decoder.py:48);
//   C. sorts the L*K candidates with score > 0 (box.py:496; NaN drops out) by (rescored score desc, position asc)
//      (box.py:505, stable-order contract) -- bitonic network with barriers only on the steps that cross waves;
//   D. walks them 64 at a time exactly like nms_kernel (ssdk_nms.hip): kept-list test split over the 16 waves,
//      suppression rows of 4 pivots per wave, in-order resolve on bitmasks by wave 0 (box.py:512-544);
//   E. writes the zero-padded [ndetections] outputs (box.py:489-491).
//
// The 24*L*K-byte per-level decode output that level_kernel wrote and nms_kernel read back never leaves the CU (it is
// still written when the caller asks for it: `mid_*`, decoder.py:48), and two dependent launches become one.
// Arithmetic and tie order are those of level_kernel / nms_kernel: same helpers, same fp32 sequences.
#include "ssdk_common.h"
#include "ssdk_select.h"
#include "ssdk_decode.h"

namespace ssdk {

constexpr int kTailThreads = 1024;
constexpr u32 kHeadMax = 512;  // candidates of the walk's head that are ordered first (phase C)

struct TailLevel {
  const void* box;
  int A, C, H, W, stride;
  u32 units, unit_base, pad;
  float anchors[SSDK_MAX_ANCHORS * 4];
};
struct TailParams {
  TailLevel lv[SSDK_MAX_LEVELS];
  int L, dtype, rescore;
  u32 units_per_image, K;
  u32 M;  // power of two >= L*K: length of the NMS key array
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
__device__ __forceinline__ void tail_decode_one(const TailLevel& d, int dtype, int rescore, u32 b, u32 idx, float score,
                                                float* o_score, float4* o_box, float* o_cls) {
  const u32 W = d.W, H = d.H, C = d.C;
  const u32 x = idx % W;
  const u32 y = (idx / W) % H;
  const u32 c = (idx / W / H) % C;   // box.py:448
  const u32 a = idx / C / H / W;     // box.py:454
  const size_t hw = (size_t)H * W;
  const size_t boff = ((size_t)b * d.A * 4 + (size_t)a * 4) * hw + (size_t)y * W + x;
  const float d0 = load_as_f32(d.box, boff, dtype);
  const float d1 = load_as_f32(d.box, boff + hw, dtype);
  const float d2 = load_as_f32(d.box, boff + 2 * hw, dtype);
  const float d3 = load_as_f32(d.box, boff + 3 * hw, dtype);
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
  const float x1 = tmax(0.0f, tmin(pcx - 0.5f * pw, Mx));           // box.py:84-87
  const float y1 = tmax(0.0f, tmin(pcy - 0.5f * ph, My));
  const float x2 = tmax(0.0f, tmin(pcx + 0.5f * pw - 1.0f, Mx));
  const float y2 = tmax(0.0f, tmin(pcy + 0.5f * ph - 1.0f, My));
  float s = score;
  if (rescore) {  // box.py:464-471
    const float gcx = (g0 + g2) / 2.0f, gcy = (g1 + g3) / 2.0f;
    const float ltx = fabsf(gcx - x1), lty = fabsf(gcy - y1);
    const float rbx = fabsf(x2 - gcx), rby = fabsf(y2 - gcy);
    const float rx = tmin(ltx, rbx) / tmax(ltx, rbx);
    const float ry = tmin(lty, rby) / tmax(lty, rby);
    s = s * sqrtf(rx * ry);
  }
  *o_score = s;
  *o_box = make_float4(x1, y1, x2, y2);
  *o_cls = (float)c;
}

__device__ __forceinline__ bool tail_suppressed_by(const float4 b, float ab, const float4 p, float ap, float thr,
                                                   int diou) {  // box.py:518-533, as nms_kernel
  const float ix1 = tmax(b.x, p.x), iy1 = tmax(b.y, p.y);
  const float ix2 = tmin(b.z, p.z), iy2 = tmin(b.w, p.w);
  float w = ix2 - ix1 + 1.0f, h = iy2 - iy1 + 1.0f;
  w = tmax(w, 0.0f);
  h = tmax(h, 0.0f);
  const float inter = w * h;
  const float iou = inter / (ab + ap - inter + 1e-7f);
  bool over = !(iou <= thr);
  if (over && diou) {
    const float ox1 = tmin(b.x, p.x), oy1 = tmin(b.y, p.y);
    const float ox2 = tmax(b.z, p.z), oy2 = tmax(b.w, p.w);
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

struct alignas(16) TailLds {  // fixed-size part of the LDS image (the arrays follow, see tail_lds_bytes)
  TailLevel lv[SSDK_MAX_LEVELS];
  u64 rows[64];
  u64 blk_dead;
  u32 nvalid, nk, nhead, pad1;
  u64 lbound[SSDK_MAX_LEVELS];  // per level: lower bound of its K-th key (phase B1)
  u32 hp[32];                   // phase C: head candidates of run r
  u64 hk[kHeadMax];             // phase C: the head candidates, gathered
};

__host__ __device__ inline size_t tail_lds_bytes(u32 units_per_image, u32 K, u32 L, u32 M, u32 ndet) {
  size_t n = sizeof(TailLds);
  n += (((size_t)units_per_image * K * 8) + 15) & ~(size_t)15;  // ukeys
  n += 2 * ((((size_t)units_per_image * 4) + 15) & ~(size_t)15);  // ucnt, uq
  n += 2 * (size_t)(M < 128u ? 128u : M) * 8;     // nkeys, sorted (whole runs of 128 keys)
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
  unsigned char* q = smem + sizeof(TailLds);
  u64* ukeys = reinterpret_cast<u64*>(q);
  q += (((size_t)upi * K * 8) + 15) & ~(size_t)15;
  u32* ucnt = reinterpret_cast<u32*>(q);
  q += (((size_t)upi * 4) + 15) & ~(size_t)15;
  u32* uq = reinterpret_cast<u32*>(q);  // keys of unit u that can still reach rank < K (a prefix of its list)
  q += (((size_t)upi * 4) + 15) & ~(size_t)15;
  const u32 Mp = M < 128u ? 128u : M;  // whole runs of 128 keys
  u64* nkeys = reinterpret_cast<u64*>(q);
  q += (size_t)Mp * 8;
  u64* sorted = reinterpret_cast<u64*>(q);
  q += (size_t)Mp * 8;
  float4* rec_box = reinterpret_cast<float4*>(q);
  q += (size_t)L * K * 16;
  float* rec_score = reinterpret_cast<float*>(q);
  float* rec_cls = rec_score + (size_t)L * K;
  q += (((size_t)L * K * 8) + 15) & ~(size_t)15;
  float4* kbox = reinterpret_cast<float4*>(q);
  q += (size_t)ndet * 16;
  float* karea = reinterpret_cast<float*>(q);
  float* kcls = karea + ndet;

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.x;
  const bool stamp = p.stamps != nullptr && b == 0 && tid == 0;
  if (stamp) p.stamps[0] = clock64();

  // ---- A: geometry + unit lists into LDS -----------------------------------------------------------------------
  {
    const u32* src = reinterpret_cast<const u32*>(&p.lv[0]);
    u32* dst = reinterpret_cast<u32*>(&S->lv[0]);
    for (u32 i = tid; i < (u32)(sizeof(TailLevel) / 4) * L; i += NT) dst[i] = src[i];
  }
  if (tid == 0) {
    S->blk_dead = 0ull;
    S->nvalid = 0;
    S->nk = 0;
    S->nhead = 0;
  }
  {
    const u64* src = p.cand + (size_t)b * upi * K;
    const u32 total = upi * K;
    for (u32 i = tid; i < total; i += NT) ukeys[i] = src[i];
    for (u32 i = tid; i < upi; i += NT) ucnt[i] = p.cand_cnt[(size_t)b * upi + i];
  }
  __syncthreads();
  // A cheap lower bound of every level's K-th key: with j = ceil(K / nruns), every full list of the level holds j keys
  // >= its own j-th, so K keys are >= the smallest of those j-th keys and nothing below it can reach rank < K.  (Lists
  // of a level are statistically alike: the bound discards ~(nruns-1)/nruns of the keys before any search.)
  if (tid < L) {
    const u32 u0 = S->lv[tid].unit_base, nruns = S->lv[tid].units;
    const u32 j = (K + nruns - 1) / nruns;
    u64 bound = ~0ull, single = 0ull;
    for (u32 v = u0; v < u0 + nruns; ++v) {
      const u64 kj = ukeys[(size_t)v * K + j - 1];  // (0 behind the end of a short list: no bound then)
      bound = kj < bound ? kj : bound;
      const u64 kk = ukeys[(size_t)v * K + K - 1];  // a full list alone proves K keys >= its last one (ties in index
      single = kk > single ? kk : single;           // order, e.g. an all-equal image: the first list IS the level's top K)
    }
    bound = bound > single ? bound : single;
    S->lbound[tid] = nruns > 1 ? bound : 0ull;
  }
  __syncthreads();
  // lists are sorted: the keys of a list at or above its level's bound are a PREFIX of it; uq[u] = its length
  if (tid < upi) {
    u32 l = 0;
    for (u32 t = 1; t < L; ++t)
      if (tid >= S->lv[t].unit_base) l = t;
    const u64 bound = S->lbound[l];
    const u64* lst = ukeys + (size_t)tid * K;
    u32 lo = 0, hi = ucnt[tid];
    while (lo < hi) {  // first index whose key is below the bound
      const u32 mid = (lo + hi) >> 1;
      if (lst[mid] >= bound) lo = mid + 1;
      else hi = mid;
    }
    uq[tid] = lo;
  }
  __syncthreads();
  if (stamp) p.stamps[1] = clock64();

  // ---- B: merged rank inside the level (B1), decode of the winners (B2) ----------------------------------------------
  const u32 LK = L * K;
  float* mid_s = p.mid_scores ? p.mid_scores + (size_t)b * LK : nullptr;
  float4* mid_b = p.mid_boxes ? reinterpret_cast<float4*>(p.mid_boxes) + (size_t)b * LK : nullptr;
  float* mid_c = p.mid_classes ? p.mid_classes + (size_t)b * LK : nullptr;
  u64* wl = sorted;  // work list of B2: wl[l*K + r] = the raw key of rank r in level l (0: no such candidate)
  {
    int steps = 1;  // binary-search steps that resolve a list of up to K keys
    while ((1u << steps) <= K) ++steps;
    for (u32 u = wave; u < upi; u += NW) {  // a unit per wave and trip: its surviving keys are its first uq[u]
      u32 l = 0;
      for (u32 v = 1; v < L; ++v)
        if (u >= S->lv[v].unit_base) l = v;
      const u32 u0 = S->lv[l].unit_base, nruns = S->lv[l].units, nq = uq[u];
      // Only the surviving PREFIXES of the sibling lists can hold keys above one of ours (everything behind a prefix is
      // below the level's bound, our keys are at or above it): the searches run over uq[], and not at all when no sibling
      // has a prefix -- an all-equal image (the reference-init network: every score ties, the first list of a level IS
      // its top K) then costs a wave 5 stores instead of 5 x nruns x 9 dependent LDS reads (36k -> 2k cycles).
      u32 others = 0;
      for (u32 v = u0; v < u0 + nruns; ++v) others += v != u ? uq[v] : 0u;  // wave-uniform
      for (u32 i = lane; i < nq; i += 64) {
        const u64 key = ukeys[(size_t)u * K + i];
        u32 rank = i;
        if (others) rank += count_greater_runs(ukeys + (size_t)u0 * K, K, uq + u0, 0, nruns, u - u0, key, steps);
        if (rank < K) wl[l * K + rank] = key;
      }
    }
    // slots without a winner (fewer than K candidates in the level) and the padding up to Mp
    for (u32 pos = tid; pos < Mp; pos += NT) {
      bool empty = pos >= LK;
      if (!empty) {
        const u32 l = pos / K, r = pos - l * K;
        u32 n = 0;
        for (u32 v = S->lv[l].unit_base; v < S->lv[l].unit_base + S->lv[l].units; ++v) n += ucnt[v];
        empty = r >= (n < K ? n : K);
        if (empty && mid_s) {
          mid_s[pos] = 0.0f;
          mid_b[pos] = make_float4(0.f, 0.f, 0.f, 0.f);
          mid_c[pos] = 0.0f;
        }
      }
      if (empty) {
        nkeys[pos] = 0ull;
        wl[pos] = 0ull;
      }
    }
  }
  if (stamp) p.stamps[21] = clock64();
  __syncthreads();
  if (stamp) p.stamps[22] = clock64();
  // B2: one winner per thread and trip (the 4 delta loads of a thread's winners are independent of each other)
  for (u32 pos = tid; pos < LK; pos += NT) {
    const u64 key = wl[pos];
    if (key == 0ull) continue;
    const u32 l = pos / K;
    float s, c;
    float4 bx;
    tail_decode_one(S->lv[l], p.dtype, p.rescore, b, key_index(key), key_score(key), &s, &bx, &c);
    rec_score[pos] = s;
    rec_box[pos] = bx;
    rec_cls[pos] = c;
    nkeys[pos] = (s > 0.0f) ? make_key(s, pos) : 0ull;  // box.py:496 (NaN drops out too)
    if (mid_s) {
      mid_s[pos] = s;
      mid_b[pos] = bx;
      mid_c[pos] = c;
    }
  }
  if (stamp) p.stamps[2] = clock64();

  // ---- C: order of the walk (box.py:505): wave-local sorted runs of 128 (ssdk_select.h), then ranks -- but only for
  // the head of the order: the walk usually stops (ndet survivors) after a few hundred candidates, and a rank costs
  // 8 random LDS reads per run.  With j = ceil(320 / runs), the runs that hold j candidates each have j keys >= their
  // j-th: keys at or above the smallest of those j-th keys (a prefix of the order, whatever its length) are ranked
  // now, the rest only if the walk ever gets there.
  __syncthreads();  // phase B's stores
  const u32 nruns_c = Mp >> 7;
  if (stamp) p.stamps[16] = clock64();
  wg_sort_runs128<NT>(nkeys, nruns_c);
  if (stamp) p.stamps[17] = clock64();
  __syncthreads();
  if (stamp) p.stamps[18] = clock64();
  u64 head_bound = ~0ull;
  {
    u32 j = (320u + nruns_c - 1) / nruns_c;
    j = j < 128u ? j : 128u;
    u32 full = 0;  // runs that hold at least j candidates
    for (u32 r = 0; r < nruns_c; ++r) {
      const u64 kj = nkeys[r * 128 + j - 1];
      if (kj != 0ull) {
        ++full;
        head_bound = kj < head_bound ? kj : head_bound;
      }
    }
    if (full * j < 128u) head_bound = 1ull;  // too few candidates for a head: rank everything now
  }
  if (stamp) p.stamps[19] = clock64();
  // the head candidates are a prefix of every sorted run: hp[r] = its length
  if (tid < nruns_c) {
    const u64* run = nkeys + tid * 128;
    u32 lo = 0, hi = 128;
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      if (run[mid] >= head_bound && run[mid] != 0ull) lo = mid + 1;
      else hi = mid;
    }
    S->hp[tid] = lo;
  }
  __syncthreads();
  u32 nh = 0;
  for (u32 r = 0; r < nruns_c; ++r) nh += S->hp[r];
  const bool small_head = nh <= kHeadMax;
  if (small_head) {
    // gather them, sort the (at most 4) runs of 128 they fill, rank among those: 3 sibling runs instead of 15
    const u32 hch = nh ? (nh + 127u) >> 7 : 1u;
    for (u32 r = wave; r < nruns_c; r += NW) {  // run r's head keys go behind those of the runs before it
      u32 off = 0;
      for (u32 q2 = 0; q2 < r; ++q2) off += S->hp[q2];
      const u32 n_r = S->hp[r];
      for (u32 i = lane; i < n_r; i += 64) S->hk[off + i] = nkeys[r * 128 + i];
    }
    for (u32 t = nh + tid; t < hch * 128u; t += NT) S->hk[t] = 0ull;
    __syncthreads();
    wg_sort_runs128<NT>(S->hk, hch);
    __syncthreads();
    (void)wg_rank_emit<NT>(S->hk, hch, 1ull, ~0ull, [&](u32 rank, u64 key) { sorted[rank] = key; });
  } else {
    (void)wg_rank_emit<NT>(nkeys, nruns_c, head_bound, ~0ull, [&](u32 rank, u64 key) { sorted[rank] = key; });
  }
  {
    if (stamp) p.stamps[20] = clock64();
    u32 nz = 0;
    for (u32 i = tid; i < Mp; i += NT) nz += nkeys[i] != 0ull ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
    if (lane == 0 && nz) atomicAdd(&S->nvalid, nz);
    if (tid == 0) S->nhead = nh;
  }
  __syncthreads();
  const u32 nvalid = S->nvalid;
  u32 nsorted = S->nhead;  // sorted[0 .. nsorted) is final
  if (stamp) p.stamps[3] = clock64();

  // ---- D: greedy walk, 64 candidates per block -------------------------------------------------------------------
  float* os = p.out_scores + (size_t)b * ndet;
  float4* ob = reinterpret_cast<float4*>(p.out_boxes) + (size_t)b * ndet;
  float* oc = p.out_classes + (size_t)b * ndet;
  const float thr = p.thr;
  const int diou = p.diou;
  u32 nk = 0;
  for (u32 base = 0; base < nvalid && nk < ndet; base += 64) {  // workgroup-uniform
    if (base + 64 > nsorted && nsorted < nvalid) {  // the walk outlived the ranked head: rank the rest (rare)
      (void)wg_rank_emit<NT>(nkeys, nruns_c, 1ull, head_bound, [&](u32 rank, u64 key) { sorted[rank] = key; });
      __syncthreads();
      nsorted = nvalid;
    }
    const u32 i = base + lane;
    const bool valid = i < nvalid;
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
      const bool sup = (cls == ck) && tail_suppressed_by(box, area, kbox[k], karea[k], thr, diou);
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
        row = __ballot(((m >> lane) & 1ull) && tail_suppressed_by(box, area, pj, aj, thr, diou));
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
  if (stamp) p.stamps[4] = clock64();

  // ---- E: zero padding (box.py:489-491) ------------------------------------------------------------------------
  for (u32 i = nk + tid; i < ndet; i += NT) {
    os[i] = 0.f;
    ob[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    oc[i] = 0.f;
  }
  if (stamp) p.stamps[5] = clock64();
}

// ------------------------------------------------------------------------------------------------------------------
// host side (called from ssdk_decode_nms, ssdk_nms.hip)
// ------------------------------------------------------------------------------------------------------------------
constexpr size_t kTailLdsMax = 160 * 1024;

static u32 tail_pow2(u32 n) {
  u32 m = 2;
  while (m < n) m <<= 1;
  return m;
}

// LDS bytes the fused tail needs for this geometry, or 0 when it cannot take it (then level_kernel + nms_kernel run)
size_t tail_fits(u32 units_per_image, int K, int L, int ndet) {
  if (K < 1 || L < 1 || ndet < 1) return 0;
  const u32 M = tail_pow2((u32)(L * K));
  if (M > 4096) return 0;
  const size_t need = tail_lds_bytes(units_per_image, (u32)K, (u32)L, M, (u32)ndet);
  return need <= kTailLdsMax ? need : 0;
}

int launch_tail(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const u32* units, const u32* unit_base,
                u32 units_per_image, const void* cand, const void* cand_cnt, float nms_thr, int ndet, int diou,
                float* os, float* ob, float* oc, float* ms, float* mb, float* mc, unsigned long long* stamps,
                hipStream_t stream) {
  const size_t lds = tail_fits(units_per_image, K, L, ndet);
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
    memcpy(p.lv[l].anchors, lv[l].anchors, sizeof(float) * 4 * lv[l].A);
  }
  p.L = L;
  p.dtype = dtype;
  p.rescore = rescore;
  p.units_per_image = units_per_image;
  p.K = (u32)K;
  p.M = tail_pow2((u32)(L * K));
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
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kTailLdsMax) != hipSuccess) {
      (void)hipGetLastError();
      set_error("decode_nms: cannot raise the dynamic LDS limit of tail_kernel");
      return SSDK_E_LAUNCH;
    }
    attr = true;
  }
  lds_poison(stream);
  hipLaunchKernelGGL(tail_kernel, dim3((unsigned)B), dim3(kTailThreads), lds, stream, p);
  return check_launch("tail_kernel");
}

}  // namespace ssdk
