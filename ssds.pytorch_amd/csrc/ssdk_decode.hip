// ssdk_decode.hip -- threshold + exact top-k + box decode + centre rescoring on gfx950.
//
// Replaces box.decode (reference ssds/modeling/layers/box.py:408-477), which the reference runs as
// ~25 ATen launches per (image, level) with nonzero() host syncs.  Here:
//
//   scan_kernel<DT>     ONE pass over the conf tensor (the HBM-bound part: 2 B/score in bf16).  Each
//                       workgroup streams a "unit" (tiles_per_unit x 256 x 16 B) of one (image, level)
//                       with 16-byte coalesced loads, 4 loads in flight per lane, and keeps the exact
//                       top-K of what it has seen in LDS (TopK stream, ssdk_select.h).  It emits <=K
//                       64-bit keys (score bits | ~flat index) per unit.
//   level_kernel        per (image, level): merges the units' keys (same streaming top-K), sorts the
//                       K winners, gathers their 4 deltas, applies delta2box (box.py:74-87) and the
//                       centre rescoring (box.py:464-471) in the reference's fp32 operation order and
//                       writes the zero-padded [B, top_n] outputs (box.py:430-432, 473-475).
//
// Algorithmic HBM bytes: the conf tensor once + 4 deltas per winner + outputs.
#include "ssdk_common.h"
#include "ssdk_select.h"

namespace ssdk {

constexpr int kScanThreads = 256;
constexpr u32 kCap = 4096;         // LDS candidate slots per workgroup
static_assert(kCap == kStreamCap, "stream buffers are kStreamCap keys");

struct ScanLevel {
  const void* cls;
  u32 n;          // A*C*H*W scores per image
  u32 units;      // units per image for this level
  u32 unit_base;  // first unit id of this level inside an image
  u32 pad;
};
struct ScanParams {
  ScanLevel lv[SSDK_MAX_LEVELS];
  int L;
  u32 units_per_image;
  u32 tiles_per_unit;
  u32 K;
  float thr;
  u64* cand;      // [B][units_per_image][K]
  u32* cand_cnt;  // [B][units_per_image]
};

struct LevelDesc {
  const void* box;
  int A, C, H, W, stride;
  u32 units, unit_base;
  float anchors[SSDK_MAX_ANCHORS * 4];
};
struct LevelParams {
  LevelDesc lv[SSDK_MAX_LEVELS];
  int L, dtype, rescore;
  u32 units_per_image, K;
  u32 out_stride;  // L*K
  const u64* cand;
  const u32* cand_cnt;
  float* scores;
  float* boxes;
  float* classes;
};

struct LevelLds {  // per-workgroup copy of the level geometry (per-lane indexed anchors live in LDS)
  const void* box;
  int A, C, H, W, stride;
  u32 units, unit_base, pad;
  float anchors[SSDK_MAX_ANCHORS * 4];
};

__host__ __device__ inline size_t lds_bytes_for(u32 K) {
  return (size_t)kCap * 8 + (size_t)((K + 1) & ~1u) * 8 + sizeof(SelScratch) + sizeof(StreamCtl) +
         sizeof(LevelLds);
}

// One 16-byte vector per lane: every element that beats the running cut (score, index) becomes a 64-bit key.
// The wave appends all of them with ONE LDS atomic: per-lane counts -> wave exclusive scan (shuffles) ->
// the last lane reserves the wave's range -> every lane writes its keys at base + prefix + local rank.
// (The first version did ballot + atomic per element slot: 8 dependent LDS-atomic round trips per tile.)
template <int DT, int E>
__device__ __forceinline__ void scan_flags(const u32x4& v, u32 idx0, u32 n, float cut, u32 cut_idx, u32& pmask,
                                           float (&sv)[DType<DT>::vec]) {
  if constexpr (E < DType<DT>::vec) {
    const float s = vec_elem<DT, E>(v);
    const u32 idx = idx0 + E;  // wraps to a huge value for the (masked) head elements
    const bool pass = (idx < n) & ((s > cut) | ((s == cut) & (idx < cut_idx)));
    pmask |= pass ? (1u << E) : 0u;
    sv[E] = s;
    scan_flags<DT, E + 1>(v, idx0, n, cut, cut_idx, pmask, sv);
  }
}

template <int DT>
__device__ __forceinline__ void scan_vec(const u32x4& v, u32 idx0, u32 n, float cut, u32 cut_idx, u64* buf,
                                         StreamCtl* ctl, u32 limit, u32 tile) {
  constexpr int VEC = DType<DT>::vec;
  u32 pmask = 0;
  float sv[VEC];
  scan_flags<DT, 0>(v, idx0, n, cut, cut_idx, pmask, sv);
  if (__ballot(pmask != 0u) == 0ull) return;  // nothing in this wave beats the cut (the common case later on)
  // exclusive prefix of the per-lane counts (0..8) without a shuffle chain: one ballot per count bit, the lanes
  // below me that have the bit set (mbcnt) weigh 2^bit.  (The first version ran a 6-step __shfl_up scan = six
  // dependent ds_bpermute round trips per 16-byte vector.)
  const u32 cnt = (u32)__popc(pmask);
  u32 excl = 0, total = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const u64 mb = __ballot((cnt >> b) & 1u);
    excl += mbcnt(mb) << b;
    total += (u32)__popcll(mb) << b;
  }
  u32 base = 0;
  if (lane_id() == 0) {
    base = atomicAdd(&ctl->cnt, total);
    if (base <= limit && base + total > limit) ctl->flag[tile & 1u] = tile + 1u;  // the unique crosser
  }
  base = (u32)__builtin_amdgcn_readfirstlane((int)base) + excl;
#pragma unroll
  for (int e = 0; e < VEC; ++e)
    if ((pmask >> e) & 1u) buf[base + (u32)__popc(pmask & ((1u << e) - 1u))] = make_key(sv[e], idx0 + (u32)e);
}

template <int DT, int PF>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(const ScanParams p) {
  constexpr int NT = kScanThreads;
  constexpr int VEC = DType<DT>::vec;
  constexpr int ES = DType<DT>::size;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* buf = reinterpret_cast<u64*>(smem);
  u64* sel = buf + kCap;
  SelScratch* ss = reinterpret_cast<SelScratch*>(sel + ((p.K + 1) & ~1u));
  StreamCtl* ctl = reinterpret_cast<StreamCtl*>(ss + 1);

  const u32 tid = threadIdx.x;
  const u32 b = blockIdx.x / p.units_per_image;
  const u32 u = blockIdx.x % p.units_per_image;
  u32 n = p.lv[0].n, ubase = 0;
  const void* cls = p.lv[0].cls;
#pragma unroll
  for (int i = 1; i < SSDK_MAX_LEVELS; ++i)
    if (i < p.L && u >= p.lv[i].unit_base) {
      n = p.lv[i].n;
      ubase = p.lv[i].unit_base;
      cls = p.lv[i].cls;
    }
  const u32 uu = u - ubase;

  // image b of this level starts at byte b*n*ES; loads are 16-byte ALIGNED vectors: `head` elements of the first
  // vector belong to the previous image and up to VEC-1 elements of the last one to the next image (or to the
  // padding of the allocation -- an aligned 16-byte vector that holds one valid byte never crosses an allocation
  // boundary); both are masked by the index test.
  const unsigned char* base = (const unsigned char*)cls + (size_t)b * n * ES;
  const u32 head = (u32)((uintptr_t)base & 15u) / ES;
  const unsigned char* abase = base - (size_t)head * ES;
  const u32 nvec = (head + n + VEC - 1) / VEC;   // vectors holding at least one element of this image (>= 1)
  const u32 vec0 = uu * p.tiles_per_unit * NT;   // first vector of this unit
  u32 ntiles = 0;
  if (vec0 < nvec) {
    ntiles = (nvec - vec0 + NT - 1) / NT;
    if (ntiles > p.tiles_per_unit) ntiles = p.tiles_per_unit;
  }

  if (tid == 0) {
    ctl->cnt = 0;
    ctl->flag[0] = 0;
    ctl->flag[1] = 0;
  }
  __syncthreads();

  const u32 K = p.K;
  const u32 limit = kCap - NT * VEC;
  float cut = p.thr;
  u32 cut_idx = 0xffffffffu;

  // Branch-free: the address is clamped to the image's last vector and what lies outside the unit is masked through
  // its index, so every tile issues exactly one load per lane and the compiler can count them (s_waitcnt vmcnt(PF-1)
  // before a tile is consumed).  (The first version guarded the load with the tile / tail tests: the waits at the
  // joins of those branches degenerated to vmcnt(0) right after the prefetch was issued -- one exposed memory
  // latency per 4 KB tile, 0.9 us, instead of PF tiles in flight.)
  const u32 vlast = nvec - 1u;
  // (a prefetch past the unit's last tile re-reads that tile -- cache hits -- instead of the next unit's data)
  auto load_vec = [&](u32 t) -> u32x4 {
    const u32 vi = vec0 + (t < ntiles ? t : ntiles - 1u) * NT + tid;
    return *reinterpret_cast<const u32x4*>(abase + (size_t)(vi < vlast ? vi : vlast) * 16);
  };

  u32x4 pf[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) pf[i] = load_vec(i);

  for (u32 t0 = 0; t0 < ntiles; t0 += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const u32 t = t0 + i;  // tiles past ntiles (the last round of a unit) run masked: no keys, one barrier
      const u32x4 v = pf[i];
      pf[i] = load_vec(t + PF);
      const u32 vi = vec0 + t * NT + tid;
      const u32 idx0 = (t < ntiles && vi <= vlast) ? vi * VEC - head : 0xffff0000u;
      scan_vec<DT>(v, idx0, n, cut, cut_idx, buf, ctl, limit, t);
      u64 T;
      if (stream_finish_tile<NT>(buf, sel, ss, ctl, t, K, &T)) {
        cut = key_score(T);
        cut_idx = key_index(T);
      }
    }
  }

  const u32 cnt = stream_finalize<NT>(buf, sel, ss, ctl, K);
  u64* out = p.cand + ((size_t)b * p.units_per_image + u) * K;
  for (u32 i = tid; i < K; i += NT) out[i] = (i < cnt) ? buf[i] : 0ull;
  if (tid == 0) p.cand_cnt[(size_t)b * p.units_per_image + u] = cnt;
}

// box.py:74-87 delta2box + box.py:459-471 for one winner; fp32, reference operation order.
__device__ __forceinline__ void decode_one(const LevelLds& d, int dtype, int rescore, u32 b, u32 idx,
                                           float score, float* o_score, float* o_box, float* o_cls) {
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
  o_box[0] = x1;
  o_box[1] = y1;
  o_box[2] = x2;
  o_box[3] = y2;
  *o_cls = (float)c;
}

constexpr int kLevelThreads = 256;

__global__ __launch_bounds__(kLevelThreads) void level_kernel(const LevelParams p) {
  constexpr int NT = kLevelThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* buf = reinterpret_cast<u64*>(smem);
  u64* sel = buf + kCap;
  SelScratch* ss = reinterpret_cast<SelScratch*>(sel + ((p.K + 1) & ~1u));
  StreamCtl* ctl = reinterpret_cast<StreamCtl*>(ss + 1);

  LevelLds* dl = reinterpret_cast<LevelLds*>(ctl + 1);

  const u32 tid = threadIdx.x;
  const u32 l = blockIdx.x, b = blockIdx.y;
#pragma unroll
  for (int i = 0; i < SSDK_MAX_LEVELS; ++i)
    if (i == (int)l) {
      if (tid == 0) {
        dl->box = p.lv[i].box;
        dl->A = p.lv[i].A;
        dl->C = p.lv[i].C;
        dl->H = p.lv[i].H;
        dl->W = p.lv[i].W;
        dl->stride = p.lv[i].stride;
        dl->units = p.lv[i].units;
        dl->unit_base = p.lv[i].unit_base;
      }
      if (tid < SSDK_MAX_ANCHORS * 4) dl->anchors[tid] = p.lv[i].anchors[tid];
    }
  __syncthreads();
  const LevelLds& d = *dl;
  const u32 K = p.K;
  const u64* src = p.cand + ((size_t)b * p.units_per_image + d.unit_base) * K;
  u32 n;
  if (d.units == 1) {  // the scan kernel already produced the exact top-K of this (image, level)
    n = p.cand_cnt[(size_t)b * p.units_per_image + d.unit_base];
    for (u32 i = tid; i < n; i += NT) buf[i] = src[i];
    __syncthreads();
  } else {
    if (tid == 0) {
      ctl->cnt = 0;
      ctl->flag[0] = 0;
      ctl->flag[1] = 0;
    }
    __syncthreads();
    const u32 total = d.units * K;
    constexpr u32 PER = 4, TILE = NT * PER;
    const u32 limit = kCap - TILE;
    u64 cutkey = 0;
    const u32 ntiles = (total + TILE - 1) / TILE;
    for (u32 t = 0; t < ntiles; ++t) {
      u64 k[PER];
#pragma unroll
      for (u32 j = 0; j < PER; ++j) {
        const u32 i = t * TILE + j * NT + tid;
        k[j] = (i < total) ? src[i] : 0ull;
      }
#pragma unroll
      for (u32 j = 0; j < PER; ++j) stream_append(buf, ctl, limit, t, k[j] > cutkey, k[j]);
      u64 T;
      if (stream_finish_tile<NT>(buf, sel, ss, ctl, t, K, &T)) cutkey = T;
    }
    n = stream_finalize<NT>(buf, sel, ss, ctl, K);
    __syncthreads();
  }

  // sort the winners: descending key == (score desc, index asc)  (torch.topk sorted=True, box.py:446)
  u32 M = 2;
  while (M < n) M <<= 1;
  for (u32 i = n + tid; i < M; i += NT) buf[i] = 0ull;
  __syncthreads();
  wg_bitonic_sort_desc<NT>(buf, M);

  float* so = p.scores + (size_t)b * p.out_stride + (size_t)l * K;
  float* bo = p.boxes + ((size_t)b * p.out_stride + (size_t)l * K) * 4;
  float* co = p.classes + (size_t)b * p.out_stride + (size_t)l * K;
  for (u32 i = tid; i < K; i += NT) {
    float s = 0.f, c = 0.f, bx[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      const u64 key = buf[i];
      decode_one(d, p.dtype, p.rescore, b, key_index(key), key_score(key), &s, bx, &c);
    }
    so[i] = s;
    co[i] = c;
    *reinterpret_cast<float4*>(bo + (size_t)i * 4) = make_float4(bx[0], bx[1], bx[2], bx[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
hipEvent_t* g_prof_events = nullptr;  // [0] before scan, [1] after scan, [2] after level

// Tail stream of the decode stage (ssdk_set_decode_tail_stream): level_kernel and nms_kernel are latency-bound work
// on 64-384 workgroups; on their own stream they run under the next batch's forward pass instead of in front of it.
// scan_kernel (the chip-filling, HBM-bound pass) stays on the caller's stream.
thread_local hipStream_t g_tail_stream = nullptr;
static hipEvent_t g_tail_fork[8];
static bool g_tail_fork_ready = false;
static unsigned g_tail_fork_i = 0;

// everything enqueued on `from` so far happens before what is enqueued on `to` from now on
int stream_fork(hipStream_t from, hipStream_t to) {
  if (!g_tail_fork_ready) {
    for (auto& e : g_tail_fork)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        set_error("decode: hipEventCreate failed");
        return SSDK_E_LAUNCH;
      }
    g_tail_fork_ready = true;
  }
  hipEvent_t e = g_tail_fork[g_tail_fork_i++ & 7u];
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
    set_error("decode: stream fork failed");
    return SSDK_E_LAUNCH;
  }
  return SSDK_OK;
}

struct DecodePlan {
  u32 tiles_per_unit, units_per_image;
  u32 units[SSDK_MAX_LEVELS], unit_base[SSDK_MAX_LEVELS], n[SSDK_MAX_LEVELS];
  size_t cand_bytes, cnt_bytes;
};

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

static int make_plan(const ssdk_level* lv, int L, int B, int dtype, int K, DecodePlan* pl) {
  if (!lv || L < 1 || L > SSDK_MAX_LEVELS || B < 1) {
    set_error("decode: need 1 <= L <= %d levels and B >= 1 (L=%d, B=%d)", SSDK_MAX_LEVELS, L, B);
    return SSDK_E_BADARG;
  }
  if (K < 1 || K > SSDK_MAX_TOPN) {
    set_error("decode: top_n=%d outside [1, %d]", K, SSDK_MAX_TOPN);
    return SSDK_E_BADARG;
  }
  if (dtype != SSDK_F32 && dtype != SSDK_BF16 && dtype != SSDK_F16) {
    set_error("decode: unknown dtype %d", dtype);
    return SSDK_E_BADARG;
  }
  const u32 vec = dtype == SSDK_F32 ? 4 : 8;
  const u32 tile = kScanThreads * vec;
  unsigned long long tiles_total = 0;
  for (int l = 0; l < L; ++l) {
    const ssdk_level& v = lv[l];
    if (v.A < 1 || v.A > SSDK_MAX_ANCHORS || v.C < 1 || v.H < 1 || v.W < 1) {
      set_error("decode: level %d has bad dims A=%d C=%d H=%d W=%d", l, v.A, v.C, v.H, v.W);
      return SSDK_E_BADARG;
    }
    const unsigned long long n = (unsigned long long)v.A * v.C * v.H * v.W;
    if (n >= (1ull << 31)) {
      set_error("decode: level %d has %llu scores per image (limit 2^31)", l, n);
      return SSDK_E_BADARG;
    }
    pl->n[l] = (u32)n;
    tiles_total += (n + vec + tile - 1) / tile;
  }
  // unit size: enough workgroups to keep 256 CUs streaming, but units as large as possible so that the per-unit
  // prunes / final select and the K keys written per unit (and merged again by level_kernel) amortise.  Measured on
  // SSD-MobileNetV2@512 batch 64: 32 tiles per unit (640 workgroups) 45 us, 21 tiles (1024 workgroups) 55 us.
  int tpu = env_int("SSDK_TILES_PER_UNIT", 0);
  if (tpu <= 0) {
    const unsigned long long target_wgs = (unsigned long long)env_int("SSDK_TARGET_WGS", 640);
    unsigned long long t = (tiles_total * (unsigned long long)B + target_wgs - 1) / target_wgs;
    tpu = (int)(t < 4 ? 4 : (t > 64 ? 64 : t));
  }
  pl->tiles_per_unit = (u32)tpu;
  u32 base = 0;
  for (int l = 0; l < L; ++l) {
    const u32 tiles = (u32)(((unsigned long long)pl->n[l] + vec + tile - 1) / tile);
    pl->units[l] = (tiles + tpu - 1) / tpu;
    pl->unit_base[l] = base;
    base += pl->units[l];
  }
  pl->units_per_image = base;
  pl->cand_bytes = (size_t)B * base * K * sizeof(u64);
  pl->cnt_bytes = (((size_t)B * base * sizeof(u32)) + 255) & ~(size_t)255;
  return SSDK_OK;
}

static int launch_decode(const ssdk_level* lv, int L, int B, int dtype, float thr, int K, int rescore,
                         float* scores, float* boxes, float* classes, void* ws, size_t ws_bytes,
                         hipStream_t stream, hipStream_t tail = nullptr) {
  DecodePlan pl;
  int rc = make_plan(lv, L, B, dtype, K, &pl);
  if (rc) return rc;
  if (!scores || !boxes || !classes) {
    set_error("decode: null output pointer");
    return SSDK_E_BADARG;
  }
  if (!ws || ws_bytes < pl.cand_bytes + pl.cnt_bytes || ((uintptr_t)ws & 15)) {
    set_error("decode: workspace too small or misaligned (%zu < %zu)", ws_bytes, pl.cand_bytes + pl.cnt_bytes);
    return SSDK_E_WORKSPACE;
  }
  ScanParams sp;
  LevelParams lp;
  memset(&sp, 0, sizeof(sp));
  memset(&lp, 0, sizeof(lp));
  for (int l = 0; l < L; ++l) {
    if (!lv[l].cls || !lv[l].box || ((uintptr_t)lv[l].cls & 15)) {
      set_error("decode: level %d has a null or non-16-byte-aligned head pointer", l);
      return SSDK_E_BADARG;
    }
    sp.lv[l].cls = lv[l].cls;
    sp.lv[l].n = pl.n[l];
    sp.lv[l].units = pl.units[l];
    sp.lv[l].unit_base = pl.unit_base[l];
    lp.lv[l].box = lv[l].box;
    lp.lv[l].A = lv[l].A;
    lp.lv[l].C = lv[l].C;
    lp.lv[l].H = lv[l].H;
    lp.lv[l].W = lv[l].W;
    lp.lv[l].stride = lv[l].stride;
    lp.lv[l].units = pl.units[l];
    lp.lv[l].unit_base = pl.unit_base[l];
    memcpy(lp.lv[l].anchors, lv[l].anchors, sizeof(float) * 4 * lv[l].A);
  }
  sp.L = L;
  sp.units_per_image = pl.units_per_image;
  sp.tiles_per_unit = pl.tiles_per_unit;
  sp.K = (u32)K;
  sp.thr = thr;
  sp.cand = (u64*)ws;
  sp.cand_cnt = (u32*)((char*)ws + pl.cand_bytes);
  lp.L = L;
  lp.dtype = dtype;
  lp.rescore = rescore;
  lp.units_per_image = pl.units_per_image;
  lp.K = (u32)K;
  lp.out_stride = (u32)(L * K);
  lp.cand = sp.cand;
  lp.cand_cnt = sp.cand_cnt;
  lp.scores = scores;
  lp.boxes = boxes;
  lp.classes = classes;

  const size_t lds = lds_bytes_for((u32)K);
  const dim3 grid((unsigned)(B * pl.units_per_image));
  lds_poison(stream);
  if (g_prof_events) (void)hipEventRecord(g_prof_events[0], stream);
  static const int pf = [] {  // 16-byte loads in flight per lane: 4 (default) or 8 (SSDK_SCAN_PF=8)
    const char* e = getenv("SSDK_SCAN_PF");
    return (e && atoi(e) == 8) ? 8 : 4;
  }();
  if (pf == 8) {
    if (dtype == SSDK_F32) hipLaunchKernelGGL((scan_kernel<SSDK_F32, 8>), grid, dim3(kScanThreads), lds, stream, sp);
    else if (dtype == SSDK_BF16) hipLaunchKernelGGL((scan_kernel<SSDK_BF16, 8>), grid, dim3(kScanThreads), lds, stream, sp);
    else hipLaunchKernelGGL((scan_kernel<SSDK_F16, 8>), grid, dim3(kScanThreads), lds, stream, sp);
  } else {
    if (dtype == SSDK_F32) hipLaunchKernelGGL((scan_kernel<SSDK_F32, 4>), grid, dim3(kScanThreads), lds, stream, sp);
    else if (dtype == SSDK_BF16) hipLaunchKernelGGL((scan_kernel<SSDK_BF16, 4>), grid, dim3(kScanThreads), lds, stream, sp);
    else hipLaunchKernelGGL((scan_kernel<SSDK_F16, 4>), grid, dim3(kScanThreads), lds, stream, sp);
  }
  rc = check_launch("scan_kernel");
  if (rc) return rc;
  if (g_prof_events) (void)hipEventRecord(g_prof_events[1], stream);
  hipStream_t st2 = stream;
  if (tail && tail != stream) {  // the rest of the stage goes to the tail stream, ordered after the scan
    rc = stream_fork(stream, tail);
    if (rc) return rc;
    st2 = tail;
  }
  lds_poison(st2);
  hipLaunchKernelGGL(level_kernel, dim3((unsigned)L, (unsigned)B), dim3(kLevelThreads), lds, st2, lp);
  rc = check_launch("level_kernel");
  if (g_prof_events) (void)hipEventRecord(g_prof_events[2], st2);
  return rc;
}

// shared with ssdk_nms.hip (fused decode_nms entry point lives there)
size_t decode_ws_bytes(const ssdk_level* lv, int L, int B, int dtype, int K) {
  DecodePlan pl;
  if (make_plan(lv, L, B, dtype, K, &pl)) return 0;
  return pl.cand_bytes + pl.cnt_bytes;
}
int decode_levels(const ssdk_level* lv, int L, int B, int dtype, float thr, int K, int rescore,
                  float* scores, float* boxes, float* classes, void* ws, size_t ws_bytes, void* stream,
                  void* tail) {
  return launch_decode(lv, L, B, dtype, thr, K, rescore, scores, boxes, classes, ws, ws_bytes,
                       (hipStream_t)stream, (hipStream_t)tail);
}

}  // namespace ssdk

extern "C" size_t ssdk_decode_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype, int top_n) {
  return ssdk::decode_ws_bytes(levels, L, B, dtype, top_n);
}

extern "C" int ssdk_decode(const ssdk_level* level, int B, int dtype, float threshold, int top_n,
                           int rescore, float* scores, float* boxes, float* classes, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return ssdk::decode_levels(level, 1, B, dtype, threshold, top_n, rescore, scores, boxes, classes,
                             workspace, workspace_bytes, stream, nullptr);
}

extern "C" int ssdk_set_decode_tail_stream(void* stream) {
  ssdk::g_tail_stream = (hipStream_t)stream;
  return SSDK_OK;
}
