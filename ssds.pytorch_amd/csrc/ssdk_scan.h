// ssdk_scan.h -- device code shared by the two scan kernels of the decode stage:
//   ssdk_decode.hip   scan_kernel<DT, PF>     every dtype / threshold: sampled cut + per-wave key buffers, TopK stream
//   ssdk_scan16.hip   scan16_kernel<DT, PF>   16-bit heads with a positive threshold (every BASELINE config): packed 16-bit
//                                             compares, candidate VECTORS buffered during the stream, keys extracted once
// Both replace box.py:435-446 (threshold + topk of one (image, level)) and fall back, unit by unit, to the same exact
// streaming top-K (unit_topk_stream below: ssdk_select.h's TopK stream).
#pragma once
#include "ssdk_common.h"
#include "ssdk_select.h"
#include "ssdk_decode.h"

namespace ssdk {

constexpr int kScanThreads = 256;
constexpr u32 kCap = 4096;         // LDS candidate slots per workgroup
static_assert(kCap == kStreamCap, "stream buffers are kStreamCap keys");

struct ScanLevel {
  const void* cls;
  u32 n;          // A*C*H*W scores per image
  u32 units;      // units per image for this level
  u32 unit_base;  // first unit id of this level inside an image
  u32 tpu;        // tiles per unit of this level
};
struct ScanParams {
  ScanLevel lv[SSDK_MAX_LEVELS];
  int L;
  u32 units_per_image, B;
  u32 K;
  float thr;
  u32 hist_base, hist_sh;  // histogram window of the seeding phase: bin = (ord(score) - hist_base) >> hist_sh
  int fast;                // 1: seeded barrier-free streaming first (SSDK_SCAN_FAST, default), 0: TopK stream only
  u32 thr16, inf16;        // scan16_kernel: smallest 16-bit pattern whose value is >= thr; pattern of +inf
  u64* cand;      // [B][units_per_image][K]
  u32* cand_cnt;  // [B][units_per_image]
  unsigned long long* stamps;  // optional (debug): shader-clock stamps of workgroup 0 at the phase boundaries
};

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

// ---- seeded, barrier-free streaming (the normal case) --------------------------------------------------------------
// The per-tile barrier + prune protocol below (TopK stream) is exact for any input but spends most of a unit's time in
// radix selects while the running cut is still low: with half of all scores above the threshold (SURVEY 8d's
// untrained-head distribution) a unit prunes 4-5 times.  So a unit first looks at a SAMPLE of its own tiles (8 tiles
// spread over the unit) through a 1024-bin LDS histogram of the score's leading bits, window [thr, 1.0]: the lower edge
// of the bin in which the sample's count from the top reaches K is a valid lower bound of the unit's K-th largest
// score (the sample alone already holds K scores at or above it).  With that cut each WAVE then streams its share of
// the unit on its own -- no barrier, no LDS atomic: a wave-uniform counter and a private 1024-key buffer -- and keeps
// every score >= cut: K * tiles / 8 keys per unit in expectation, which fit.  One exact select + sort at the end.
// Whenever that does not work out (a wave's buffer overflows: heavy ties at the cut such as an all-equal image, an
// unrepresentative sample) the unit falls back to the TopK stream, seeded with the same cut.  Both paths are exact.
struct FastCtl {  // LDS
  u32 wcount[kScanThreads / 64];
  u32 overflow, cutbin, cum, total;
  u32 cutord, nge, pad0, pad1;
};

// LDS image of scan_kernel: [buf: kCap keys][SelScratch][StreamCtl][FastCtl][ring: waves x PF x 1 KiB].  The K-key
// staging area `sel` of the selects is only used once the ring has drained and lives on top of it.  53.5 KB with
// PF = 4: three workgroups (12 waves) per CU.
__host__ __device__ inline size_t scan_fixed_bytes() {
  return ((size_t)kCap * 8 + sizeof(SelScratch) + sizeof(StreamCtl) + sizeof(FastCtl) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t scan_lds_bytes(u32 K, int pf) {
  const size_t ring = (size_t)(kScanThreads / 64) * pf * 1024, sel = (size_t)((K + 1) & ~1u) * 8;
  return scan_fixed_bytes() + (ring > sel ? ring : sel);
}

constexpr u32 kWaveCap = kCap / (kScanThreads / 64);  // keys per wave buffer (1024)
constexpr u32 kSample = 8;                             // sample tiles of the histogram phase
constexpr u32 kHistBins = 1024;

// one LDS add per (wave, distinct bin of the leading lane): heavy ties (an all-equal image puts every score of the
// wave into ONE bin) would otherwise serialise 64 same-address atomics per instruction
__device__ __forceinline__ void hist_add(u32* hist, bool pass, u32 bin) {
  const u64 m = __ballot(pass);
  if (m == 0ull) return;
  const u32 lead = (u32)__ffsll((long long)m) - 1u;
  const u32 b0 = (u32)__builtin_amdgcn_readlane((int)bin, (int)lead);
  const u64 same = __ballot(pass && bin == b0);
  if (lane_id() == lead) atomicAdd(&hist[b0], (u32)__popcll(same));
  if (pass && bin != b0) atomicAdd(&hist[bin], 1u);
}

__device__ __forceinline__ u32 hist_bin(u32 ord_score, u32 hbase, u32 hsh) {
  const u32 bin = (ord_score - hbase) >> hsh;
  return bin < kHistBins - 1 ? bin : kHistBins - 1;
}

// Prefetch ring through LDS.  Written as ordinary loads into registers, the ring of PF tiles ends every round with
// register copies that wait for ALL outstanding loads (vmcnt(0) at the loop's back edge): the pipeline drains every PF
// tiles and the stream runs at the latency of single requests.  Here every wave owns PF slots of 1 KiB in LDS; a tile
// is fetched straight into its slot (global_load_lds_dwordx4: 16 bytes per lane at slot + lane * 16) and taken out by
// the same lane with a ds_read_b128 behind a hand-counted s_waitcnt vmcnt(PF-1) -- PF requests per lane stay in flight
// from the first tile to the last.  (Both halves of the hand-off are one asm block, so the compiler can neither move
// the read above the wait nor add waits of its own.)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) const unsigned char glb_u8;

__device__ __forceinline__ void ring_issue(const void* g, unsigned char* slot) {
  __builtin_amdgcn_global_load_lds((glb_u8*)g, (lds_u8*)slot, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ u32x4 ring_take(const unsigned char* slot_lane) {
  u32x4 v;
  const u32 a = (u32)(size_t)(const lds_u8*)slot_lane;
  asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(N) : "memory");
  return v;
}
__device__ __forceinline__ void ring_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- one unit of one (image, level): geometry of its 16-byte vectors ---------------------------------------------------
// Image b of a level starts at byte b*n*ES; loads are 16-byte ALIGNED vectors: `head` elements of the first vector
// belong to the previous image and up to VEC-1 elements of the last one to the next image (or to the padding of the
// allocation -- an aligned 16-byte vector that holds one valid byte never crosses an allocation boundary); both are
// masked by the index test.  Tile t of the unit = vectors vec0 + t*256 .. +255, one per thread.
struct ScanUnit {
  const unsigned char* abase;  // 16-byte aligned address of the image's first vector
  u32 n;       // scores of the image in this level
  u32 head;    // elements of vector 0 that belong to the previous image
  u32 vlast;   // last vector holding an element of this image
  u32 vec0;    // first vector of the unit
  u32 ntiles;  // tiles of the unit (0: nothing to do)

  template <int VEC>
  __device__ __forceinline__ u32 first_index(u32 t, u32 tid) const {  // flat index of the lane's first element (huge: masked)
    const u32 vi = vec0 + t * kScanThreads + tid;
    return (t < ntiles && vi <= vlast) ? vi * VEC - head : 0xffff0000u;
  }
  // address of the lane's vector of tile t < ntiles as uniform base + 32-bit lane offset (the saddr form of global_load:
  // no 64-bit per-lane arithmetic), clamped to the image's last vector
  __device__ __forceinline__ const void* addr_in_unit(u32 t, u32 tid) const {
    const u32 rel = t * kScanThreads + tid, rel_last = vlast - vec0;
    return abase + (size_t)vec0 * 16 + (size_t)((rel < rel_last ? rel : rel_last) * 16u);
  }
  // branch-free address: clamped to the unit's last tile and the image's last vector (what lies outside is masked by index)
  __device__ __forceinline__ const void* addr(u32 t, u32 tid) const {
    const u32 vi = vec0 + (t < ntiles ? t : ntiles - 1u) * kScanThreads + tid;
    return abase + (size_t)(vi < vlast ? vi : vlast) * 16;
  }
};

template <int DT>
__device__ __forceinline__ ScanUnit make_scan_unit(const void* cls, u32 n, u32 b, u32 uu, u32 tpu) {
  constexpr int VEC = DType<DT>::vec, ES = DType<DT>::size;
  ScanUnit U;
  const unsigned char* base = (const unsigned char*)cls + (size_t)b * n * ES;
  U.n = n;
  U.head = (u32)((uintptr_t)base & 15u) / ES;
  U.abase = base - (size_t)U.head * ES;
  const u32 nvec = (U.head + n + VEC - 1) / VEC;  // vectors holding at least one element of this image (>= 1)
  U.vlast = nvec - 1u;
  U.vec0 = uu * tpu * kScanThreads;
  U.ntiles = 0;
  if (U.vec0 < nvec) {
    U.ntiles = (nvec - U.vec0 + kScanThreads - 1) / kScanThreads;
    if (U.ntiles > tpu) U.ntiles = tpu;
  }
  return U;
}

// Exact top-min(K, #scores >= cut0) of the unit, for every input (the fallback of both scan kernels): ssdk_select.h's TopK
// stream over the LDS ring, one barrier per tile.  Winners end up at the front of buf, unordered; returns their number.
// LDS: buf kCap keys, ss, ctl, stage = the ring (waves x PF x 1 KiB; `sel` lives on top of it once it has drained).
template <int DT, int PF>
__device__ __forceinline__ u32 unit_topk_stream(const ScanUnit& U, float cut0, u32 K, u64* buf, SelScratch* ss, StreamCtl* ctl,
                                                unsigned char* stage) {
  constexpr int NT = kScanThreads, VEC = DType<DT>::vec;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  u64* sel = reinterpret_cast<u64*>(stage);
  __syncthreads();  // whatever used buf / the ring before is done
  if (tid == 0) {
    ctl->cnt = 0;
    ctl->flag[0] = 0;
    ctl->flag[1] = 0;
  }
  __syncthreads();
  const u32 limit = kCap - NT * VEC;
  float cut = cut0;  // a valid lower bound of the unit's K-th score (or the threshold)
  u32 cut_idx = 0xffffffffu;
  unsigned char* ring = stage + (size_t)wave * PF * 1024;
  if (U.ntiles > 0) {
#pragma unroll
    for (int i = 0; i < PF; ++i) ring_issue(U.addr((u32)i, tid), ring + i * 1024);
    for (u32 t0 = 0; t0 < U.ntiles; t0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const u32 t = t0 + i;  // tiles past ntiles (the last round of a unit) run masked: no keys, one barrier
        const u32x4 v = ring_take<PF - 1>(ring + i * 1024 + lane * 16);
        ring_issue(U.addr(t + PF, tid), ring + i * 1024);
        scan_vec<DT>(v, U.first_index<VEC>(t, tid), U.n, cut, cut_idx, buf, ctl, limit, t);
        u64 T;
        if (stream_finish_tile<NT>(buf, sel, ss, ctl, t, K, &T)) {
          cut = key_score(T);
          cut_idx = key_index(T);
        }
      }
    }
    ring_drain();
  }
  return stream_finalize<NT>(buf, sel, ss, ctl, K);
}

}  // namespace ssdk
