// ssdk_scan16.hip -- threshold + exact top-K of one unit of one (image, level) for 16-bit heads (bf16 / f16) and a
// positive threshold: what every BASELINE configuration runs.  Replaces box.py:435-446 like scan_kernel (ssdk_decode.hip);
// same unit geometry, same workspace contract (<= K keys per unit -- here UNORDERED -- plus their count), same exact
// fallback.  What is different, and why (round-2 counters: 58 VALU instructions per 16-byte vector, 47 % of wave cycles
// parked, sample tiles fetched twice):
//
//   * scores are compared as packed 16-bit INTEGERS.  For positive finite bf16 / f16 values the bit pattern orders like
//     the value, negative values are negative integers (below every positive cut) and positive NaNs are the patterns above
//     +inf, so "does this 16-byte vector hold a score >= cut" is three v_pk_max_i16 over its four dwords, one more against
//     (cut-1, cut-1) and one compare: 5 VALU instructions per vector.
//   * the stream buffers candidate VECTORS, not keys: one ballot, one ds_write_b128 + one ds_write_b16 (the vector's number
//     inside the unit) for the lanes that have anything.  Keys (score bits | ~flat index) are built once, behind the
//     stream, from the few hundred buffered vectors -- with the exact tests of the reference there (index inside the image,
//     fp32 compare semantics: NaN never passes, box.py:440).
//   * the cut comes from a sample of the unit's OWN tiles which are then NOT streamed again: the eight sample vectors of
//     a lane stay in registers, give the cut (top-2 per 16-bit half-lane by packed min / max, a 1024-bin histogram of those
//     1024 values, the bin in which their count from the top reaches K) and are filtered from the registers.  Every byte of
//     the conf tensor is fetched once.
//   * the unit's winners leave unordered: the consumer (levelsel_kernel) selects per LEVEL with a histogram anyway and no
//     longer merges sorted runs, so the per-unit sort (9 k cycles of 83 k) is gone and the per-unit select is one
//     histogram pass over the extracted keys.
//
// Ties at the cut (the reference-init network: every score of a level is the same bf16 value) keep the round-2 rule: of
// the scores EQUAL to the cut only the first K in index order can be winners; they are collected by a short prefix walk
// and the stream then looks for cut + 1.  A unit whose sample misleads (buffers overflow) or holds a NaN among its
// largest patterns runs the exact TopK stream instead (unit_topk_stream, ssdk_scan.h).
//
// Round 5 (what a sample cannot promise): the extracted keys share ONE buffer behind an LDS cursor (a segment per
// wave overflowed when the candidates clustered in two lane quarters); the cut comes from the sample rank that predicts
// sqrt(K x capacity) candidates, with the PROVEN cut (rank K) kept for the rerun when the list comes back short; a unit
// whose cut value is suspiciously frequent among the maxima counts its sample registers exactly and raises the cut (proven
// while K sample elements lie above it, predicted while they promise 2 K in the unit) or settles the cut value as a tie.
#include "ssdk_scan.h"

namespace ssdk {

constexpr u32 kVq = 128;   // slots of a wave's circular queue of candidate vectors (a round of 64 is extracted as soon as it is full)
constexpr u32 kSeg = 512;  // keys per wave segment of the key buffer
constexpr u32 kNearTie = 64;   // occurrences of the cut value among the 1024 sample maxima that make a unit count its sample exactly
constexpr u32 kNearCap = 1536; // predicted keys of a unit above which a segment (4 x kSeg, uneven over the waves) is expected to overflow
// REG (template parameter of the kernel, SSDK_SCAN_REG = 8 | 16 | 32): tiles of a unit whose vectors are the SAMPLE and stay
// in registers (4 VGPRs each); the rest of the unit streams through the LDS ring behind the cut

struct S16Ctl {  // LDS
  u32 wcount[kScanThreads / 64];
  u32 cutbin, above, cb, ngt;    // ngt / neq: exact counts over the sample registers (near-tie units)
  u32 cut16, eq, nan, ovf;
  u32 kcnt, ocnt, scnt, neq;
  u32 cutbin_k, pad0, pad1, pad2;  // bin of the K-th sample maximum (the PROVEN cut) when cutbin is that of a smaller rank
  u32 sub[32];
};

// LDS image: [X: kCap keys = 32 KiB of the fallback | kbuf 4 x 512 keys, tiebuf 512 keys, vq 4 x 128 x (16 + 2) B]
//            [SelScratch][StreamCtl][S16Ctl][ring: waves x PF x 1 KiB | sel of the fallback]
__host__ __device__ inline size_t scan16_fixed_bytes() {
  return ((size_t)kCap * 8 + sizeof(SelScratch) + sizeof(StreamCtl) + sizeof(S16Ctl) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t scan16_lds_bytes(int pf) {
  return scan16_fixed_bytes() + (size_t)(kScanThreads / 64) * pf * 1024;  // (>= K keys of `sel`: K <= 512)
}
static_assert((size_t)(kScanThreads / 64) * (kSeg * 8 + kVq * 18) + 512 * 8 <= (size_t)kCap * 8, "the fast path's buffers must fit the key buffer");

__device__ __forceinline__ u32 pk_max_i16(u32 a, u32 b) {
  u32 r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ u32 pk_min_i16(u32 a, u32 b) {
  u32 r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// LDS-only barrier: __syncthreads() would also drain the global loads the ring keeps in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int DT>
__device__ __forceinline__ float f32_of16(u32 h) {
  if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
  else return f16_bits_to_f32(h);
}
template <int DT>
__device__ __forceinline__ u32 ord_of16(u32 h) {  // ord_f32 of a NON-NEGATIVE 16-bit score
  if constexpr (DT == SSDK_BF16) return (h << 16) | 0x80000000u;
  else return __builtin_bit_cast(u32, f16_bits_to_f32(h)) | 0x80000000u;
}
// smallest non-negative 16-bit pattern whose value is >= the positive float with ordered bits `o`
template <int DT>
__device__ __forceinline__ u32 ceil16_of_ord(u32 o) {
  const u32 bits = o & 0x7fffffffu;
  if constexpr (DT == SSDK_BF16) {
    return (bits >> 16) + ((bits & 0xffffu) ? 1u : 0u);
  } else {
    const float f = __builtin_bit_cast(float, bits);
    _Float16 h = (_Float16)f;  // round to nearest even
    u32 hb = (u32)__builtin_bit_cast(u16, h);
    if ((float)h < f) hb += 1u;  // (positive: the next pattern is the next value up; max finite + 1 = inf)
    return hb;
  }
}

__device__ __forceinline__ u32 half_of(const u32x4& v, int e) {  // compile-time e
  const u32 w = v[e >> 1];
  return (e & 1) ? (w >> 16) : (w & 0xffffu);
}

// zero the halves of a vector whose elements lie outside the image (head of the first vector, tail of the last one)
__device__ __forceinline__ u32x4 mask_outside(const u32x4& v, u32 idx0, u32 n) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const u32 lo = (idx0 + 2u * d < n) ? 0x0000ffffu : 0u;
    const u32 hi = (idx0 + 2u * d + 1u < n) ? 0xffff0000u : 0u;
    r[d] = v[d] & (lo | hi);
  }
  return r;
}

// does the vector hold a 16-bit pattern >= cut (signed compare)?  cm1 = (cut-1) in both halves
__device__ __forceinline__ bool any_ge(const u32x4& v, u32 cm1) {
  const u32 mm = pk_max_i16(pk_max_i16(v[0], v[1]), pk_max_i16(v[2], v[3]));
  return pk_max_i16(mm, cm1) != cm1;
}

// the lanes whose vector passes append it (and its number inside the unit) to the wave's circular queue, in lane order
__device__ __forceinline__ void append_vec(const u32x4& v, bool pass, u32 lid, u32x4* vq, u16* iq, u32& wcnt) {
  const u64 any = __ballot(pass);
  if (any == 0ull) return;
  if (pass) {
    const u32 pos = (wcnt + mbcnt(any)) & (kVq - 1u);
    vq[pos] = v;
    iq[pos] = (u16)lid;
  }
  wcnt += (u32)__popcll(any);
}

// K-th bin from the top of ss->hist (kHistBins bins): sc->cutbin / above / cb, or cutbin stays ~0 when fewer than K
// values were counted.  Ends with a barrier.
__device__ __forceinline__ void hist_kth_bin(SelScratch* ss, S16Ctl* sc, u32 K) {
  constexpr int NT = kScanThreads, BPT = kHistBins / NT;
  const u32 tid = threadIdx.x;
  u32 local = 0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) local += ss->hist[tid * BPT + j];
  const u32 incl = wg_incl_suffix_sum<NT>(local, ss->wsum);
  const u32 excl = incl - local;
  if (excl < K && K <= incl) {  // at most one thread
    u32 acc = excl;
    for (int j = BPT - 1; j >= 0; --j) {
      const u32 h = ss->hist[tid * BPT + j];
      if (acc + h >= K) {
        sc->cutbin = tid * BPT + j;
        sc->above = acc;
        sc->cb = h;
        break;
      }
      acc += h;
    }
  }
  __syncthreads();
}

// The same for two ranks R <= K in one pass: sc->cutbin / above / cb describe rank R, sc->cutbin_k is the bin of rank K (~0
// when fewer values were counted).  Ends with a barrier.
__device__ __forceinline__ void hist_two_ranks(SelScratch* ss, S16Ctl* sc, u32 R, u32 K) {
  constexpr int NT = kScanThreads, BPT = kHistBins / NT;
  const u32 tid = threadIdx.x;
  u32 local = 0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) local += ss->hist[tid * BPT + j];
  const u32 incl = wg_incl_suffix_sum<NT>(local, ss->wsum);
  const u32 excl = incl - local;
  if (excl < R && R <= incl) {  // at most one thread
    u32 acc = excl;
    for (int j = BPT - 1; j >= 0; --j) {
      const u32 h = ss->hist[tid * BPT + j];
      if (acc + h >= R) {
        sc->cutbin = tid * BPT + j;
        sc->above = acc;
        sc->cb = h;
        break;
      }
      acc += h;
    }
  }
  if (excl < K && K <= incl) {  // at most one thread
    u32 acc = excl;
    for (int j = BPT - 1; j >= 0; --j) {
      acc += ss->hist[tid * BPT + j];
      if (acc >= K) {
        sc->cutbin_k = tid * BPT + j;
        break;
      }
    }
  }
  __syncthreads();
}

template <int DT, int PF, int REG>
__global__ __launch_bounds__(kScanThreads, 3) void scan16_kernel(const ScanParams p) {
  constexpr int NT = kScanThreads;
  constexpr u32 NW = NT / 64;
  constexpr u32 kReg = (u32)REG;
  constexpr u32 ULP_SH = DT == SSDK_BF16 ? 16u : 13u;  // ordered-fp32 bits below one ulp of the head's dtype
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* buf = reinterpret_cast<u64*>(smem);  // the fallback's key buffer; the fast path's vector buffers live on it
  u64* kbuf = reinterpret_cast<u64*>(smem);                                 // [NW * kSeg] extracted keys (one cursor: sc->kcnt)
  u64* tiebuf = kbuf + (size_t)NW * kSeg;                                   // [512] the unit's tie keys (tie-rich units)
  u32x4* vqb = reinterpret_cast<u32x4*>(tiebuf + 512);                      // [NW][kVq] candidate vectors waiting for extraction
  u16* iqb = reinterpret_cast<u16*>(smem + ((size_t)NW * kSeg + 512) * 8 + (size_t)NW * kVq * 16);  // [NW][kVq] their numbers
  SelScratch* ss = reinterpret_cast<SelScratch*>(buf + kCap);
  StreamCtl* ctl = reinterpret_cast<StreamCtl*>(ss + 1);
  S16Ctl* sc = reinterpret_cast<S16Ctl*>(ctl + 1);
  unsigned char* stage = smem + scan16_fixed_bytes();  // [wave][PF][1 KiB]

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 u = blockIdx.x / p.B;  // unit-major block order (see scan_kernel)
  const u32 b = blockIdx.x % p.B;
  u32 n = p.lv[0].n, ubase = 0, tpu = p.lv[0].tpu;
  const void* cls = p.lv[0].cls;
#pragma unroll
  for (int i = 1; i < SSDK_MAX_LEVELS; ++i)
    if (i < p.L && u >= p.lv[i].unit_base) {
      n = p.lv[i].n;
      ubase = p.lv[i].unit_base;
      cls = p.lv[i].cls;
      tpu = p.lv[i].tpu;
    }
  const ScanUnit U = make_scan_unit<DT>(cls, n, b, u - ubase, tpu);
  const u32 K = p.K;
  u64* out = p.cand + ((size_t)b * p.units_per_image + u) * K;
  u32* out_cnt = p.cand_cnt + (size_t)b * p.units_per_image + u;
  const bool stamp = p.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) p.stamps[0] = clock64();
  const bool wall = p.stamps != nullptr && blockIdx.x < 4096u && tid == 0;  // (p.stamps = ctx stamps + 24: timeline at +24)
  if (wall) p.stamps[24 + 2 * blockIdx.x] = wall_clock64();
  const u32 ntiles = U.ntiles;
  if (ntiles == 0) {  // (a plan never produces an empty unit; kept for forced unit sizes)
    for (u32 i = tid; i < K; i += NT) out[i] = 0ull;
    if (tid == 0) *out_cnt = 0;
    return;
  }

  // ---- sample tiles: S <= REG tiles spread over the unit, one 16-byte vector per lane and tile, kept in registers: they
  // give the cut (the K-th largest of the sample sits at rank ~K * tiles / S of the unit) and are filtered from the
  // registers afterwards, so no byte is fetched twice; the other tiles stream through the LDS ring.
  const u32 S = ntiles < kReg ? ntiles : kReg;
  const u32 sstride = ntiles / S;
  u32x4 sv[kReg];
#pragma unroll
  for (u32 i = 0; i < kReg; ++i)
    if (i < S) sv[i] = *reinterpret_cast<const u32x4*>(U.addr_in_unit(i * sstride, tid));  // wave-uniform predicate
    else sv[i] = u32x4{0u, 0u, 0u, 0u};

  // ---- the stream's tiles: everything that is not a sample tile -----------------------------------------------------------
  // iterator over the non-sample tiles: (t, ns) = current tile, next sample tile at or after it (~0: none left)
  const u32 M = ntiles - S;
  const u32 send = S * sstride;
  auto adv = [&](u32& t, u32& ns) {
    ++t;
    if (t == ns) {
      ++t;
      ns += sstride;
      if (ns >= send) ns = ~0u;
    }
  };
  u32 ti = sstride == 1u ? S : 1u, nsi = sstride == 1u ? ~0u : sstride;  // issue side
  u32 tc = ti, nsc = nsi;                                                   // consume side
  // ---- phase H: the cut ------------------------------------------------------------------------------------------------
  for (u32 i = tid; i < kHistBins; i += NT) ss->hist[i] = 0;
  if (tid == 0) {
    sc->cutbin = ~0u;
    sc->cutbin_k = ~0u;
    sc->above = 0;
    sc->cb = 0;
    sc->eq = 0;
    sc->nan = 0;
    sc->ovf = 0;
    sc->kcnt = 0;
    sc->ocnt = 0;
    sc->scnt = 0;
  }
  if (tid < 32) sc->sub[tid] = 0;
  u32 m1 = 0, m2 = 0;  // per 16-bit half-lane: largest / second largest pattern of its 4 x S sample scores
#pragma unroll
  for (u32 i = 0; i < kReg; ++i) {
    if (i < S) {  // wave-uniform
      const u32 idx0 = U.first_index<8>(i * sstride, tid);
      const bool full = (idx0 < n) & (idx0 + 7u < n);
      if (__ballot(!full) != 0ull) sv[i] = mask_outside(sv[i], idx0, n);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const u32 x = sv[i][d];
        const u32 lo = pk_min_i16(m1, x);
        m1 = pk_max_i16(m1, x);
        m2 = pk_max_i16(m2, lo);
      }
    }
  }
  // the ring is primed here, behind the last use of the sample loads (the compiler waits for ALL outstanding loads before
  // it touches the first sample vector): the first stream tiles travel while the cut is computed
  unsigned char* ring = stage + (size_t)wave * PF * 1024;
  if (M > 0u) {  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      ring_issue(U.addr(ti, tid), ring + i * 1024);  // (past the unit's last tile: that tile again, cache hits)
      adv(ti, nsi);
    }
  }
  if (stamp) p.stamps[8] = clock64();
  const u32 inf16 = p.inf16, thr16 = p.thr16;
  const u32 sv4[4] = {m1 & 0xffffu, m1 >> 16, m2 & 0xffffu, m2 >> 16};  // (all >= 0: negative patterns lost against the 0 seed)
  lds_barrier();  // histogram zeroed, control words initialised
  if (__ballot((sv4[0] > inf16) | (sv4[1] > inf16)) != 0ull && lane == 0) sc->nan = 1u;  // a NaN among the largest patterns
#pragma unroll
  for (int j = 0; j < 4; ++j) hist_add(ss->hist, sv4[j] >= thr16, hist_bin(ord_of16<DT>(sv4[j]), p.hist_base, p.hist_sh));
  __syncthreads();
  if (stamp) p.stamps[9] = clock64();
  // Which sample maximum gives the cut.  The K-th largest is a PROVEN lower bound of the unit's K-th score (the K-th largest
  // of a subset is <= that of the whole) and predicts ~K ntiles / S candidates -- 1 500 of the buffer's 2 048 in an 81-tile
  // unit: a head with channel structure (the sample tiles miss the channels that score high) exceeds that by a third and the
  // unit overflows.  Round 5: the cut comes from the rank that predicts sqrt(K x capacity) candidates (the same factor of
  // safety against a short list as against an overflow: 2.6 for K = 300); a list that comes back short reruns the unit
  // from the proven cut, so the result is exact either way.
  u32 R = K;
  if (ntiles > S) {
    const u32 r = (u32)(sqrtf((float)K * (float)(NW * kSeg)) * (float)S / (float)ntiles);
    R = r < 1u ? 1u : (r < K ? r : K);
  }
  hist_two_ranks(ss, sc, R, K);  // (ends with a barrier)
  u32 cut16 = thr16, eq = 0;
  {
    const u32 cutbin = sc->cutbin;
    if (cutbin != ~0u) {  // workgroup-uniform
      const u32 c0 = ceil16_of_ord<DT>(p.hist_base + (cutbin << p.hist_sh));
      cut16 = c0 > thr16 ? c0 : thr16;
      const bool exact = cutbin < kHistBins - 1 && p.hist_sh <= ULP_SH;  // the bin holds ONE representable value
      if (exact) {
        eq = sc->cb;
      } else if (cutbin < kHistBins - 1 && p.hist_sh - ULP_SH <= 5u) {
        // several representable values per bin (f16: 8): the K-th largest of the sample values, exactly, from a
        // 2^(sh - ulp)-bin histogram of the bin's members -- an all-equal f16 image must find its tie value too
        const u32 span = 1u << (p.hist_sh - ULP_SH);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (sv4[j] >= cut16 && hist_bin(ord_of16<DT>(sv4[j]), p.hist_base, p.hist_sh) == cutbin) {
            const u32 k = sv4[j] - cut16;
            atomicAdd(&sc->sub[k < 31u ? k : 31u], 1u);
          }
        __syncthreads();
        if (tid == 0) {
          const u32 need = R - sc->above;
          u32 acc = 0, add = 0, e = 0;
          for (int j = (int)(span < 32u ? span : 32u) - 1; j >= 0; --j) {
            const u32 h = sc->sub[j];
            if (acc + h >= need) {
              add = (u32)j;
              e = h;
              break;
            }
            acc += h;
          }
          sc->cut16 = cut16 + add;
          sc->eq = e;
        }
        __syncthreads();
        cut16 = sc->cut16;
        eq = sc->eq;
      }
    }
  }
  cut16 = (u32)__builtin_amdgcn_readfirstlane((int)cut16);
  eq = (u32)__builtin_amdgcn_readfirstlane((int)eq);
  bool tie_rich = eq >= K && cut16 < inf16;  // the cut value itself fills K of the 1024 sample slots
  // ---- near ties (round 5).  Heads whose scores sit on a few hundred neighbouring 16-bit values (calibrated random weights,
  // the reference-init FPN / BiFPN towers) have thousands of elements per value: the K-th largest of the 1024 sample maxima
  // is then a value whose occurrences alone overflow the key segments (the exact fallback re-reads the unit).  When the cut
  // value is suspiciously frequent among the maxima, COUNT: the S sample tiles are in registers, and the exact numbers G / E
  // of their elements above / at the cut predict what the stream will find (x ntiles / S).
  //   G >= K          K elements of the sample alone lie above the cut: cut + 1 is a PROVEN lower bound of the K-th score;
  //   G predicts >= 2K in the unit: raise SPECULATIVELY -- the stream collects everything above the cut, and if fewer than K
  //                   keys come back the unit runs the exact fallback from the last proven cut (the result stays exact);
  //   else            stop: if G + E fits, collect everything at or above the cut as usual; if not, the cut value itself is
  //                   needed -- settle it as a tie if it is dense enough for a short prefix walk.
  u32 cutv = thr16;  // the last PROVEN lower bound (what the fallback may start from): the lowest value of the K-th maximum's bin
  {
    const u32 cbk = sc->cutbin_k;
    if (cbk != ~0u) {
      const u32 c0 = ceil16_of_ord<DT>(p.hist_base + (cbk << p.hist_sh));
      cutv = c0 > thr16 ? c0 : thr16;
    }
    // the cut IS proven when it comes from rank K itself, or when its value fills K of the sample maxima
    cutv = (R == K || tie_rich || cutv > cut16) ? cut16 : cutv;
    cutv = (u32)__builtin_amdgcn_readfirstlane((int)cutv);
  }
  bool spec = cut16 > cutv;  // cut16 is a prediction
  bool counted = false;
  if (!tie_rich && eq >= kNearTie && cut16 < inf16) {  // workgroup-uniform
    const u32 per = ntiles / S;  // (>= 1) unit tiles per sample tile
    u32 G = 0, E = 0;
    bool fits = false;
    counted = true;
    for (int it = 0; it < 8; ++it) {
      if (tid == 0) {
        sc->ngt = 0;
        sc->neq = 0;
      }
      __syncthreads();
      u32 g = 0, e = 0;
#pragma unroll
      for (u32 i = 0; i < kReg; ++i)
        if (i < S) {  // wave-uniform (vectors outside the image were zeroed by mask_outside)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const u32 x = sv[i][d], lo = x & 0xffffu, hi = x >> 16;
            g += ((int)(short)(u16)lo > (int)cut16 ? 1u : 0u) + ((int)(short)(u16)hi > (int)cut16 ? 1u : 0u);
            e += (lo == cut16 ? 1u : 0u) + (hi == cut16 ? 1u : 0u);
          }
        }
      u32 ge = (g << 16) | e;  // (<= 128 each per lane)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ge += (u32)__shfl_xor((int)ge, o, 64);  // <= 8192 per field per wave
      if (lane == 0) {
        atomicAdd(&sc->ngt, ge >> 16);
        atomicAdd(&sc->neq, ge & 0xffffu);
      }
      __syncthreads();
      G = sc->ngt;
      E = sc->neq;
      __syncthreads();  // every thread has its copy before thread 0 clears the counters again: the decisions below must be uniform
      const bool room = it < 7 && cut16 + 1u < inf16;
      if (G >= K && room) {  // "above the cut" IS "at or above the next pattern"
        cut16 += 1u;
        cutv = cut16;
        spec = false;
        continue;
      }
      // (not "the first cut that fits": the sample of a channel-structured head under-predicts by a third and more, so the
      //  cut goes up as long as the prediction stays at 2 K -- the SMALLEST list the sample still vouches for)
      if (G * per >= 2u * K && room) {
        cut16 += 1u;
        spec = true;
        continue;
      }
      fits = (G + E) * per <= kNearCap;
      break;
    }
    cut16 = (u32)__builtin_amdgcn_readfirstlane((int)cut16);
    cutv = (u32)__builtin_amdgcn_readfirstlane((int)cutv);
    // a tie walk visits ~K S / E tiles, one dependent load each: worth it up to about four samples' worth of tiles
    if (!fits && !spec && G * per <= kNearCap && E * 4u >= K) tie_rich = true;
  }
  if (stamp) p.stamps[10] = clock64();

  // ---- tie prefix (tie-rich units only): the first K scores EQUAL to the cut, in index order ------------------------------
  u32 ntie = 0;
  u32 ccut = cut16;  // what the filters compare against
  if (tie_rich) {
    u32 need = K;
    for (u32 t = 0; t < ntiles && need > 0u; ++t) {  // workgroup-uniform
      const u32x4 v = *reinterpret_cast<const u32x4*>(U.addr(t, tid));
      const u32 idx0 = U.first_index<8>(t, tid);
      u32 em = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) em |= ((half_of(v, e) == cut16) & (idx0 + (u32)e < n)) ? (1u << e) : 0u;
      const u32 c = (u32)__popc(em);
      u32 excl = 0, tot = 0;
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const u64 mb = __ballot((c >> bit) & 1u);
        excl += mbcnt(mb) << bit;
        tot += (u32)__popcll(mb) << bit;
      }
      if (lane == 0) sc->wcount[wave] = tot;
      __syncthreads();
      u32 before = 0, all = 0;
#pragma unroll
      for (u32 w = 0; w < NW; ++w) {
        const u32 cw = sc->wcount[w];
        before += w < wave ? cw : 0u;
        all += cw;
      }
      u32 left = em, pos = before + excl;  // positions of this lane's ties among the tile's ties (index order)
      const u32 base = K - need;
      while (left) {
        const u32 e = (u32)__ffs((int)left) - 1u;
        left &= left - 1u;
        if (pos < need) tiebuf[base + pos] = ((u64)ord_of16<DT>(cut16) << 32) | (u64)(~(idx0 + e));
        ++pos;
      }
      need = need > all ? need - all : 0u;
      __syncthreads();
    }
    ntie = K - need;
    ccut = cut16 + 1u;  // the unit's ties are settled: everything else must beat the cut
  }
  if (stamp) p.stamps[1] = clock64();

  // ---- filter + extraction, interleaved.  A vector that holds a 16-bit pattern >= cut joins the wave's queue; as soon as
  // 64 are waiting, the wave turns them into keys (one vector per lane, dense) in its own segment of the key buffer and
  // counts them in the select histogram -- in the shadow of the ring's loads, which is where this wave would otherwise
  // sit in s_waitcnt.  Tests of the reference there: value >= cut as a NUMBER (NaN never passes, box.py:440), index inside
  // the image.  No cross-wave traffic, no barrier until the stream is over.
  const u32 cm1 = (ccut - 1u) | ((ccut - 1u) << 16);
  for (u32 i = tid; i < kHistBins; i += NT) ss->hist[i] = 0;  // (phase H is done with it; the barrier is below)
  u32x4* vq = vqb + (size_t)wave * kVq;
  u16* iq = iqb + (size_t)wave * kVq;
  u32 wcnt = 0, xdone = 0, kc = 0;  // vectors queued / extracted, keys in the segment (wave-uniform)
  bool ovf = false;
  auto extract = [&](u32 nround) {  // the next min(64, nround) queued vectors
    const bool have = lane < nround;
    const u32 slot = (xdone + lane) & (kVq - 1u);
    const u32x4 v = vq[slot];
    const u32 idx0 = (U.vec0 + (u32)iq[slot]) * 8u - U.head;
    xdone += nround < 64u ? nround : 64u;
    u32 pm = 0;  // elements that pass
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const u32 w16 = v[d];
      if (pk_max_i16(w16, cm1) != cm1) {  // one of the two halves is >= cut: the exact tests on both
        const u32 lo = w16 & 0xffffu, hi = w16 >> 16;
        pm |= (((int)(short)(u16)lo >= (int)ccut) & (lo <= inf16) & (idx0 + 2u * d < n)) ? (1u << (2 * d)) : 0u;
        pm |= (((int)(short)(u16)hi >= (int)ccut) & (hi <= inf16) & (idx0 + 2u * d + 1u < n)) ? (2u << (2 * d)) : 0u;
      }
    }
    pm = have ? pm : 0u;
    const u32 c = (u32)__popc(pm);
    const u64 any = __ballot(c != 0u);
    if (any == 0ull) return;
    u32 excl, tot;
    if (__ballot(c > 1u) == 0ull) {  // the usual case: one candidate per vector
      excl = mbcnt(any);
      tot = (u32)__popcll(any);
    } else {
      excl = 0;
      tot = 0;
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const u64 mb = __ballot((c >> bit) & 1u);
        excl += mbcnt(mb) << bit;
        tot += (u32)__popcll(mb) << bit;
      }
    }
    // one key buffer for the workgroup, one LDS cursor (round 5: with a segment per wave, a unit whose candidates cluster in
    // two of the four lane quarters overflowed at 1 660 of its 2 048 slots -- channel-structured heads do that)
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&sc->kcnt, tot);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    if (base + tot > NW * kSeg) {  // wave-uniform: more candidates than the buffer holds (a misleading sample)
      ovf = true;
      return;
    }
    u32 at = base + excl, left = pm;
    while (left) {
      const u32 e = (u32)__ffs((int)left) - 1u;
      left &= left - 1u;
      const u32 w16 = e < 2u ? v[0] : (e < 4u ? v[1] : (e < 6u ? v[2] : v[3]));
      const u32 h = (e & 1u) ? (w16 >> 16) : (w16 & 0xffffu);
      const u32 o = ord_of16<DT>(h);
      kbuf[at++] = ((u64)o << 32) | (u64)(~(idx0 + e));
      atomicAdd(&ss->hist[hist_bin(o, p.hist_base, p.hist_sh)], 1u);
    }
    kc += tot;
  };
  lds_barrier();  // histogram zeroed (and, in a tie-rich unit, the tie keys written)
  if (tid == 0 && ntie) atomicAdd(&ss->hist[hist_bin(ord_of16<DT>(cut16), p.hist_base, p.hist_sh)], ntie);
#pragma unroll
  for (u32 i = 0; i < kReg; ++i)
    if (i < S && !ovf) {  // wave-uniform
      append_vec(sv[i], any_ge(sv[i], cm1), i * sstride * NT + tid, vq, iq, wcnt);
      if (wcnt - xdone >= 64u) extract(64u);
    }
  for (u32 q = 0; q < M; ++q) {
    unsigned char* slot = ring + (size_t)(q & (u32)(PF - 1)) * 1024;
    const u32x4 v = ring_take<PF - 1>(slot + lane * 16);  // the oldest of the PF requests has landed
    ring_issue(U.addr(ti, tid), slot);                    // (past the unit's last tile: that tile again, cache hits)
    adv(ti, nsi);
    const u32 lid = tc * NT + tid;
    adv(tc, nsc);
    if (ovf) continue;
    // a lane behind the image's last vector re-read that vector (clamped address): it must not count twice
    append_vec(v, any_ge(v, cm1) & (U.vec0 + lid <= U.vlast), lid, vq, iq, wcnt);
    if (wcnt - xdone >= 64u) extract(64u);
  }
  if (M > 0u) ring_drain();
  while (!ovf && xdone < wcnt) extract(wcnt - xdone);  // (at most two trips)
  if (lane == 0) {
    sc->wcount[wave] = kc;
    if (ovf) sc->ovf = 1u;
  }
  __syncthreads();
  if (wall && tid == 0 && sc->ovf) {  // (debug) what the overflowing unit looked like
    p.stamps[24 + 2 * 3000] = ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)cut16 << 16) | (unsigned long long)ntiles;
    p.stamps[24 + 2 * 3000 + 1] = ((unsigned long long)sc->wcount[0] << 48) | ((unsigned long long)sc->wcount[1] << 32) |
                                  ((unsigned long long)sc->wcount[2] << 16) | (unsigned long long)sc->wcount[3];
    p.stamps[24 + 2 * 3001] = ((unsigned long long)S << 48) | ((unsigned long long)sstride << 32) | ((unsigned long long)eq << 16) | (unsigned long long)sc->above;
    p.stamps[24 + 2 * 3001 + 1] = ((unsigned long long)sc->cb << 32) | (unsigned long long)(tie_rich ? 1u : 0u) | ((unsigned long long)thr16 << 8);
  }
  if (stamp) p.stamps[2] = clock64();

  u32 cnt = 0;
  bool done = false;
  bool ordered = false;  // the unit's list is its ties alone, in index order (the flag in cand_cnt: see the end of the kernel)
  if ((sc->ovf | sc->nan) == 0u) {
    if (stamp) p.stamps[11] = clock64();
    // ---- select: exact top-K of the extracted keys (+ the tie keys), unordered, straight to the workspace ----------
    const u32 nkeys = sc->kcnt;  // (<= NW * kSeg: an overflowing round returns before it writes, and sets ovf)
    const u32 nk = ntie + nkeys;
    if (spec && nk < K) {
      // the prediction failed (fewer than K scores above the raised cut): the exact stream from the proven cut below
    } else if (nk <= K) {
      for (u32 i = tid; i < nkeys; i += NT) out[i] = kbuf[i];
      for (u32 i = tid; i < ntie; i += NT) out[nk - ntie + i] = tiebuf[i];
      cnt = nk;
      ordered = ntie != 0u && nk == ntie;  // nothing beat the cut: the list IS the tie prefix (workgroup-uniform)
    } else {
      if (tid == 0) sc->cutbin = ~0u;
      __syncthreads();
      hist_kth_bin(ss, sc, K);
      const u32 cbin = sc->cutbin, above = sc->above, cb = sc->cb, need = K - above;
      if (cb > 512u) {  // heavy ties inside one bin: the generic exact select over one contiguous array (rare)
        // the tie keys behind the extracted ones (tiebuf follows kbuf: the ranges may overlap -- through registers)
        u64 t0 = 0, t1 = 0;
        if (tid < ntie) t0 = tiebuf[tid];
        if (tid + NT < ntie) t1 = tiebuf[tid + NT];
        __syncthreads();
        if (tid < ntie) kbuf[nkeys + tid] = t0;
        if (tid + NT < ntie) kbuf[nkeys + tid + NT] = t1;
        __syncthreads();
        const u64 T = wg_select_kth<NT>(kbuf, nk, K, ss);
        for (u32 i = tid; i < nk; i += NT) {
          const u64 k = kbuf[i];
          if (k >= T) out[atomicAdd(&sc->ocnt, 1u)] = k;
        }
      } else {
        u64* small = reinterpret_cast<u64*>(ss->hist);  // (the histogram has been read: cb <= 512 keys fit)
        auto classify = [&](u64 k, bool have) {  // all lanes of the wave
          const u32 kb = hist_bin((u32)(k >> 32), p.hist_base, p.hist_sh);
          const bool win = have && kb > cbin, edge = have && kb == cbin;
          const u64 mw = __ballot(win), me = __ballot(edge);
          u32 bw = 0, be = 0;
          if (lane == 0) {
            if (mw) bw = atomicAdd(&sc->ocnt, (u32)__popcll(mw));
            if (me) be = atomicAdd(&sc->scnt, (u32)__popcll(me));
          }
          bw = (u32)__builtin_amdgcn_readfirstlane((int)bw);
          be = (u32)__builtin_amdgcn_readfirstlane((int)be);
          if (win) out[bw + mbcnt(mw)] = k;
          if (edge) small[be + mbcnt(me)] = k;
        };
        for (u32 i0 = 0; i0 < nkeys; i0 += NT) classify(i0 + tid < nkeys ? kbuf[i0 + tid] : 0ull, i0 + tid < nkeys);
        for (u32 i0 = 0; i0 < ntie; i0 += NT) classify(i0 + tid < ntie ? tiebuf[i0 + tid] : 0ull, i0 + tid < ntie);
        __syncthreads();
        for (u32 t = tid; t < cb; t += NT) {
          const u64 me = small[t];
          const u32 r = lds_count_greater(small, 0, cb, me);
          if (r < need) out[above + r] = me;
        }
      }
      cnt = K;
    }
    done = !(spec && nk < K);
  }
  if (!done) {
    // ---- the exact TopK stream over the whole unit (a misleading sample, NaNs, adversarial inputs) ----------------------
    const float c16 = f32_of16<DT>(cutv <= inf16 ? cutv : inf16);  // (the PROVEN cut: cut16 may be a failed prediction)
    const float cut0 = (sc->nan == 0u && c16 > p.thr) ? c16 : p.thr;  // the sample's cut stays a valid lower bound
    cnt = unit_topk_stream<DT, PF>(U, cut0, K, buf, ss, ctl, stage);
    for (u32 i = tid; i < cnt; i += NT) out[i] = buf[i];
  }
  for (u32 i = cnt + tid; i < K; i += NT) out[i] = 0ull;
  // bit 31: every key of the list carries the SAME score and the list is in descending key (= ascending index) order -- what a
  // unit of an all-equal image leaves (the reference-init network: every score of a level is one bf16 value).  levelsel_kernel
  // then takes a full first unit as the level's answer without selecting anything (ssdk_tail.hip).
  if (tid == 0) *out_cnt = cnt | (ordered ? 0x80000000u : 0u);
  // bit 0: took the exact fallback; bits 1-3 (near-tie rule): counted its sample | ended on a predicted cut | settled a tie; bit 4: a key segment overflowed
  if (wall)
    p.stamps[24 + 2 * blockIdx.x + 1] = (wall_clock64() & ~31ull) | (done ? 0ull : 1ull) | (counted ? 2ull : 0ull) | (spec ? 4ull : 0ull) |
                                        ((counted && tie_rich) ? 8ull : 0ull) | (sc->ovf ? 16ull : 0ull);
  if (stamp) {
    p.stamps[3] = clock64();
    p.stamps[4] = clock64();
    p.stamps[5] = ((unsigned long long)(done ? 1u : 0u) << 32) | cnt;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
static float host_f16_to_f32(u32 h) {
  const u32 e = (h >> 10) & 31u, m = h & 1023u;
  float v;
  if (e == 0) v = (float)m * (1.0f / 16777216.0f);  // m * 2^-24
  else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
  else {
    const u32 bits = ((e + 112u) << 23) | (m << 13);
    memcpy(&v, &bits, 4);
  }
  return v;
}

// smallest non-negative 16-bit pattern of `dtype` whose value is >= thr (thr > 0, finite or +inf)
u32 scan16_threshold_pattern(int dtype, float thr) {
  if (dtype == SSDK_BF16) {
    u32 bits;
    memcpy(&bits, &thr, 4);
    return (bits >> 16) + ((bits & 0xffffu) ? 1u : 0u);
  }
  u32 lo = 0, hi = 0x7c00u;  // patterns 0 .. +inf are ordered like their values
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (host_f16_to_f32(mid) >= thr) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

bool scan16_applies(int dtype, float thr, int K) {
  const char* e = getenv("SSDK_SCAN16");  // (read on every call: tests switch kernels inside one process)
  const int env = (e && *e) ? atoi(e) : 1;
  return env != 0 && (dtype == SSDK_BF16 || dtype == SSDK_F16) && thr > 0.0f && thr == thr && K >= 1 && K <= 512;
}

static int scan16_reg() {
  const char* e = getenv("SSDK_SCAN_REG");
  const int r = (e && *e) ? atoi(e) : 16;
  return r == 32 ? 32 : (r == 16 ? 16 : 8);
}

// tiles per unit above which the per-wave key segments are expected to overflow: a unit of T tiles extracts about
// K * max(T / REG, 1.1) keys (the sample's K-th value sits at rank ~K * T / REG of the unit); 3/4 of the four segments.
// (<= 255: a vector's number inside the unit is stored in 16 bits.)
u32 scan16_max_tiles_per_unit(int K) {
  const u32 t = (u32)(3ull * (kScanThreads / 64) * kSeg * (unsigned)scan16_reg() / 4ull / (unsigned)K);
  return t < 4u ? 4u : (t > 255u ? 255u : t);
}

int launch_scan16(const ScanParams& sp, int dtype, int B, u32 units_per_image, hipStream_t stream) {
  const int pf = getenv("SSDK_SCAN_PF") && atoi(getenv("SSDK_SCAN_PF")) == 8 ? 8 : 4;  // 16-byte requests in flight per lane
  const size_t lds = scan16_lds_bytes(pf);
  const dim3 grid((unsigned)((size_t)B * units_per_image));
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(kScanThreads), lds, stream, sp);
  };
  const int reg = scan16_reg();
#define SSDK_S16(DT_)                                      \
  do {                                                     \
    if (pf == 8) {                                         \
      if (reg == 32) go(scan16_kernel<DT_, 8, 32>);        \
      else if (reg == 16) go(scan16_kernel<DT_, 8, 16>);   \
      else go(scan16_kernel<DT_, 8, 8>);                   \
    } else {                                               \
      if (reg == 32) go(scan16_kernel<DT_, 4, 32>);        \
      else if (reg == 16) go(scan16_kernel<DT_, 4, 16>);   \
      else go(scan16_kernel<DT_, 4, 8>);                   \
    }                                                      \
  } while (0)
  if (dtype == SSDK_BF16) SSDK_S16(SSDK_BF16);
  else SSDK_S16(SSDK_F16);
#undef SSDK_S16
  return check_launch("scan16_kernel");
}

}  // namespace ssdk
