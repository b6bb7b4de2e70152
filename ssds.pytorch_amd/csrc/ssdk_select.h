// ssdk_select.h -- workgroup-level exact top-K machinery on 64-bit composite keys held in LDS.
//
// Keys are unique (score bits | inverted flat index), so "the K largest keys" is unambiguous and
// equals the reference's topk under the (score desc, index asc) tie contract.
//
//   wg_select_kth   exact K-th largest key of buf[0..n) by adaptive radix select: the 10-bit window
//                   always starts at the highest bit that still differs among the candidates, so
//                   low-entropy keys (bf16 scores, small indices) cost no wasted passes; once <=512
//                   candidates remain the rest is resolved by a rank count.
//   wg_compact_ge   moves the K keys >= T to the front of buf.
//   wg_bitonic_sort_desc   in-LDS bitonic sort (descending).
//   TopKStream      streaming accumulator: append keys that beat the running cut, prune back to the
//                   exact top-K when the buffer passes `limit`; one barrier per tile.
#pragma once
#include "ssdk_common.h"

namespace ssdk {

#ifndef SSDK_RANK_MAX
#define SSDK_RANK_MAX 96
#endif
constexpr u32 kRankMax = SSDK_RANK_MAX;  // candidates resolved by O(cb^2/NT) rank counting (<= 512)

struct SelScratch {  // LDS, 8-byte aligned
  u64 acc_or, acc_and, T;
  u32 hist[1024];  // reused as u64 small[512]
  u32 wsum[32];
  u32 bin, above, cb, small_cnt, sel_cnt, pad;
};

// inclusive suffix sum over the workgroup: returns sum of v over threads with id >= tid
template <int NT>
__device__ __forceinline__ u32 wg_incl_suffix_sum(u32 v, u32* wsum) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 y = __shfl_down(x, d);
    if (lane + d < 64) x += y;
  }
  if (lane == 0) wsum[wave] = x;
  __syncthreads();
  u32 add = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w)
    if (w > (int)wave) add += wsum[w];
  return x + add;
}

// approx_max > 0 (intermediate prunes of a stream): stop after the FIRST radix pass when the bin holding the K-th
// key plus everything above it is at most approx_max keys, and return the bin's lowest possible key: pruning with
// `key >= T` then keeps a superset of the top K (K .. approx_max keys) at a third of the cost -- the ties inside the
// bin (bf16 scores: hundreds per value) are only resolved by the exact select at the end of the stream.
// `fetch(i)`, i < n: the i-th key of the set (any storage: LDS, or global memory behind a predicate).  A source may
// return 0 for "no key here": zero is below every real key (real keys carry the sign-flipped score bit), so it never
// changes which key is the K-th LARGEST as long as K real keys exist.
template <int NT, class Fetch>
__device__ u64 wg_select_kth_f(Fetch fetch, u32 n, u32 K, SelScratch* s, u32 approx_max = 0) {
  const u32 tid = threadIdx.x, lane = tid & 63u;
  constexpr int BPT = 1024 / NT;
  if (tid == 0) {
    s->acc_or = 0;
    s->acc_and = ~0ull;
  }
  __syncthreads();
  u64 pfx_mask = 0, pfx_val = 0, T = 0;
  u32 want = K;
  for (int pass = 0; pass < 8; ++pass) {  // uniform loop; every exit condition is workgroup-uniform
    u64 o = 0, a = ~0ull;
    for (u32 i = tid; i < n; i += NT) {
      u64 k = fetch(i);
      if ((k & pfx_mask) == pfx_val) {
        o |= k;
        a &= k;
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      o |= shfl_xor_u64(o, d);
      a &= shfl_xor_u64(a, d);
    }
    if (lane == 0) {
      atomicOr(&s->acc_or, o);
      atomicAnd(&s->acc_and, a);
    }
    __syncthreads();
    const u64 common = s->acc_and;
    const u64 diff = s->acc_or ^ common;
    if (diff == 0) {  // a single candidate left
      T = s->acc_or;
      break;
    }
    const int top = 63 - __clzll((long long)diff);
    const int shift = top >= 9 ? top - 9 : 0;
    const u32 nbmask = (1u << (top - shift + 1)) - 1u;
    for (u32 i = tid; i < 1024; i += NT) s->hist[i] = 0;
    __syncthreads();
    if (tid == 0) {  // re-arm the reducers for the next pass (everyone has read them)
      s->acc_or = 0;
      s->acc_and = ~0ull;
    }
    for (u32 i0 = 0; i0 < n; i0 += NT) {  // uniform trip count: the wave-level aggregation below needs all lanes
      const u32 i = i0 + tid;
      const u64 k = i < n ? fetch(i) : 0ull;
      const bool c = i < n && (k & pfx_mask) == pfx_val;
      const u32 bin = (u32)(k >> shift) & nbmask;
      const u64 m = __ballot(c);
      if (m) {
        // heavy ties put most candidates of a wave into one bin: let one lane add the whole group
        const u32 lead = (u32)__ffsll((long long)m) - 1u;
        const u32 b0 = __shfl(bin, (int)lead);
        const u64 same = __ballot(c && bin == b0);
        if (lane == lead) atomicAdd(&s->hist[b0], (u32)__popcll(same));
        if (c && bin != b0) atomicAdd(&s->hist[bin], 1u);
      }
    }
    __syncthreads();
    u32 local = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) local += s->hist[tid * BPT + j];
    const u32 incl = wg_incl_suffix_sum<NT>(local, s->wsum);
    const u32 excl = incl - local;
    if (excl < want && want <= incl) {  // exactly one thread
      u32 acc = excl;
      for (int j = BPT - 1; j >= 0; --j) {
        u32 h = s->hist[tid * BPT + j];
        if (acc + h >= want) {
          s->bin = tid * BPT + j;
          s->above = acc;
          s->cb = h;
          break;
        }
        acc += h;
      }
    }
    __syncthreads();
    const u32 bin = s->bin, cb = s->cb;
    want -= s->above;
    const u64 above_top = (top == 63) ? 0ull : (~0ull << (top + 1));
    pfx_val = (common & above_top) | ((u64)bin << shift);
    pfx_mask = (shift == 0) ? ~0ull : (~0ull << shift);
    if (cb == want || (approx_max && pass == 0 && (K - want) + cb <= approx_max)) {
      T = pfx_val;  // the whole bin is selected (or kept: approximate prune); T = smallest possible key of the bin
      break;
    }
    if (cb <= kRankMax) {  // resolve the remaining candidates by rank counting
      u64* small = reinterpret_cast<u64*>(s->hist);
      if (tid == 0) s->small_cnt = 0;
      __syncthreads();
      for (u32 i = tid; i < n; i += NT) {
        u64 k = fetch(i);
        if ((k & pfx_mask) == pfx_val) small[atomicAdd(&s->small_cnt, 1u)] = k;
      }
      __syncthreads();
      for (u32 t = tid; t < cb; t += NT) {
        const u64 me = small[t];
        u32 r = 0;
        for (u32 j = 0; j < cb; ++j) r += (small[j] > me) ? 1u : 0u;
        if (r == want - 1) s->T = me;
      }
      __syncthreads();
      T = s->T;
      break;
    }
  }
  __syncthreads();
  return T;
}

template <int NT>
__device__ u64 wg_select_kth(const u64* buf, u32 n, u32 K, SelScratch* s, u32 approx_max = 0) {
  return wg_select_kth_f<NT>([buf](u32 i) { return buf[i]; }, n, K, s, approx_max);
}

// number of keys of a[j0 .. j1) (LDS) that are greater than `me`: the reads go out eight at a time, so the LDS latency is
// paid once per eight keys instead of once per key (a plain loop waits for every read before it compares)
__device__ __forceinline__ u32 lds_count_greater(const u64* a, u32 j0, u32 j1, u64 me) {
  u32 g = 0, j = j0;
  for (; j + 8 <= j1; j += 8) {
    u64 k[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) k[t] = a[j + t];
#pragma unroll
    for (int t = 0; t < 8; ++t) g += k[t] > me ? 1u : 0u;
  }
  for (; j < j1; ++j) g += a[j] > me ? 1u : 0u;
  return g;
}

// keep the K keys >= T at the front of buf (sel is a K-entry LDS staging area)
template <int NT>
__device__ void wg_compact_ge(u64* buf, u32 n, u64 T, u32 K, u64* sel, SelScratch* s) {
  const u32 tid = threadIdx.x;
  if (tid == 0) s->sel_cnt = 0;
  __syncthreads();
  for (u32 i = tid; i < n; i += NT) {
    u64 k = buf[i];
    if (k >= T) {
      u32 p = atomicAdd(&s->sel_cnt, 1u);
      if (p < K) sel[p] = k;
    }
  }
  __syncthreads();
  for (u32 i = tid; i < K; i += NT) buf[i] = sel[i];
  __syncthreads();
}

// keep every key >= T at the front of buf, in place: each thread first pulls its (<= PER) keys into registers, so
// nobody overwrites a key that has not been read yet.  Returns the number kept (n <= PER * NT).
template <int NT, int PER>
__device__ u32 wg_compact_ge_inplace(u64* buf, u32 n, u64 T, SelScratch* s) {
  const u32 tid = threadIdx.x;
  u64 r[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const u32 i = tid + (u32)j * NT;
    r[j] = i < n ? buf[i] : 0ull;
  }
  if (tid == 0) s->sel_cnt = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (r[j] >= T && r[j] != 0ull) buf[atomicAdd(&s->sel_cnt, 1u)] = r[j];
  __syncthreads();
  return s->sel_cnt;
}

template <int NT>
__device__ void wg_bitonic_sort_desc(u64* a, u32 M) {  // M power of two
  const u32 tid = threadIdx.x;
  for (u32 k = 2; k <= M; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 i = tid; i < (M >> 1); i += NT) {
        const u32 lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const u32 hi = lo | j;
        const bool desc = (lo & k) == 0;
        const u64 x = a[lo], y = a[hi];
        if ((x < y) == desc) {
          a[lo] = y;
          a[hi] = x;
        }
      }
      __syncthreads();
    }
  }
}

// ---- sorting by wave-local runs + rank merge ---------------------------------------------------
// A wave sorts 128 keys in registers (2 per lane, element e = lane + 64 r): 28 compare-exchange steps, cross-lane
// partners by shuffles, no LDS traffic and no barrier.  Sorted runs of 128 are then merged WITHOUT a merge network:
// keys are unique, so the rank of a key in the union is its index in its own run plus, for every other run, the number
// of larger keys there (8-step binary search; the searches over up to 8 sibling runs advance together, so a key costs
// 8 LDS round trips per group of 8 siblings instead of 64).  Zero keys are padding: they sort to the end of a run and
// take no rank.
template <u32 J>
__device__ __forceinline__ u64 xor_lane_u64(u64 v, u32 idx32) {  // value of lane (lane ^ J)
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  if constexpr (J < 32) {  // ds_swizzle, bit-mask mode (and 0x1f, or 0, xor J): no address register, no index math
    lo = (u32)__builtin_amdgcn_ds_swizzle((int)lo, (int)((J << 10) | 0x1fu));
    hi = (u32)__builtin_amdgcn_ds_swizzle((int)hi, (int)((J << 10) | 0x1fu));
  } else {
    lo = (u32)__builtin_amdgcn_ds_bpermute((int)idx32, (int)lo);
    hi = (u32)__builtin_amdgcn_ds_bpermute((int)idx32, (int)hi);
  }
  return ((u64)hi << 32) | lo;
}

template <u32 K2, u32 J>
__device__ __forceinline__ void wave_sort128_step(u64& x0, u64& x1, u32 lane, u32 idx32) {
  if constexpr (J == 64) {  // partner lives in the same lane; K2 == 128: one descending block
    const bool gt = x0 > x1;
    const u64 mx = gt ? x0 : x1, mn = gt ? x1 : x0;
    x0 = mx;
    x1 = mn;
  } else {
    const u64 y0 = xor_lane_u64<J>(x0, idx32), y1 = xor_lane_u64<J>(x1, idx32);
    const bool lo = (lane & J) == 0;  // this element is the lower index of its pair
    // block direction: descending iff (e & K2) == 0, e = lane + 64 r
    const bool d0 = K2 < 64 ? (lane & K2) == 0 : true;
    const bool d1 = K2 < 64 ? (lane & K2) == 0 : (K2 == 64 ? false : true);
    const bool m0 = lo == d0, m1 = lo == d1;  // keep the larger of the pair?
    const bool g0 = x0 > y0, g1 = x1 > y1;
    x0 = (g0 == m0) ? x0 : y0;
    x1 = (g1 == m1) ? x1 : y1;
  }
}
template <u32 K2, u32 J>
__device__ __forceinline__ void wave_sort128_merge(u64& x0, u64& x1, u32 lane, u32 idx32) {
  wave_sort128_step<K2, J>(x0, x1, lane, idx32);
  if constexpr (J > 1) wave_sort128_merge<K2, J / 2>(x0, x1, lane, idx32);
}
template <u32 K2>
__device__ __forceinline__ void wave_sort128_stage(u64& x0, u64& x1, u32 lane, u32 idx32) {
  wave_sort128_merge<K2, K2 / 2>(x0, x1, lane, idx32);
  if constexpr (K2 < 128) wave_sort128_stage<K2 * 2>(x0, x1, lane, idx32);
}
__device__ __forceinline__ void wave_sort128_desc(u64& x0, u64& x1) {
  const u32 lane = lane_id();
  wave_sort128_stage<2>(x0, x1, lane, (lane ^ 32u) << 2);
}

// number of keys of the descending run `run[0..len)` that are larger than `key`, for W runs at once (runs r0 .. r0+W-1
// of `stride` keys each at `base`; run `skip` and runs >= nruns count 0).  `steps` >= bits(len).
template <int W>
__device__ __forceinline__ u32 count_greater_xw(const u64* base, u32 stride, const u32* lens, u32 fixed_len, u32 r0,
                                                u32 nruns, u32 skip, u64 key, int steps) {
  // Branch-free: every probe is an unconditional read of an in-range slot (bitwise &, no short-circuit -- with `&&` the
  // compiler sinks each load into its own branch and waits for it there, which serialises the W searches).
  u32 pos[W], len[W];
  const u64* run[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const u32 r = r0 + (u32)j;
    const bool on = (r < nruns) & (r != skip);
    const u32 rr = r < nruns ? r : r0;
    run[j] = base + (size_t)rr * stride;
    len[j] = on ? (lens ? lens[rr] : fixed_len) : 0u;
    pos[j] = 0;
  }
  for (u32 step = 1u << (steps - 1); step > 0; step >>= 1) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const u32 probe = pos[j] + step;                       // "are the first `probe` keys all larger?"
      const u32 at = (probe < stride ? probe : stride) - 1u;  // clamped: always a slot of this run
      const u64 v = run[j][at];
      pos[j] = ((probe <= len[j]) & (v > key)) ? probe : pos[j];
    }
  }
  u32 sum = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) sum += pos[j];
  return sum;
}

// keys larger than `key` in all runs but `skip` (random LDS reads: the width follows the number of runs left, so a level
// with two lists costs 1/4 of one with eight)
__device__ __forceinline__ u32 count_greater_runs(const u64* base, u32 stride, const u32* lens, u32 fixed_len, u32 nruns,
                                                  u32 skip, u64 key, int steps) {
  u32 sum = 0, r0 = 0;
  while (r0 < nruns) {
    const u32 left = nruns - r0;
    if (left >= 8) {
      sum += count_greater_xw<8>(base, stride, lens, fixed_len, r0, nruns, skip, key, steps);
      r0 += 8;
    } else if (left >= 4) {
      sum += count_greater_xw<4>(base, stride, lens, fixed_len, r0, nruns, skip, key, steps);
      r0 += 4;
    } else if (left >= 2) {
      sum += count_greater_xw<2>(base, stride, lens, fixed_len, r0, nruns, skip, key, steps);
      r0 += 2;
    } else {
      if (r0 != skip) sum += count_greater_xw<1>(base, stride, lens, fixed_len, r0, nruns, skip, key, steps);
      r0 += 1;
    }
  }
  return sum;
}

// a[0 .. nchunks*128) in LDS (zero = padding): every wave sorts its chunks of 128 in place (descending).  Needs a
// barrier before (a[] complete) and one after (before anybody reads another wave's run).
template <int NT>
__device__ __forceinline__ void wg_sort_runs128(u64* a, u32 nchunks) {
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;
  constexpr u32 NW = NT / 64;
  for (u32 c = wave; c < nchunks; c += NW) {
    u64 x0 = a[c * 128 + lane], x1 = a[c * 128 + 64 + lane];
    wave_sort128_desc(x0, x1);
    a[c * 128 + lane] = x0;
    a[c * 128 + 64 + lane] = x1;
  }
}

// every non-zero key k of the sorted runs with kmin <= k < kmax is handed to emit(rank, key), rank = its position in
// the descending order of ALL keys.  Returns (per thread) how many keys it emitted.
template <int NT, class Emit>
__device__ __forceinline__ u32 wg_rank_emit(const u64* a, u32 nchunks, u64 kmin, u64 kmax, Emit emit) {
  u32 mine = 0;
  for (u32 i = threadIdx.x; i < nchunks * 128; i += NT) {
    const u64 key = a[i];
    if (key == 0ull || key < kmin || key >= kmax) continue;
    emit((i & 127u) + count_greater_runs(a, 128, nullptr, 128, nchunks, i >> 7, key, 8), key);
    ++mine;
  }
  return mine;
}

// sort + rank of everything (barrier before; none after)
template <int NT, class Emit>
__device__ __forceinline__ void wg_rank_sort_desc(u64* a, u32 nchunks, Emit emit) {
  wg_sort_runs128<NT>(a, nchunks);
  __syncthreads();
  (void)wg_rank_emit<NT>(a, nchunks, 1ull, ~0ull, emit);
}

// ---- streaming accumulator --------------------------------------------------------------------
constexpr u32 kStreamCap = 4096;    // LDS key slots of a stream (== kCap of the kernels that use it)
constexpr u32 kApproxKeep = 1024;   // an intermediate prune may keep up to this many keys
struct StreamCtl {  // LDS
  u32 cnt;
  u32 flag[2];
  u32 pad;
};

// Appends the lanes of this wave whose `pass` is set.  Must be called by all 64 lanes of a wave.
// `tile` is the workgroup-uniform tile counter (prune protocol, see finish_tile()).
__device__ __forceinline__ void stream_append(u64* buf, StreamCtl* c, u32 limit, u32 tile, bool pass,
                                              u64 key) {
  const u64 m = __ballot(pass);
  if (m) {
    const u32 nw = __popcll(m);
    u32 base = 0;
    if (lane_id() == 0) {
      base = atomicAdd(&c->cnt, nw);
      if (base <= limit && base + nw > limit) c->flag[tile & 1u] = tile + 1u;  // the unique crosser
    }
    base = __builtin_amdgcn_readfirstlane(base);
    if (pass) buf[base + mbcnt(m)] = key;
  }
}

// After every tile: one barrier, then (workgroup-uniform) prune back to the exact top-K if the buffer
// crossed `limit` during this tile.  Returns true when the running cut (key T) changed.
template <int NT>
__device__ __forceinline__ bool stream_finish_tile(u64* buf, u64* sel, SelScratch* s, StreamCtl* c,
                                                   u32 tile, u32 K, u64* T_out) {
  // LDS-only synchronisation: the appends are LDS operations, so lgkmcnt(0) + s_barrier orders them for the flag
  // read below.  (__syncthreads() also drains vmcnt, i.e. the caller's global prefetches, on every tile.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (c->flag[tile & 1u] != tile + 1u) return false;
  const u32 n = c->cnt;
  (void)sel;
  const u64 T = wg_select_kth<NT>(buf, n, K, s, kApproxKeep);  // superset prune: K .. kApproxKeep keys survive
  const u32 kept = wg_compact_ge_inplace<NT, kStreamCap / NT>(buf, n, T, s);
  if (threadIdx.x == 0) c->cnt = kept;
  __syncthreads();
  *T_out = T;
  return true;
}

// End of stream: exact top-min(K, cnt) at the front of buf (unordered).  Returns the count.
template <int NT>
__device__ __forceinline__ u32 stream_finalize(u64* buf, u64* sel, SelScratch* s, StreamCtl* c, u32 K) {
  __syncthreads();
  u32 n = c->cnt;
  if (n > K) {
    const u64 T = wg_select_kth<NT>(buf, n, K, s);
    wg_compact_ge<NT>(buf, n, T, K, sel, s);
    n = K;
  }
  return n;
}

}  // namespace ssdk
