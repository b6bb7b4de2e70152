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
template <int NT>
__device__ u64 wg_select_kth(const u64* buf, u32 n, u32 K, SelScratch* s, u32 approx_max = 0) {
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
      u64 k = buf[i];
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
      const u64 k = i < n ? buf[i] : 0ull;
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
        u64 k = buf[i];
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
