// ssdk_decode.hip -- threshold + exact top-k + box decode + centre rescoring on gfx950.
//
// Replaces box.decode (reference ssds/modeling/layers/box.py:408-477), which the reference runs as
// ~25 ATen launches per (image, level) with nonzero() host syncs.  Here:
//
//   scan_kernel<DT>     ONE pass over the conf tensor (the HBM-bound part: 2 B/score in bf16).  Each
//                       workgroup streams a "unit" (tpu x 256 x 16 B) of one (image, level)
//                       with 16-byte coalesced loads, 4 loads in flight per lane, and keeps the exact
//                       top-K of what it has seen in LDS (TopK stream, ssdk_select.h).  It emits <=K
//                       64-bit keys (score bits | ~flat index) per unit.
//   level_kernel        per (image, level): merges the units' keys (same streaming top-K), sorts the
//                       K winners, gathers their 4 deltas, applies delta2box (box.py:74-87) and the
//                       centre rescoring (box.py:464-471) in the reference's fp32 operation order and
//                       writes the zero-padded [B, top_n] outputs (box.py:430-432, 473-475).
//
// Algorithmic HBM bytes: the conf tensor once + 4 deltas per winner + outputs.
#include "ssdk_scan.h"

namespace ssdk {



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



template <int DT, int E>
__device__ __forceinline__ void hist_elems(const u32x4& v, u32 idx0, u32 n, float thr, u32 hbase, u32 hsh, u32* hist) {
  if constexpr (E < DType<DT>::vec) {
    const float s = vec_elem<DT, E>(v);
    hist_add(hist, (idx0 + E < n) & (s >= thr), hist_bin(ord_f32(s), hbase, hsh));
    hist_elems<DT, E + 1>(v, idx0, n, thr, hbase, hsh, hist);
  }
}

template <int DT, int E>
__device__ __forceinline__ void sample_top2(const u32x4& v, u32 idx0, u32 n, float thr, u32& m1, u32& m2) {
  if constexpr (E < DType<DT>::vec) {
    const float s = vec_elem<DT, E>(v);
    const u32 o = ((idx0 + E < n) & (s >= thr)) ? ord_f32(s) : 0u;
    const u32 lo = o < m1 ? o : m1;
    m2 = lo > m2 ? lo : m2;
    m1 = o > m1 ? o : m1;
    sample_top2<DT, E + 1>(v, idx0, n, thr, m1, m2);
  }
}

// the next float above f (finite f; -0 counts as +0)
__device__ __forceinline__ float next_up(float f) {
  const u32 b = __builtin_bit_cast(u32, f);
  if ((b & 0x7fffffffu) == 0u) return __builtin_bit_cast(float, 1u);
  return __builtin_bit_cast(float, (b & 0x80000000u) ? b - 1u : b + 1u);
}

// raw bits of element e (run-time) of a 16-byte vector, upcast to fp32
template <int DT>
__device__ __forceinline__ float vec_elem_dyn(const u32x4& v, u32 e) {
  if constexpr (DT == SSDK_F32) {
    const u32 w = e == 0 ? v[0] : (e == 1 ? v[1] : (e == 2 ? v[2] : v[3]));
    return __builtin_bit_cast(float, w);
  } else {
    const u32 q = e >> 1;
    const u32 w = q == 0 ? v[0] : (q == 1 ? v[1] : (q == 2 ? v[2] : v[3]));
    const u32 h = (e & 1u) ? (w >> 16) : (w & 0xffffu);
    if constexpr (DT == SSDK_BF16) return bf16_bits_to_f32(h);
    else return f16_bits_to_f32(h);
  }
}

template <int DT, int E, bool CHECK = true>
__device__ __forceinline__ void fast_flags(const u32x4& v, u32 idx0, u32 n, float cut, u32& pmask) {
  if constexpr (E < DType<DT>::vec) {
    const float s = vec_elem<DT, E>(v);
    if constexpr (CHECK) pmask |= ((idx0 + E < n) & (s >= cut)) ? (1u << E) : 0u;
    else pmask |= (s >= cut) ? (1u << E) : 0u;  // a tile that lies inside the image: no index test
    fast_flags<DT, E + 1, CHECK>(v, idx0, n, cut, pmask);
  }
}


template <int DT, int PF>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(const ScanParams p) {
  constexpr int NT = kScanThreads;
  constexpr int VEC = DType<DT>::vec;
  constexpr int ES = DType<DT>::size;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* buf = reinterpret_cast<u64*>(smem);
  SelScratch* ss = reinterpret_cast<SelScratch*>(buf + kCap);
  StreamCtl* ctl = reinterpret_cast<StreamCtl*>(ss + 1);
  FastCtl* fc = reinterpret_cast<FastCtl*>(ctl + 1);
  unsigned char* stage = smem + scan_fixed_bytes();  // [wave][PF][1 KiB]
  u64* sel = reinterpret_cast<u64*>(stage);          // (used only after the ring has drained)

  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // unit-major block order: the big units of every image are dispatched first, the small levels fill in behind them
  const u32 u = blockIdx.x / p.B;
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
  const u32 uu = u - ubase;

  // image b of this level starts at byte b*n*ES; loads are 16-byte ALIGNED vectors: `head` elements of the first
  // vector belong to the previous image and up to VEC-1 elements of the last one to the next image (or to the
  // padding of the allocation -- an aligned 16-byte vector that holds one valid byte never crosses an allocation
  // boundary); both are masked by the index test.
  const unsigned char* base = (const unsigned char*)cls + (size_t)b * n * ES;
  const u32 head = (u32)((uintptr_t)base & 15u) / ES;
  const unsigned char* abase = base - (size_t)head * ES;
  const u32 nvec = (head + n + VEC - 1) / VEC;   // vectors holding at least one element of this image (>= 1)
  const u32 vec0 = uu * tpu * NT;                // first vector of this unit
  u32 ntiles = 0;
  if (vec0 < nvec) {
    ntiles = (nvec - vec0 + NT - 1) / NT;
    if (ntiles > tpu) ntiles = tpu;
  }
  const u32 K = p.K;
  const bool stamp = p.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) p.stamps[0] = clock64();

  // Branch-free loads: the address is clamped to the image's last vector and what lies outside the unit is masked
  // through its index, so every tile issues exactly one load per lane and the compiler can count them (s_waitcnt
  // vmcnt(PF-1) before a tile is consumed).  A prefetch past the unit's last tile re-reads that tile (cache hits).
  const u32 vlast = nvec - 1u;
  auto load_vec = [&](u32 t) -> u32x4 {
    const u32 vi = vec0 + (t < ntiles ? t : ntiles - 1u) * NT + tid;
    return *reinterpret_cast<const u32x4*>(abase + (size_t)(vi < vlast ? vi : vlast) * 16);
  };
  auto first_index = [&](u32 t) -> u32 {  // flat index of this lane's first element of tile t (huge when masked)
    const u32 vi = vec0 + t * NT + tid;
    return (t < ntiles && vi <= vlast) ? vi * VEC - head : 0xffff0000u;
  };

  // ---- phase H: a cut from a sample of the unit's own tiles -------------------------------------------------------
  // Every lane keeps the two largest scores of its share of 8 sample tiles (64 scores): 512 distinct scores of the
  // unit per workgroup.  The K-th largest of those (K <= 512) is a valid lower bound of the unit's K-th largest
  // score -- K scores at or above it have been seen -- and close to the K-th largest of the whole sample, since a lane
  // rarely owns more than two of the sample's top K.  Found without LDS atomics and without a sort (below).
  float cut0 = p.thr;
  bool fast = false, tie_rich = false;
  if (p.fast && ntiles > 0 && K <= 2 * NT) {  // (a wave offers 128 values: kw = K / 4 of them must exist)
    if (tid == 0) {
      fc->overflow = 0;
      fc->cutord = 0;
      fc->nge = 0;
      fc->cum = 0;
    }
    const u32 S = ntiles < kSample ? ntiles : kSample;
    const u32 stride = ntiles / S;
    u32x4 sv[kSample];
#pragma unroll
    for (u32 i = 0; i < kSample; ++i) sv[i] = load_vec((i < S ? i : S - 1u) * stride);
    u32 m1 = 0, m2 = 0;  // ordered bits of the lane's largest / second largest sample score >= thr (0: none)
#pragma unroll
    for (u32 i = 0; i < kSample; ++i) {
      const u32 idx0 = i < S ? first_index(i * stride) : 0xffff0000u;
      sample_top2<DT, 0>(sv[i], idx0, n, p.thr, m1, m2);
    }
    if (stamp) p.stamps[8] = clock64();
    // the kw-th largest of the wave's 128 values, kw = ceil(K / waves), bit by bit on ballots: every wave then holds
    // kw scores >= its own value, i.e. K scores are >= the smallest of the waves' values
    constexpr u32 NWh = NT / 64;
    const u32 kw = (K + NWh - 1) / NWh;
    // (16-bit scores upcast to fp32 carry zeros below their mantissa: those bits of the answer are zero, no trips for them)
    constexpr int kLowBit = DT == SSDK_BF16 ? 16 : DT == SSDK_F16 ? 13 : 0;
    u32 wv = 0;
    for (int bit = 31; bit >= kLowBit; --bit) {
      const u32 c = wv | (1u << bit);
      const u32 cntc = (u32)__popcll(__ballot(m1 >= c)) + (u32)__popcll(__ballot(m2 >= c));
      wv = cntc >= kw ? c : wv;  // wave-uniform
    }
    if (lane == 0) fc->wcount[wave] = wv;
    if (stamp) p.stamps[9] = clock64();
    __syncthreads();
    if (tid == 0) {
      u32 mn = ~0u;
#pragma unroll
      for (u32 w = 0; w < NWh; ++w) mn = fc->wcount[w] < mn ? fc->wcount[w] : mn;
      fc->cutord = mn;
    }
    if (stamp) p.stamps[10] = clock64();
    __syncthreads();
    const u32 cutord = fc->cutord;
    if (cutord) {
      const float edge = unord_f32(cutord);
      cut0 = edge > p.thr ? edge : p.thr;
    }
    // how many keys the streaming pass will collect, extrapolated from the sample: scores of the sample above the cut,
    // plus the scores equal to it as far as the waves' tie budgets (below) let them in
    const float cut_next = next_up(cut0);
    u32 nge = 0, ngt = 0;
#pragma unroll
    for (u32 i = 0; i < kSample; ++i) {
      const u32 idx0 = i < S ? first_index(i * stride) : 0xffff0000u;
      u32 pm = 0, pg = 0;
      fast_flags<DT, 0>(sv[i], idx0, n, cut0, pm);
      fast_flags<DT, 0>(sv[i], idx0, n, cut_next, pg);
      nge += (u32)__popc(pm);
      ngt += (u32)__popc(pg);
    }
    u32 packed = (nge << 16) | ngt;  // (<= 64 scores per lane: both sums of a wave fit 16 bits)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) packed += __shfl_xor(packed, d);
    if (lane == 0 && packed) {
      atomicAdd(&fc->nge, packed >> 16);
      atomicAdd(&fc->cum, packed & 0xffffu);
    }
    __syncthreads();
    // per wave: its share of the scores above the cut plus the ties its budget lets in (at most K + 511); beyond ~7/8
    // of a wave buffer the TopK stream (which prunes) is the better tool
    constexpr u32 NWv = NT / 64;
    const unsigned long long gt_w = (unsigned long long)fc->cum * ntiles / ((unsigned long long)S * NWv);
    unsigned long long eq_w = (unsigned long long)(fc->nge - fc->cum) * ntiles / ((unsigned long long)S * NWv);
    tie_rich = eq_w * NWv >= K;  // the unit is expected to hold K scores equal to the cut: see the tie prefix below
    eq_w = eq_w < K + 511u ? eq_w : K + 511u;
    fast = gt_w + eq_w <= (unsigned long long)(kWaveCap - kWaveCap / 8);
  }

  if (stamp) p.stamps[1] = clock64();
  u32 cnt = 0;
  bool done = false;
  if (fast) {
    // ---- phase S: every wave on its own ----------------------------------------------------------------------------
    // Ties at the cut: of the scores EQUAL to the cut the unit's top K holds the lowest indices, at most K of them -- and
    // within one wave's share of the unit those are a prefix of the wave's ties (its tiles come in index order).  So a
    // wave takes ties only until it has seen K of them (whole vectors: up to K + 511), then compares against the next
    // float above the cut.  An all-equal image (every score of the random-init network) costs each wave one vector.
    u64* wbuf = buf + wave * kWaveCap;
    u32 wcnt = 0, ties = 0;
    float ccut = cut0;
    bool ovf = false;
    auto addr = [&](u32 t) -> const void* {
      const u32 vi = vec0 + (t < ntiles ? t : ntiles - 1u) * NT + tid;
      return abase + (size_t)(vi < vlast ? vi : vlast) * 16;
    };
    if (tie_rich) {
      // Tie prefix: when the unit is rich in scores equal to the cut (an all-equal image: every score), the per-wave
      // budgets would still collect waves x (K + 511) ties, one histogram bin, and the final select would have to
      // order them by index.  The first K ties of the UNIT in index order are found directly instead: tile by tile,
      // the waves exchange their tie counts (one barrier pair per tile; K ties are reached within the first tiles),
      // each wave keeps the ties whose position in the unit's tie order is below K, and the streaming pass then
      // compares against the next float above the cut everywhere.
      u32 need = K;
      for (u32 t = 0; t < ntiles && need > 0u; ++t) {  // workgroup-uniform
        const u32x4 v = load_vec(t);
        const u32 idx0 = first_index(t);
        u32 ge = 0, gt = 0;
        fast_flags<DT, 0, true>(v, idx0, n, cut0, ge);
        fast_flags<DT, 0, true>(v, idx0, n, next_up(cut0), gt);
        const u32 em = ge & ~gt;  // elements equal to the cut
        const u32 c = (u32)__popc(em);
        u32 excl = 0, tot = 0;
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
          const u64 mb = __ballot((c >> bit) & 1u);
          excl += mbcnt(mb) << bit;
          tot += (u32)__popcll(mb) << bit;
        }
        if (lane == 0) fc->wcount[wave] = tot;
        __syncthreads();
        u32 before = 0, all = 0;
#pragma unroll
        for (u32 w = 0; w < NT / 64; ++w) {
          const u32 cw = fc->wcount[w];
          before += w < wave ? cw : 0u;
          all += cw;
        }
        // this lane's ties sit at positions before + excl .. before + excl + c - 1 of the tile's ties (index order)
        u32 left = em, pos = before + excl;
        const u32 room = need > before ? need - before : 0u;
        const u32 take_w = tot < room ? tot : room;  // ties this wave keeps (wave-uniform)
        u32 at = wcnt + excl;
        while (left) {
          const u32 e = (u32)__ffs((int)left) - 1u;
          left &= left - 1u;
          if (pos < need) wbuf[at] = make_key(vec_elem_dyn<DT>(v, e), idx0 + e);
          ++pos;
          ++at;
        }
        wcnt += take_w;
        need = need > all ? need - all : 0u;
        __syncthreads();
      }
      ccut = next_up(cut0);  // the unit's ties are settled: the stream only looks for scores above the cut
    }
    unsigned char* ring = stage + (size_t)wave * PF * 1024;
#pragma unroll
    for (int i = 0; i < PF; ++i) ring_issue(addr(i), ring + i * 1024);
    for (u32 t0 = 0; t0 < ntiles; t0 += PF) {
      // the PF tiles of this round lie inside the image (no head elements of the previous image, no partial last
      // vector, no masked tile behind the unit's end): their scores need no index test
      const bool inside = vec0 + t0 * NT > 0u && t0 + PF <= ntiles && vec0 + (t0 + PF) * NT <= vlast;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const u32 t = t0 + i;
        const u32x4 v = ring_take<PF - 1>(ring + i * 1024 + lane * 16);  // the oldest of the PF requests has landed
        ring_issue(addr(t + PF), ring + i * 1024);  // (past the unit's last tile: that tile again, cache hits)
        u32 pmask = 0;
        if (inside) fast_flags<DT, 0, false>(v, 0u, n, ccut, pmask);
        else fast_flags<DT, 0, true>(v, first_index(t), n, ccut, pmask);
        const u64 any = __ballot(pmask != 0u);
        if (ovf || any == 0ull) continue;
        const u32 idx0 = first_index(t);
        // wave-level compaction in index order (lane-major, element-minor).  A wave-vector of 512 scores holds a
        // handful of candidates, almost always at most one per lane: one ballot then gives every lane its slot.
        u32 excl, tot;
        const u32 c = (u32)__popc(pmask);
        if (__ballot(c > 1u) == 0ull) {
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
        if (wcnt + tot > kWaveCap) {  // wave-uniform
          ovf = true;
          continue;
        }
        u32 at = wcnt + excl, left = pmask, mine = 0;
        while (left) {  // one trip for nearly every lane that has anything
          const u32 e = (u32)__ffs((int)left) - 1u;
          left &= left - 1u;
          const float sc = vec_elem_dyn<DT>(v, e);
          mine += sc == cut0 ? 1u : 0u;
          wbuf[at++] = make_key(sc, idx0 + e);
        }
        wcnt += tot;
        if (ccut == cut0 && __ballot(mine != 0u) != 0ull) {  // wave-uniform: budget still open and ties in this vector
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
          ties += mine;
          if (ties >= K) ccut = next_up(cut0);
        }
      }
    }
    ring_drain();
    for (u32 i = wcnt + lane; i < kWaveCap; i += 64) wbuf[i] = 0ull;  // padding: smaller than every key
    if (lane == 0) {
      fc->wcount[wave] = wcnt;
      if (ovf) fc->overflow = 1u;
    }
    __syncthreads();
    if (stamp) p.stamps[2] = clock64();
    if (fc->overflow == 0u) {
      u32 total = 0;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) total += fc->wcount[w];
      cnt = total < K ? total : K;
      if (total <= K) {
        wg_compact_ge<NT>(buf, kCap, 1ull, K, sel, ss);  // every collected key is a winner: to the front
      } else {
        // exact top-K of the collected keys through the SAME 1024-bin window (every key is >= the cut, so the bins
        // resolve them well): bins above the one holding the K-th key are winners outright, that bin itself is
        // resolved by rank counting -- no radix passes over the buffer.
        for (u32 i = tid; i < kHistBins; i += NT) ss->hist[i] = 0;
        if (tid == 0) {
          fc->cutbin = 0;
          fc->cum = 0;
          ss->sel_cnt = 0;
          ss->small_cnt = 0;
        }
        __syncthreads();
        constexpr u32 PER = kCap / NT;
        u64 mykey[PER];
        u32 mybin[PER];
#pragma unroll
        for (u32 j = 0; j < PER; ++j) {
          mykey[j] = buf[tid + j * NT];
          mybin[j] = hist_bin((u32)(mykey[j] >> 32), p.hist_base, p.hist_sh);
          hist_add(ss->hist, mykey[j] != 0ull, mybin[j]);
        }
        __syncthreads();
        constexpr int BPT = kHistBins / NT;
        u32 local = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) local += ss->hist[tid * BPT + j];
        const u32 incl = wg_incl_suffix_sum<NT>(local, ss->wsum);
        const u32 excl = incl - local;
        if (excl < K && K <= incl) {  // exactly one thread
          u32 acc = excl;
          for (int j = BPT - 1; j >= 0; --j) {
            const u32 h = ss->hist[tid * BPT + j];
            if (acc + h >= K) {
              fc->cutbin = tid * BPT + j;
              fc->cum = acc;  // keys in the bins above
              ss->cb = h;
              break;
            }
            acc += h;
          }
        }
        __syncthreads();
        const u32 cbin = fc->cutbin, above = fc->cum, cb = ss->cb, need = K - above;
        if (cb > 512u) {  // heavy ties inside one bin: the generic exact select
          const u64 T = wg_select_kth<NT>(buf, kCap, K, ss);
          wg_compact_ge<NT>(buf, kCap, T, K, sel, ss);
        } else {
          u64* small = reinterpret_cast<u64*>(ss->hist);  // (the histogram has been read: cb <= 512 keys fit)
          __syncthreads();
#pragma unroll
          for (u32 j = 0; j < PER; ++j) {
            if (mykey[j] == 0ull) continue;
            if (mybin[j] > cbin) sel[atomicAdd(&ss->sel_cnt, 1u)] = mykey[j];
            else if (mybin[j] == cbin) small[atomicAdd(&ss->small_cnt, 1u)] = mykey[j];
          }
          __syncthreads();
          for (u32 t = tid; t < cb; t += NT) {
            const u64 me = small[t];
            u32 r = 0;
            for (u32 j = 0; j < cb; ++j) r += small[j] > me ? 1u : 0u;
            if (r < need) sel[above + r] = me;
          }
          __syncthreads();
          for (u32 i = tid; i < K; i += NT) buf[i] = sel[i];
          __syncthreads();
        }
      }
      done = true;
    }
    __syncthreads();
  }

  if (!done) {
    // ---- TopK stream: exact for every input (ssdk_select.h), one barrier per tile ----------------------------------
    if (tid == 0) {
      ctl->cnt = 0;
      ctl->flag[0] = 0;
      ctl->flag[1] = 0;
    }
    __syncthreads();
    const u32 limit = kCap - NT * VEC;
    float cut = cut0;  // a valid lower bound of the unit's K-th score (or the threshold)
    u32 cut_idx = 0xffffffffu;
    auto addr = [&](u32 t) -> const void* {
      const u32 vi = vec0 + (t < ntiles ? t : ntiles - 1u) * NT + tid;
      return abase + (size_t)(vi < vlast ? vi : vlast) * 16;
    };
    unsigned char* ring = stage + (size_t)wave * PF * 1024;
#pragma unroll
    for (int i = 0; i < PF; ++i) ring_issue(addr(i), ring + i * 1024);
    for (u32 t0 = 0; t0 < ntiles; t0 += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const u32 t = t0 + i;  // tiles past ntiles (the last round of a unit) run masked: no keys, one barrier
        const u32x4 v = ring_take<PF - 1>(ring + i * 1024 + lane * 16);
        ring_issue(addr(t + PF), ring + i * 1024);
        scan_vec<DT>(v, first_index(t), n, cut, cut_idx, buf, ctl, limit, t);
        u64 T;
        if (stream_finish_tile<NT>(buf, sel, ss, ctl, t, K, &T)) {
          cut = key_score(T);
          cut_idx = key_index(T);
        }
      }
    }
    ring_drain();
    cnt = stream_finalize<NT>(buf, sel, ss, ctl, K);
  }

  if (stamp) p.stamps[3] = clock64();
  // The unit's winners leave SORTED (descending key = score desc, index asc): the fused tail kernel merges the units
  // of a level by binary-search ranks instead of selecting again, and single-unit levels need no sort at all there.
  const u32 nchunks = cnt ? (cnt + 127u) >> 7 : 1u;
  for (u32 i = cnt + tid; i < nchunks * 128u; i += NT) buf[i] = 0ull;
  __syncthreads();
  u64* out = p.cand + ((size_t)b * p.units_per_image + u) * K;
  wg_rank_sort_desc<NT>(buf, nchunks, [&](u32 rank, u64 key) { out[rank] = key; });
  for (u32 i = cnt + tid; i < K; i += NT) out[i] = 0ull;
  if (tid == 0) p.cand_cnt[(size_t)b * p.units_per_image + u] = cnt;
  if (stamp) {
    p.stamps[4] = clock64();
    p.stamps[5] = ((unsigned long long)(done ? 1u : 0u) << 32) | cnt;
  }
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
    n = p.cand_cnt[(size_t)b * p.units_per_image + d.unit_base] & 0x7fffffffu;  // (bit 31: scan16_kernel's "ordered ties" flag)
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
static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Level l is cut into round(tiles_l / target) units of equal size (at least one): every unit is close to `target`
// tiles, so the workgroups that carry the bulk of the bytes are balanced (a CU's share of the HBM stream is what
// bounds the kernel), and the small levels are one small unit each.  `exact`: every level uses exactly `target`
// tiles per unit (SSDK_TILES_PER_UNIT, tests).
// `cap`: the largest unit the scan kernel takes on its fast path (scan16_kernel buffers ~K * tiles / REG candidate keys per
// unit: a longer unit overflows its key segments and re-runs on the exact stream).  The balanced split rounds the number of
// units to the NEAREST, so without the cap a level of 1.4 x target tiles became ONE unit of 1.4 x target: found on the
// BiFPN@896 heads, where one 92-tile unit (cap 81) per image took the fallback and tripled the kernel's time.
static void plan_units(DecodePlan* pl, int L, int B, int K, u32 vec, u32 tile, u32 target, bool exact, u32 cap) {
  pl->tiles_per_unit = target;
  u32 base = 0;
  for (int l = 0; l < L; ++l) {
    const u32 tiles = (u32)(((unsigned long long)pl->n[l] + vec + tile - 1) / tile);
    u32 units = exact ? (tiles + target - 1) / target : (tiles + target / 2) / target;
    if (units < 1) units = 1;
    if (!exact && (tiles + units - 1) / units > cap) units = (tiles + cap - 1) / cap;
    pl->tpu[l] = exact ? target : (tiles + units - 1) / units;
    pl->units[l] = (tiles + pl->tpu[l] - 1) / pl->tpu[l];
    pl->unit_base[l] = base;
    base += pl->units[l];
  }
  pl->units_per_image = base;
  pl->cand_bytes = (size_t)B * base * K * sizeof(u64);
  pl->cnt_bytes = (((size_t)B * base * sizeof(u32)) + 255) & ~(size_t)255;
}

// ndet > 0: the caller is ssdk_decode_nms and would like the fused tail kernel (ssdk_tail.hip); pl->fused says whether
// the geometry qualifies (it reads the unit lists from the workspace, so any number of units does).  Units are cut for
// the 16-bit scan (ssdk_scan16.hip) whenever dtype and K allow it -- the threshold is not known when the workspace is
// sized, and scan_kernel takes any unit size.
int make_plan(const ssdk_level* lv, int L, int B, int dtype, int K, DecodePlan* pl, int ndet) {
  pl->fused = false;
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
  // prunes / final select / sort and the K keys written per unit (and merged again behind the scan) amortise.
  int tpu = env_int("SSDK_TILES_PER_UNIT", 0);
  const bool forced = tpu > 0;
  const bool plan16 = scan16_applies(dtype, 1.0f, K);
  if (!forced) {
    const unsigned long long target_wgs = (unsigned long long)env_int("SSDK_TARGET_WGS", 640);
    unsigned long long t = (tiles_total * (unsigned long long)B + target_wgs - 1) / target_wgs;
    tpu = (int)(t < 4 ? 4 : (t > 4096 ? 4096 : t));
    // scan16_kernel buffers ~K * tiles / 8 candidate vectors per unit: longer units would overflow into the fallback
    if (plan16 && tpu > (int)scan16_max_tiles_per_unit(K)) tpu = (int)scan16_max_tiles_per_unit(K);
  } else if (plan16 && tpu > 255) {
    tpu = 255;  // (a vector's number inside its unit is stored in 16 bits)
  }
  plan_units(pl, L, B, K, vec, tile, (u32)tpu, forced, plan16 ? scan16_max_tiles_per_unit(K) : 4096u);
  if (ndet > 0 && env_int("SSDK_DECODE_FUSED", 1) != 0) pl->fused = tail_fits(K, L, ndet) != 0;
  return SSDK_OK;
}

// histogram window shared by the scans' seeding / select phases and the tail's per-level select: [thr, max(1, 2 thr)] in
// 1023 bins of 2^sh ordered-float ulps + one overflow bin (bf16 heads, thr = 0.01: sh = 16, one bin per bf16 value)
void hist_window(float thr, u32* base, u32* shift) {
  const u32 lo = ord_f32(thr);
  const float top = thr < 0.5f ? 1.0f : (thr > 0.f ? 2.0f * thr : 1.0f);
  const u32 hi = ord_f32(top) > lo ? ord_f32(top) : lo + 1u;
  u32 sh = 0;
  while (((hi - lo) >> sh) >= kHistBins - 1) ++sh;
  *shift = sh;
  *base = (lo >> sh) << sh;
}

static int check_levels(const ssdk_level* lv, int L) {
  for (int l = 0; l < L; ++l)
    if (!lv[l].cls || !lv[l].box || ((uintptr_t)lv[l].cls & 15)) {
      set_error("decode: level %d has a null or non-16-byte-aligned head pointer", l);
      return SSDK_E_BADARG;
    }
  return SSDK_OK;
}

// scan_kernel: cand[B][units_per_image][K] sorted keys + cand_cnt[B][units_per_image] into `ws`
int launch_scan(const ssdk_level* lv, int L, int B, int dtype, float thr, int K, const DecodePlan& pl, void* ws,
                size_t ws_bytes, hipStream_t stream, unsigned long long* stamps) {
  int rc = check_levels(lv, L);
  if (rc) return rc;
  if (!ws || ws_bytes < pl.cand_bytes + pl.cnt_bytes || ((uintptr_t)ws & 15)) {
    set_error("decode: workspace too small or misaligned (%zu < %zu)", ws_bytes, pl.cand_bytes + pl.cnt_bytes);
    return SSDK_E_WORKSPACE;
  }
  ScanParams sp;
  memset(&sp, 0, sizeof(sp));
  for (int l = 0; l < L; ++l) {
    sp.lv[l].cls = lv[l].cls;
    sp.lv[l].n = pl.n[l];
    sp.lv[l].units = pl.units[l];
    sp.lv[l].unit_base = pl.unit_base[l];
    sp.lv[l].tpu = pl.tpu[l];
  }
  sp.L = L;
  sp.units_per_image = pl.units_per_image;
  sp.B = (u32)B;
  sp.K = (u32)K;
  sp.thr = thr;
  sp.fast = env_int("SSDK_SCAN_FAST", 1) != 0 && thr == thr;
  hist_window(thr, &sp.hist_base, &sp.hist_sh);
  sp.cand = (u64*)ws;
  sp.cand_cnt = (u32*)((char*)ws + pl.cand_bytes);
  sp.stamps = stamps;
  const dim3 grid((unsigned)(B * pl.units_per_image));
  lds_poison(stream);
  if (scan16_applies(dtype, thr, K)) {  // 16-bit heads, positive threshold: packed compares, vector buffers (ssdk_scan16.hip)
    bool fits = true;
    for (int l = 0; l < L; ++l) fits = fits && pl.tpu[l] <= 255u;
    if (fits) {
      sp.thr16 = scan16_threshold_pattern(dtype, thr);
      sp.inf16 = dtype == SSDK_BF16 ? 0x7f80u : 0x7c00u;
      return launch_scan16(sp, dtype, B, pl.units_per_image, stream);
    }
  }
  const int pf = env_int("SSDK_SCAN_PF", 4) == 8 ? 8 : 4;  // 16-byte requests in flight per lane
  const size_t lds = scan_lds_bytes((u32)K, pf);
  auto go = [&](auto kern) {
    if (lds > 64 * 1024)  // above the default dynamic-LDS limit (PF = 8): raise it for this instantiation
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(kScanThreads), lds, stream, sp);
  };
  if (pf == 8) {
    if (dtype == SSDK_F32) go(scan_kernel<SSDK_F32, 8>);
    else if (dtype == SSDK_BF16) go(scan_kernel<SSDK_BF16, 8>);
    else go(scan_kernel<SSDK_F16, 8>);
  } else {
    if (dtype == SSDK_F32) go(scan_kernel<SSDK_F32, 4>);
    else if (dtype == SSDK_BF16) go(scan_kernel<SSDK_BF16, 4>);
    else go(scan_kernel<SSDK_F16, 4>);
  }
  return check_launch("scan_kernel");
}

// level_kernel on the scan's output: the zero-padded [B, L*K] per-level decode of box.decode
int launch_level(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const DecodePlan& pl, const void* ws,
                 float* scores, float* boxes, float* classes, hipStream_t stream) {
  if (!scores || !boxes || !classes) {
    set_error("decode: null output pointer");
    return SSDK_E_BADARG;
  }
  LevelParams lp;
  memset(&lp, 0, sizeof(lp));
  for (int l = 0; l < L; ++l) {
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
  lp.L = L;
  lp.dtype = dtype;
  lp.rescore = rescore;
  lp.units_per_image = pl.units_per_image;
  lp.K = (u32)K;
  lp.out_stride = (u32)(L * K);
  lp.cand = (const u64*)ws;
  lp.cand_cnt = (const u32*)((const char*)ws + pl.cand_bytes);
  lp.scores = scores;
  lp.boxes = boxes;
  lp.classes = classes;
  lds_poison(stream);
  hipLaunchKernelGGL(level_kernel, dim3((unsigned)L, (unsigned)B), dim3(kLevelThreads), lds_bytes_for((u32)K), stream, lp);
  return check_launch("level_kernel");
}

}  // namespace ssdk

extern "C" size_t ssdk_decode_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype, int top_n) {
  ssdk::DecodePlan pl;
  if (ssdk::make_plan(levels, L, B, dtype, top_n, &pl, 0)) return 0;
  return pl.cand_bytes + pl.cnt_bytes;
}

extern "C" int ssdk_decode(const ssdk_level* level, int B, int dtype, float threshold, int top_n,
                           int rescore, float* scores, float* boxes, float* classes, void* workspace,
                           size_t workspace_bytes, void* stream) {
  ssdk::DecodePlan pl;
  int rc = ssdk::make_plan(level, 1, B, dtype, top_n, &pl, 0);
  if (rc) return rc;
  rc = ssdk::launch_scan(level, 1, B, dtype, threshold, top_n, pl, workspace, workspace_bytes, (hipStream_t)stream,
                         nullptr);
  if (rc) return rc;
  return ssdk::launch_level(level, 1, B, dtype, top_n, rescore, pl, workspace, scores, boxes, classes,
                            (hipStream_t)stream);
}
