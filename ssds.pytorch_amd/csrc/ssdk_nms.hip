// ssdk_nms.hip -- greedy class-aware (D)IoU NMS on gfx950, one workgroup per image.
//
// Replaces box.nms (reference ssds/modeling/layers/box.py:480-546): per image `nonzero, sort` and then
// up to `ndetections` dependent iterations of ~20 tiny ATen ops with 5 host syncs each.
//
// The reference's loop is the standard greedy NMS in score order, truncated to the first `ndetections`
// survivors.  Here: all waves build 64-bit keys (score bits | ~position) for the
// candidates with score > 0 (box.py:496; NaN drops out too) and bitonic-sort them in LDS (box.py:505,
// stable order contract); then ONE wave walks the sorted list in blocks of 64 candidates:
//   1. every lane tests its candidate against the survivors kept so far (skipping, wave-uniformly,
//      survivors whose class no lane shares),
//   2. the block is resolved in order with a scalar loop over its 64 lanes (readlane pivot),
//   3. the block's survivors are appended to the kept list and written straight to the output.
// IoU / DIoU arithmetic follows box.py:518-533 in fp32 without FMA contraction (+1 pixel convention,
// eps 1e-7, DIoU penalty = squared TOP-LEFT corner distance / squared outer diagonal), so keep sets are
// bit-exact with the CPU reference.
#include "ssdk_common.h"
#include "ssdk_select.h"
#include "ssdk_decode.h"

namespace ssdk {

constexpr int kNmsThreads = 256;

struct NmsParams {
  const float* scores;
  const float* boxes;
  const float* classes;
  int N, ndet, diou;
  float thr;
  float* out_scores;
  float* out_boxes;
  float* out_classes;
};

// suppression test of candidate (box b, area ab) by pivot (box p, area ap): returns true when the
// candidate must be dropped, i.e. NOT (iou <= thr)   (box.py:518-533)
__device__ __forceinline__ bool suppressed_by(const float4 b, float ab, const float4 p, float ap,
                                              float thr, int diou) {
  const float ix1 = tmax(b.x, p.x), iy1 = tmax(b.y, p.y);
  const float ix2 = tmin(b.z, p.z), iy2 = tmin(b.w, p.w);
  float w = ix2 - ix1 + 1.0f, h = iy2 - iy1 + 1.0f;
  w = tmax(w, 0.0f);  // clamp(0) keeps NaN like torch
  h = tmax(h, 0.0f);
  const float inter = w * h;
  const float iou = inter / (ab + ap - inter + 1e-7f);
  bool over = !(iou <= thr);
  // DIoU = clamp(IoU - penalty, -1, 1) with penalty >= 0: a pair with IoU <= thr can never exceed thr, so the
  // penalty (a second division) is only evaluated for the few pairs that overlap enough (finite boxes).
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

// broadcast of lane j's value when j is wave-uniform: v_readlane (scalar path) instead of ds_bpermute
__device__ __forceinline__ float bcast(float v, u32 j) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)j));
}

constexpr u32 kNmsRound = 256;  // candidates sorted + walked per round

// Lazy sort: the greedy walk almost always stops (ndetections survivors) inside the first few hundred
// candidates, so instead of sorting all L*K keys the workgroup repeatedly extracts the exact top-256 of the
// remaining keys (radix select), sorts those 256 and lets one wave walk them; further rounds only run
// while fewer than ndetections boxes survived and candidates remain.  Result identical to a full sort.
__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsParams p) {
  constexpr int NT = kNmsThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 NW = NT / 64;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.x;
  const u32 N = (u32)p.N;
  u64* keys = reinterpret_cast<u64*>(smem);                        // N (unsorted; taken keys are zeroed)
  u64* top = keys + ((N + 1) & ~1u);                               // kNmsRound
  u64* rows = top + kNmsRound;                                     // 64 suppression rows of the current block
  u64* blk_dead = rows + 64;                                       // (+1 pad) dead bits from the kept-list test
  float4* kbox = reinterpret_cast<float4*>(rows + 66);             // ndet
  float* karea = reinterpret_cast<float*>(kbox + p.ndet);          // ndet
  float* kcls = karea + p.ndet;                                    // ndet
  u32* ctl = reinterpret_cast<u32*>(kcls + p.ndet);                // [0] nvalid, [1] nk, [2] top count
  SelScratch* ss = reinterpret_cast<SelScratch*>(ctl + 4);

  const float* sc = p.scores + (size_t)b * N;
  const float4* bx = reinterpret_cast<const float4*>(p.boxes) + (size_t)b * N;
  const float* cl = p.classes + (size_t)b * N;
  const u32 ndet = (u32)p.ndet;
  float* os = p.out_scores + (size_t)b * ndet;
  float4* ob = reinterpret_cast<float4*>(p.out_boxes) + (size_t)b * ndet;
  float* oc = p.out_classes + (size_t)b * ndet;
  const float thr = p.thr;
  const int diou = p.diou;

  if (tid == 0) {
    ctl[0] = 0;
    ctl[1] = 0;
    *blk_dead = 0ull;
  }
  __syncthreads();
  u32 local = 0;
  for (u32 i = tid; i < N; i += NT) {
    u64 k = 0;
    const float s = sc[i];
    if (s > 0.0f) {  // box.py:496 (NaN drops out too)
      k = make_key(s, i);
      ++local;
    }
    keys[i] = k;
  }
  if (local) atomicAdd(&ctl[0], local);
  __syncthreads();
  u32 left = ctl[0];
  u32 nk = 0;

  while (left > 0 && nk < ndet) {  // workgroup-uniform
    const u32 r = left < kNmsRound ? left : kNmsRound;
    u64 T = 1;  // every remaining (non-zero) key
    if (left > r) T = wg_select_kth<NT>(keys, N, r, ss);  // r-th largest of the remaining keys (box.py:505)
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
    for (u32 i = tid; i < N; i += NT) {
      const u64 k = keys[i];
      if (k >= T) {
        top[atomicAdd(&ctl[2], 1u)] = k;
        keys[i] = 0;
      }
    }
    __syncthreads();
    for (u32 i = r + tid; i < kNmsRound; i += NT) top[i] = 0;
    __syncthreads();
    wg_bitonic_sort_desc<NT>(top, kNmsRound);  // descending key == (score desc, position asc)

    // Greedy walk over the sorted round, 64 candidates (one per lane) at a time.  Every wave holds the same
    // block; the pair tests are split across the waves and only the final in-order resolve is serial:
    //   1. wave w tests the block against kept survivors k = w, w+NW, ...        -> dead bits (LDS OR)
    //   2. wave w builds the suppression rows of pivots j in [w*64/NW, (w+1)*64/NW): row[j] = lanes i > j of
    //      the same class that j would suppress                                   -> rows[] (LDS)
    //   3. wave 0 walks the alive bits in order applying rows of pivots that are still alive -- exactly the
    //      reference's sequential loop (box.py:505-530), minus the arithmetic.
    for (u32 base = 0; base < r && nk < ndet; base += 64) {  // workgroup-uniform
      const u32 i = base + lane;
      const bool valid = i < r;
      float score = 0.f, cls = -1.f, area = 0.f;
      float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const u64 k = top[i];
        const u32 pos = key_index(k);
        score = key_score(k);
        box = bx[pos];
        cls = cl[pos];
        area = (box.z - box.x + 1.0f) * (box.w - box.y + 1.0f);  // box.py:507
      }
      bool alive = valid;
      for (u32 k = wave; k < nk; k += NW) {
        const float ck = kcls[k];
        if (__ballot(alive && cls == ck) == 0ull) continue;
        const bool sup = (cls == ck) && suppressed_by(box, area, kbox[k], karea[k], thr, diou);
        alive = alive && !sup;
      }
      const u64 dead = __ballot(valid && !alive);
      if (lane == 0 && dead) atomicOr(blk_dead, dead);
      constexpr u32 PPW = 64 / NW;
      for (u32 jj = 0; jj < PPW; ++jj) {
        const u32 j = wave * PPW + jj;
        const float cj = bcast(cls, j);
        const u64 m = __ballot(valid && lane > j && cls == cj);
        u64 row = 0;
        if (m != 0ull) {
          float4 pj;
          pj.x = bcast(box.x, j);
          pj.y = bcast(box.y, j);
          pj.z = bcast(box.z, j);
          pj.w = bcast(box.w, j);
          const float aj = bcast(area, j);
          row = __ballot(((m >> lane) & 1ull) && suppressed_by(box, area, pj, aj, thr, diou));
        }
        if (lane == 0) rows[j] = row;
      }
      __syncthreads();
      if (wave == 0) {
        u64 am = __ballot(valid) & ~*blk_dead;
        const u64 myrow = rows[lane];
        const u32 row_lo = (u32)myrow, row_hi = (u32)(myrow >> 32);
        u32 kept_here = 0;
        u64 todo = am;
        while (todo) {
          const u32 j = (u32)__ffsll((long long)todo) - 1u;
          if (nk + kept_here >= ndet) {  // truncated to ndetections survivors (box.py:512)
            am &= (1ull << j) - 1ull;
            break;
          }
          ++kept_here;
          const u64 rj = (u64)(u32)__builtin_amdgcn_readlane((int)row_lo, (int)j) |
                         ((u64)(u32)__builtin_amdgcn_readlane((int)row_hi, (int)j) << 32);
          am &= ~rj;
          todo = am & ~((2ull << j) - 1ull);  // alive candidates after j
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
          ctl[1] = nk2;
          *blk_dead = 0ull;
        }
      }
      __syncthreads();
      nk = ctl[1];
    }
    left -= r;
  }
  // zero padding (box.py:489-491)
  for (u32 i = nk + tid; i < ndet; i += NT) {
    os[i] = 0.f;
    ob[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    oc[i] = 0.f;
  }
}

static size_t nms_lds_bytes(int N, int ndet) {
  return (size_t)((N + 1) & ~1) * 8 + kNmsRound * 8 + 66 * 8 + (size_t)ndet * (16 + 4 + 4) + 16 + sizeof(SelScratch) + 16;
}

int launch_nms(const float* scores, const float* boxes, const float* classes, int B, int N,
                      float thr, int ndet, int diou, float* os, float* ob, float* oc, hipStream_t stream) {
  if (!scores || !boxes || !classes || !os || !ob || !oc) {
    set_error("nms: null pointer");
    return SSDK_E_BADARG;
  }
  if (B < 1 || N < 1 || N > SSDK_MAX_NMS_N || ndet < 1 || ndet > SSDK_MAX_NDET) {
    set_error("nms: B=%d N=%d ndetections=%d out of range (N<=%d, ndet<=%d)", B, N, ndet,
              SSDK_MAX_NMS_N, SSDK_MAX_NDET);
    return SSDK_E_BADARG;
  }
  if (((uintptr_t)boxes & 15) || ((uintptr_t)ob & 15)) {
    set_error("nms: box pointers must be 16-byte aligned");
    return SSDK_E_BADARG;
  }
  NmsParams p;
  p.scores = scores;
  p.boxes = boxes;
  p.classes = classes;
  p.N = N;
  p.ndet = ndet;
  p.diou = diou;
  p.thr = thr;
  p.out_scores = os;
  p.out_boxes = ob;
  p.out_classes = oc;
  lds_poison(stream);
  hipLaunchKernelGGL(nms_kernel, dim3((unsigned)B), dim3(kNmsThreads), nms_lds_bytes(N, ndet), stream, p);
  return check_launch("nms_kernel");
}

}  // namespace ssdk

extern "C" size_t ssdk_nms_workspace_bytes(int B, int N, int ndetections) {
  (void)B;
  (void)N;
  (void)ndetections;
  return 0;  // everything lives in LDS
}

extern "C" int ssdk_nms(const float* scores, const float* boxes, const float* classes, int B, int N,
                        float nms_threshold, int ndetections, int using_diou, float* out_scores,
                        float* out_boxes, float* out_classes, void* workspace, size_t workspace_bytes,
                        void* stream) {
  (void)workspace;
  (void)workspace_bytes;
  return ssdk::launch_nms(scores, boxes, classes, B, N, nms_threshold, ndetections, using_diou,
                          out_scores, out_boxes, out_classes, (hipStream_t)stream);
}

