// ssdk_map.hip -- VOC-style mean-average-precision bookkeeping on gfx950.
//
// Replaces MeanAveragePrecision.__call__ / get_results (reference core/evaluation_metrics.py:15-61, 63-142).  The
// reference walks images x classes in Python, computes an IoU matrix per (image, class) and resolves the greedy
// "first detection to claim a ground-truth box" rule in a Python loop; an eval epoch spends its time there.
//
//  map_match_kernel  one workgroup per image: each detection above the score threshold finds the same-class
//                    ground-truth box of highest IoU (first maximum, like argmax on CPU); the detection of lowest
//                    index among those reaching the IoU threshold on a box is the true positive (LDS atomicMin --
//                    integer, order independent), every other detection a false positive.  Emits one record per
//                    detection slot: a 64-bit sort key (class, score descending) and the TP flag; padded /
//                    below-threshold slots get class = num_classes so that they sort behind every real class.
//                    Ground-truth counts per class are integer atomics (deterministic).
//  map_ap_kernel     one wave per class over the records sorted by key: precision at every rank in fp64, its
//                    running maximum from the right (the VOC envelope), summed at the ranks where recall moves.
//
// Sorting the records (once per epoch) is left to the caller (rocPRIM radix sort through torch.sort).
#include "ssdk_common.h"

namespace ssdk {

constexpr int kMapThreads = 256;
constexpr int kMapPer = 8;  // detections per thread: D <= kMapThreads * kMapPer

struct MapParams {
  const float* scores;   // [B, D]
  const float* boxes;    // [B, D, 4] ltrb
  const float* classes;  // [B, D]
  const float* targets;  // [B, G, 5] ltrb + label (label < 0: padding)
  int D, G, C;
  float conf_thr, iou_thr;
  long long* keys;     // [B, D]
  unsigned char* tp;   // [B, D]
  int* npos;           // [C], accumulated
};

__global__ __launch_bounds__(kMapThreads) void map_match_kernel(const MapParams p) {
  __shared__ float gt[SSDK_MAX_GT][5];
  __shared__ int first[SSDK_MAX_GT];
  const int tid = (int)threadIdx.x, b = (int)blockIdx.x;
  for (int g = tid; g < p.G; g += kMapThreads) {
    const float* t = p.targets + ((size_t)b * p.G + g) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) gt[g][k] = t[k];
    first[g] = 0x7fffffff;
    const float lab = t[4];
    if (lab >= 0.f && lab < (float)p.C) atomicAdd(&p.npos[(int)lab], 1);  // evaluation_metrics.py:36,56
  }
  __syncthreads();

  int best_g[kMapPer];
#pragma unroll
  for (int r = 0; r < kMapPer; ++r) {
    best_g[r] = -1;
    const int i = tid + r * kMapThreads;
    if (i >= p.D) continue;
    const size_t o = (size_t)b * p.D + i;
    const float s = p.scores[o];
    if (!(s > p.conf_thr)) continue;  // evaluation_metrics.py:29-31
    const float c = p.classes[o];
    const float x1 = p.boxes[o * 4 + 0], y1 = p.boxes[o * 4 + 1], x2 = p.boxes[o * 4 + 2], y2 = p.boxes[o * 4 + 3];
    const float area_a = (x2 - x1) * (y2 - y1);  // evaluation_metrics.py:24
    float best = 0.f;
    int bg = -1;
    for (int g = 0; g < p.G; ++g) {
      if (gt[g][4] != c) continue;  // evaluation_metrics.py:33
      const float lx = fmaxf(x1, gt[g][0]), ly = fmaxf(y1, gt[g][1]);  // :20-21
      const float rx = fminf(x2, gt[g][2]), ry = fminf(y2, gt[g][3]);
      const float inter = (lx < rx && ly < ry) ? (rx - lx) * (ry - ly) : 0.f;  // :23
      const float area_b = (gt[g][2] - gt[g][0]) * (gt[g][3] - gt[g][1]);      // :25
      const float iou = inter / (area_a + area_b - inter);                     // :26
      if (bg < 0 || iou > best) {  // :49 argmax, first maximum
        best = iou;
        bg = g;
      }
    }
    if (bg >= 0 && best >= p.iou_thr) {  // :54
      best_g[r] = bg;
      atomicMin(&first[bg], i);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kMapPer; ++r) {
    const int i = tid + r * kMapThreads;
    if (i >= p.D) continue;
    const size_t o = (size_t)b * p.D + i;
    const float s = p.scores[o];
    const float c = p.classes[o];
    const bool valid = (s > p.conf_thr) && c >= 0.f && c < (float)p.C;
    const u32 cls = valid ? (u32)c : (u32)p.C;
    // ascending key order = (class ascending, score descending); slots of equal key keep their arrival order
    // under a stable sort
    p.keys[o] = (long long)(((u64)cls << 32) | (u64)(~ord_f32(s)));
    p.tp[o] = (best_g[r] >= 0 && first[best_g[r]] == i) ? 1 : 0;  // :54-56
  }
}

// records sorted by key; seg[c] .. seg[c+1] is class c.  ap[c] = NaN when the class has no ground truth (:124-128),
// 0 when it has no detections (:97-98).
__global__ __launch_bounds__(64) void map_ap_kernel(const unsigned char* tp, const long long* seg, const int* npos,
                                                    double* ap) {
  const int c = (int)blockIdx.x;
  const u32 lane = threadIdx.x;
  const long long s0 = seg[c], n = seg[c + 1] - seg[c];
  const int np = npos[c];
  if (np == 0) {
    if (lane == 0) ap[c] = __builtin_nan("");
    return;
  }
  const double dnp = (double)np;
  long long T = 0;
  for (long long j = 0; j < n; j += 64) {
    const bool t = (j + lane < n) && tp[s0 + j + lane] != 0;
    T += __popcll(__ballot(t));
  }
  double carry = 0.0, acc = 0.0;
  long long tp_ge = 0;  // true positives at ranks >= the current chunk
  for (long long j = ((n - 1) / 64) * 64; j >= 0 && n > 0; j -= 64) {
    const bool in = j + lane < n;
    const bool t = in && tp[s0 + j + lane] != 0;
    const u64 m = __ballot(t);
    tp_ge += __popcll(m);
    const long long tpj = (T - tp_ge) + (long long)__popcll(m & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull)));
    // :136-137  prec = tp / max(tp + fp, eps) with tp + fp = rank + 1
    double pr = in ? (double)tpj / (double)(j + lane + 1) : 0.0;
    pr = pr > carry ? pr : carry;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {  // suffix maximum across the wave (:104-106)
      const double o = __shfl_down(pr, d);
      if (lane + d < 64u) pr = pr > o ? pr : o;
    }
    carry = __shfl(pr, 0);
    // :108-111  recall moves exactly at the true positives
    if (t) acc += ((double)tpj / dnp - (double)(tpj - 1) / dnp) * pr;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) ap[c] = acc;
}

}  // namespace ssdk

extern "C" int ssdk_map_match(const float* scores, const float* boxes, const float* classes, int B, int D,
                              const float* targets, int G, int num_classes, float conf_threshold,
                              float iou_threshold, long long* keys, unsigned char* tp, int* npos, void* stream) {
  using namespace ssdk;
  if (!scores || !boxes || !classes || !targets || !keys || !tp || !npos) {
    set_error("map_match: null pointer");
    return SSDK_E_BADARG;
  }
  if (B < 1 || D < 1 || D > kMapThreads * kMapPer || G < 0 || G > SSDK_MAX_GT || num_classes < 1) {
    set_error("map_match: bad dims B=%d D=%d (<=%d) G=%d (<=%d) classes=%d", B, D, kMapThreads * kMapPer, G,
              SSDK_MAX_GT, num_classes);
    return SSDK_E_BADARG;
  }
  MapParams p;
  p.scores = scores;
  p.boxes = boxes;
  p.classes = classes;
  p.targets = targets;
  p.D = D;
  p.G = G;
  p.C = num_classes;
  p.conf_thr = conf_threshold;
  p.iou_thr = iou_threshold;
  p.keys = keys;
  p.tp = tp;
  p.npos = npos;
  hipLaunchKernelGGL(map_match_kernel, dim3((unsigned)B), dim3(kMapThreads), 0, (hipStream_t)stream, p);
  return check_launch("map_match_kernel");
}

extern "C" int ssdk_map_average_precision(const unsigned char* tp_sorted, const long long* seg_offsets,
                                          const int* npos, int num_classes, double* ap, void* stream) {
  using namespace ssdk;
  if (!tp_sorted || !seg_offsets || !npos || !ap || num_classes < 1) {
    set_error("map_average_precision: bad argument");
    return SSDK_E_BADARG;
  }
  hipLaunchKernelGGL(map_ap_kernel, dim3((unsigned)num_classes), dim3(64), 0, (hipStream_t)stream, tp_sorted,
                     seg_offsets, npos, ap);
  return check_launch("map_ap_kernel");
}
