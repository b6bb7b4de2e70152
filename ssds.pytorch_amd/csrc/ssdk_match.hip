// ssdk_match.hip -- ground-truth <-> anchor matching, box encoding and one-hot targets on gfx950.
//
// Replaces box.extract_targets + box.snap_to_anchors_by_iou (reference box.py:362-405, 116-226): the
// reference loops over the B images in Python and issues ~35 ATen ops per (image, level), building
// [A*W*H, G] temporaries.  Here the whole batch of one level is ONE launch: a thread owns one grid
// anchor (a, y, x), walks the image's valid ground-truth rows (staged in LDS, order preserved so that
// the first maximum wins like torch.max on CPU), and writes its column of the three target tensors.
// Write-bound: (A*C + A*4 + A)*H*W*4 bytes per image.
//
// by_scale = 1 selects box.snap_to_anchors_by_scale (reference box.py:229-359, FCOS-style): a ground-truth box
// is a candidate for a grid point when the point lies inside it (or inside its centre region) and the box's
// regression range (or sqrt-area) falls in [lo, hi] * sqrt(anchor area); the smallest candidate wins.
#include "ssdk_conv_common.h"

namespace ssdk {

constexpr int kMatchThreads = 256;

struct MatchParams {
  const float* targets;  // [B, G, 5]
  int G, A, C, H, W, stride;
  float hi, lo, radius_px;  // radius_px = float(stride * radius); 0 = off
  int use_radius;
  int by_scale;  // 0: IoU matching (hi/lo = match/unmatch thresholds); 1: scale-range matching (lo/hi multipliers)
  float anchors[SSDK_MAX_ANCHORS * 4];
  float* cls_target;  // [B, A, C, H, W]
  float* box_target;  // [B, A, 4, H, W]
  float* depth;       // [B, A, 1, H, W]
  // fused loss mode (match_loss_kernel): logits in, gradients + per-workgroup partial sums out
  const void* conf;  // [B, A, C, H, W] logits
  const void* loc;   // [B, A, 4, H, W]
  void* d_conf;      // same shapes / dtype: d(sum cls loss)/d conf, d(sum loc loss)/d loc
  void* d_loc;
  float* partial;    // [gridDim.y * gridDim.x, 3]: cls sum, loc sum, #foreground
  float alpha, gamma, beta;
  int loc_loss;  // 0: smooth-L1(beta); 1..4: IoU / GIoU / DIoU / CIoU on the deltas (criterion.py:154-239)
  // MultiBoxLoss mode (LOSS = 2): hardness keys of the negatives, [B, A*H*W] words, and the label the one-hot target of every
  // anchor carries (C = none), [B, A*H*W] halfwords: a mined negative is NOT always an all-zero target -- with IoU matching
  // and center_sampling_radius > 0 an anchor that overlaps a box by >= the match threshold but lies outside the sampling
  // region has depth 0 (box.py:183-191) and keeps its one-hot class target (box.py:195-207)
  u32* keys;
  u16* labs;
};

struct GtRow {
  float x1, y1, x2, y2, area, label, pad0, pad1;
};

template <int DT>
__device__ __forceinline__ float ld_elem(const void* p, size_t i) {
  if constexpr (DT == SSDK_F32) return ((const float*)p)[i];
  else return bits16_to_f32<DT>(((const u16*)p)[i]);
}
template <int DT>
__device__ __forceinline__ void st_elem(void* p, size_t i, float v) {
  if constexpr (DT == SSDK_F32) ((float*)p)[i] = v;
  else ((u16*)p)[i] = f32_to_bits16<DT>(v);
}

// ---- forward-mode differentiation of the IoU-family losses ------------------------------------------------------
// value + the four partials w.r.t. the predicted deltas (cx, cy, log w, log h); the rules are those of the torch ops
// the reference composes (criterion.py:173-239): max/min split the gradient on ties, clamp passes it inside [lo, hi]
// (NaN falls through, like torch.clamp), the (lt < rb) masks and CIoU's alpha are constants.
struct D4 {
  float v, g[4];
};
__device__ __forceinline__ D4 d_cst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 d_var(float x, int k) {
  D4 r = d_cst(x);
  r.g[k] = 1.f;
  return r;
}
#define SSDK_D4(expr_v, expr_g) \
  D4 r;                         \
  r.v = (expr_v);               \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) r.g[k] = (expr_g); \
  return r;
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { SSDK_D4(a.v + b.v, a.g[k] + b.g[k]) }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { SSDK_D4(a.v - b.v, a.g[k] - b.g[k]) }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { SSDK_D4(a.v * b.v, a.g[k] * b.v + b.g[k] * a.v) }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
  SSDK_D4(a.v / b.v, (a.g[k] * b.v - b.g[k] * a.v) / (b.v * b.v))
}
__device__ __forceinline__ D4 d_scale(const D4& a, float c) { SSDK_D4(a.v * c, a.g[k] * c) }
__device__ __forceinline__ D4 d_exp(const D4& a) {
  const float e = expf(a.v);
  SSDK_D4(e, a.g[k] * e)
}
__device__ __forceinline__ D4 d_atan(const D4& a) {
  const float q = 1.0f / (1.0f + a.v * a.v);
  SSDK_D4(atanf(a.v), a.g[k] * q)
}
__device__ __forceinline__ D4 d_max(const D4& a, const D4& b) {
  if (a.v > b.v) return a;
  if (b.v > a.v) return b;
  SSDK_D4(a.v, 0.5f * (a.g[k] + b.g[k]))
}
__device__ __forceinline__ D4 d_min(const D4& a, const D4& b) {
  if (a.v < b.v) return a;
  if (b.v < a.v) return b;
  SSDK_D4(a.v, 0.5f * (a.g[k] + b.g[k]))
}
__device__ __forceinline__ D4 d_clamp(const D4& a, float lo, float hi) {
  if (a.v < lo) return d_cst(lo);
  if (a.v > hi) return d_cst(hi);
  return a;
}
#undef SSDK_D4

// 1 - IoU-family overlap of the boxes encoded by the predicted and the target deltas; kind 1..4 = iou, giou, diou, ciou
__device__ __forceinline__ D4 iou_family_loss(const float (&pd)[4], const float (&td)[4], int kind) {
  constexpr float kEps = 1e-7f;
  const D4 px = d_var(pd[0], 0), py = d_var(pd[1], 1);
  const D4 pw = d_exp(d_var(pd[2], 2)), ph = d_exp(d_var(pd[3], 3));  // criterion.py:226-231 delta2ltrb
  const D4 tx = d_cst(td[0]), ty = d_cst(td[1]), tw = d_cst(expf(td[2])), th = d_cst(expf(td[3]));
  const D4 pl = px - d_scale(pw, 0.5f), pr = px + d_scale(pw, 0.5f), pt = py - d_scale(ph, 0.5f), pb = py + d_scale(ph, 0.5f);
  const D4 tl = tx - d_scale(tw, 0.5f), tr = tx + d_scale(tw, 0.5f), tt = ty - d_scale(th, 0.5f), tb = ty + d_scale(th, 0.5f);
  const D4 l = d_max(pl, tl), r = d_min(pr, tr), t = d_max(pt, tt), b = d_min(pb, tb);  // :183-184
  const float m = (l.v < r.v && t.v < b.v) ? 1.0f : 0.0f;
  const D4 inter = d_scale((r - l) * (b - t), m);                                        // :186
  const D4 uni = pw * ph + tw * th - inter;                                              // :187-190
  const D4 iou = (inter + d_cst(kEps)) / (uni + d_cst(kEps));                            // :191
  D4 q;
  if (kind == 1) {
    q = d_clamp(iou, 0.0f, 1.0f);  // :193-195
  } else {
    const D4 ol = d_min(pl, tl), orr = d_max(pr, tr), ot = d_min(pt, tt), ob = d_max(pb, tb);  // :197-198
    if (kind == 2) {
      const float mo = (ol.v < orr.v && ot.v < ob.v) ? 1.0f : 0.0f;
      const D4 hull = d_scale((orr - ol) * (ob - ot), mo) + d_cst(kEps);  // :201-205
      q = d_clamp(iou - (hull - uni) / hull, -1.0f, 1.0f);                // :206-208
    } else {
      const D4 dx = px - tx, dy = py - ty, ow = orr - ol, oh = ob - ot;
      D4 pen = (dx * dx + dy * dy) / (ow * ow + oh * oh + d_cst(kEps));  // :210-211
      if (kind == 4) {                                                   // :218-231
        const D4 da = d_atan(tw / th) - d_atan(pw / ph);
        const D4 v = d_scale(da * da, 0.40528473456935116f);  // 4 / pi^2
        const float alpha = v.v / (1.0f - iou.v + v.v);       // constant for the gradient (torch.no_grad)
        pen = pen + d_scale(v, alpha);
      }
      q = d_clamp(iou - pen, -1.0f, 1.0f);  // :213-216, :232-234
    }
  }
  D4 out;
  out.v = 1.0f - q.v;
#pragma unroll
  for (int k = 0; k < 4; ++k) out.g[k] = -q.g[k];
  return out;
}

// LOSS = 0: write the three target tensors.  LOSS = 1: never materialise them -- evaluate the focal and smooth-L1
// terms of this anchor against the logits (criterion.py:74-151, masks of pipeline_anchor_apex.py:55-66), write the
// closed-form gradients and reduce (cls sum, loc sum, #foreground) per workgroup in a fixed order.
// LOSS = 2: the class term is MultiBoxLoss (criterion.py:43-71): sigmoid cross entropy; the positives' terms and gradients
// are final here, every other gradient is written as zero and each negative (depth == 0) leaves its hardness
// (max over classes of its terms, :59-61) as an ordered key for mine_select_kernel / mine_apply_kernel.
template <int LOSS, int DT>
__global__ __launch_bounds__(kMatchThreads) void match_kernel(const MatchParams p) {
  __shared__ GtRow gt[SSDK_MAX_GT];
  __shared__ float s_anchor[SSDK_MAX_ANCHORS * 4];
  __shared__ u32 wcnt[kMatchThreads / 64];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.y;
  const int G = p.G;

  // stage + compact the valid rows (label > -1, box.py:375), order preserved
  if (tid < SSDK_MAX_ANCHORS * 4) s_anchor[tid] = p.anchors[tid];
  float r[5] = {0.f, 0.f, 0.f, 0.f, -1.f};
  bool valid = false;
  if ((int)tid < G) {
    const float* t = p.targets + ((size_t)b * G + tid) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = t[k];
    valid = r[4] > -1.0f;
  }
  const u64 m = __ballot(valid);
  if (lane == 0) wcnt[wave] = (u32)__popcll(m);
  __syncthreads();
  u32 off = 0, ng = 0;
#pragma unroll
  for (int w = 0; w < kMatchThreads / 64; ++w) {
    if (w < (int)wave) off += wcnt[w];
    ng += wcnt[w];
  }
  if (valid) {
    GtRow g;
    g.x1 = r[0];
    g.y1 = r[1];
    g.x2 = r[0] + r[2] - 1.0f;  // box.py:162
    g.y2 = r[1] + r[3] - 1.0f;
    g.area = (g.x2 - g.x1 + 1.0f) * (g.y2 - g.y1 + 1.0f);  // box.py:166
    g.label = r[4];
    g.pad0 = sqrtf(g.area);  // box.py:287 (scale matching compares sqrt-areas)
    g.pad1 = 0.f;
    gt[off + mbcnt(m)] = g;
  }
  __syncthreads();

  const u32 HW = (u32)(p.H * p.W);
  const u32 total = (u32)p.A * HW;
  const u32 t = blockIdx.x * kMatchThreads + tid;
  float s_cls = 0.f, s_loc = 0.f, s_fg = 0.f;  // LOSS: this anchor's terms
  if (t < total) {
  const u32 a = t / HW;
  const u32 yx = t % HW;
  const u32 iy = yx / (u32)p.W, ix = yx % (u32)p.W;

  const size_t cls_i = (((size_t)b * p.A + a) * p.C) * HW + yx;
  const size_t box_i = (((size_t)b * p.A + a) * 4) * HW + yx;
  float* cls_o = p.cls_target + cls_i;
  float* box_o = p.box_target + box_i;
  float* dep_o = p.depth + ((size_t)b * p.A + a) * HW + yx;

  float dep = 0.f, delta[4] = {0.f, 0.f, 0.f, 0.f};
  int lab = p.C;
  if (ng != 0) {  // (ng == 0: box.py:133-146, all-zero targets)

  const float fx = (float)(ix * (u32)p.stride), fy = (float)(iy * (u32)p.stride);  // box.py:151-156
  const float ax1 = fx + s_anchor[a * 4 + 0], ay1 = fy + s_anchor[a * 4 + 1];     // box.py:157-159
  const float ax2 = fx + s_anchor[a * 4 + 2], ay2 = fy + s_anchor[a * 4 + 3];
  const float aarea = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);                    // box.py:167

  float best = 0.f;
  u32 bi = 0;
  bool inside = false;
  const float apx = fx + (float)(p.stride / 2), apy = fy + (float)(p.stride / 2);  // box.py:185, 281
  if (p.by_scale) {
    const float asz = sqrtf((s_anchor[a * 4 + 2] - s_anchor[a * 4 + 0] + 1.0f) *
                            (s_anchor[a * 4 + 3] - s_anchor[a * 4 + 1] + 1.0f));  // box.py:265-266
    const float lo = tmax(p.lo * asz, -1.0f), hi = p.hi * asz;                     // box.py:267-268
    for (u32 g = 0; g < ng; ++g) {
      const GtRow q = gt[g];
      bool cared, in;
      if (p.use_radius) {  // box.py:290-299 (get_sample_region is called with its default radius 1.5)
        cared = (q.pad0 >= lo) && (q.pad0 <= hi);
        const float cx = (q.x1 + q.x2) / 2.0f, cy = (q.y1 + q.y2) / 2.0f;
        const float lx = apx - tmax(cx - p.radius_px, q.x1), ly = apy - tmax(cy - p.radius_px, q.y1);
        const float rx = tmin(cx + p.radius_px, q.x2) - apx, ry = tmin(cy + p.radius_px, q.y2) - apy;
        in = tmin(tmin(lx, ly), tmin(rx, ry)) > 0.f;
      } else {  // box.py:300-311
        const float l = apx - q.x1, t2 = apy - q.y1, r2 = q.x2 - apx, b2 = q.y2 - apy;
        const float mx = tmax(tmax(l, t2), tmax(r2, b2));
        cared = (mx >= lo) && (mx <= hi);
        in = tmin(tmin(l, t2), tmin(r2, b2)) > 0.f;
      }
      const bool cand = cared && in;
      const float v = cand ? q.pad0 : 100000.0f;  // box.py:5,316-318: INF for non-candidates
      inside = inside || cand;
      if (g == 0 || v < best) {  // box.py:320: first minimum wins
        best = v;
        bi = g;
      }
    }
  } else {
    for (u32 g = 0; g < ng; ++g) {
      const GtRow q = gt[g];
      const float x1 = tmax(ax1, q.x1), y1 = tmax(ay1, q.y1);  // box.py:163-165
      const float x2 = tmin(ax2, q.x2), y2 = tmin(ay2, q.y2);
      float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
      w = tmax(w, 0.f);
      h = tmax(h, 0.f);
      const float inter = w * h;
      const float ov = inter / (aarea + q.area - inter);  // box.py:168 (no epsilon)
      if (g == 0 || ov > best) {  // box.py:171: first maximum wins
        best = ov;
        bi = g;
      }
      if (p.use_radius) {  // box.py:90-113 get_sample_region
        const float cx = (q.x1 + q.x2) / 2.0f, cy = (q.y1 + q.y2) / 2.0f;
        const float lx = apx - tmax(cx - p.radius_px, q.x1), ly = apy - tmax(cy - p.radius_px, q.y1);
        const float rx = tmin(cx + p.radius_px, q.x2) - apx, ry = tmin(cy + p.radius_px, q.y2) - apy;
        const float mn = tmin(tmin(lx, ly), tmin(rx, ry));
        inside = inside || (mn > 0.f);
      }
    }
  }
  const GtRow q = gt[bi];
  // box.py:61-71 box2delta(boxes[indices], anchors)
  const float aw = ax2 - ax1 + 1.0f, ah = ay2 - ay1 + 1.0f;
  const float acx = ax1 + 0.5f * aw, acy = ay1 + 0.5f * ah;
  const float bw = q.x2 - q.x1 + 1.0f, bh = q.y2 - q.y1 + 1.0f;
  const float bcx = q.x1 + 0.5f * bw, bcy = q.y1 + 0.5f * bh;
  delta[0] = (bcx - acx) / aw;
  delta[1] = (bcy - acy) / ah;
  delta[2] = logf(bw / aw);
  delta[3] = logf(bh / ah);

  dep = -1.0f;  // box.py:177-182
  if (p.by_scale) {  // box.py:328-330, 340: no ignore band
    dep = inside ? q.label + 1.0f : 0.f;
    lab = inside ? (int)(long long)q.label : p.C;
  } else {
    if (best < p.lo) dep = 0.f;
    if (best >= p.hi) dep = q.label + 1.0f;
    if (p.use_radius) dep = tmin(dep, inside ? 1.0f : 0.f);  // box.py:191
    // box.py:195-207: one-hot at the matched label unless background (overlap < unmatch threshold)
    lab = (best < p.lo) ? p.C : (int)(long long)q.label;
  }
  }  // ng != 0

  if constexpr (LOSS == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) box_o[(size_t)k * HW] = delta[k];
    dep_o[0] = dep;
    for (int c = 0; c < p.C; ++c) cls_o[(size_t)c * HW] = (c == lab) ? 1.0f : 0.f;
  } else {
    // focal loss on logits (criterion.py:74-108), masked by depth >= 0 (pipeline_anchor_apex.py:55-58):
    //   t=1: L = -alpha (1-p)^g log p        dL/dz = alpha (1-p)^g (g p log p - (1-p))
    //   t=0: L = -(1-alpha) p^g log(1-p)     dL/dz = (1-alpha) p^g (p - g (1-p) log(1-p))
    if constexpr (LOSS == 2) {
      const bool pos = dep > 0.f;
      float hard = 0.f;
      for (int c = 0; c < p.C; ++c) {
        const size_t i = cls_i + (size_t)c * HW;
        const float z = ld_elem<DT>(p.conf, i);
        const float e = expf(-fabsf(z));
        const float inv = 1.0f / (1.0f + e);
        const float pr = z >= 0.f ? inv : e * inv;                                  // sigmoid(z)
        const float ce = tmax(z, 0.f) - (c == lab ? z : 0.f) + log1pf(e);           // BCE with logits, criterion.py:56
        hard = tmax(hard, ce);
        s_cls += pos ? ce : 0.f;
        st_elem<DT>(p.d_conf, i, pos ? pr - (c == lab ? 1.0f : 0.f) : 0.f);
      }
      // ce >= 0, so the float bits order like the values; +1 keeps 0 for "never ranks" (positives, ignored: :61)
      p.keys[(size_t)b * total + t] = dep == 0.f ? __float_as_uint(hard) + 1u : 0u;
      p.labs[(size_t)b * total + t] = (u16)lab;
    } else {
    const bool care = dep >= 0.f;
    const bool g2 = p.gamma == 2.0f;
    for (int c = 0; c < p.C; ++c) {
      const size_t i = cls_i + (size_t)c * HW;
      const float z = ld_elem<DT>(p.conf, i);
      const float e = __expf(-fabsf(z));                 // exp(-|z|) in (0, 1]
      const float inv = 1.0f / (1.0f + e);
      const float pr = z >= 0.f ? inv : e * inv;         // sigmoid(z)
      const float qr = z >= 0.f ? e * inv : inv;         // 1 - sigmoid(z), no cancellation
      const float l1p = log1pf(e);
      const float logp = -(l1p + (z < 0.f ? -z : 0.f));  // log sigmoid(z)
      const float logq = -(l1p + (z > 0.f ? z : 0.f));   // log (1 - sigmoid(z))
      const bool pos = c == lab;
      const float u = pos ? qr : pr;                     // 1 - p_t
      const float v = pos ? pr : qr;                     // p_t
      const float lg = pos ? logp : logq;                // log p_t
      const float w = pos ? p.alpha : 1.0f - p.alpha;
      const float ug = g2 ? u * u : (u > 0.f ? __powf(u, p.gamma) : 0.f);
      const float loss = -w * ug * lg;
      float grad = w * ug * (p.gamma * v * lg - u);      // d/dz for t=1; the t=0 case is its mirror image
      grad = pos ? grad : -grad;
      s_cls += care ? loss : 0.f;
      st_elem<DT>(p.d_conf, i, care ? grad : 0.f);
    }
    }
    // localisation loss masked by depth > 0 (pipeline_anchor_apex.py:62-66)
    const bool fg = dep > 0.f;
    if (p.loc_loss != 0) {  // IoU family (criterion.py:154-239): one term per anchor
      float pd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pd[k] = ld_elem<DT>(p.loc, box_i + (size_t)k * HW);
      if (fg) {
        const D4 r = iou_family_loss(pd, delta, p.loc_loss);
        s_loc += r.v;
#pragma unroll
        for (int k = 0; k < 4; ++k) st_elem<DT>(p.d_loc, box_i + (size_t)k * HW, r.g[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) st_elem<DT>(p.d_loc, box_i + (size_t)k * HW, 0.f);
      }
    } else {
    // smooth-L1 (criterion.py:111-151)
    const float rb = 1.0f / p.beta;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = box_i + (size_t)k * HW;
      const float d = ld_elem<DT>(p.loc, i) - delta[k];
      const float x = fabsf(d);
      const bool lin = x >= p.beta;
      const float loss = lin ? x - 0.5f * p.beta : 0.5f * x * x * rb;
      const float grad = lin ? (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f)) : d * rb;
      s_loc += fg ? loss : 0.f;
      st_elem<DT>(p.d_loc, i, fg ? grad : 0.f);
    }
    }
    s_fg = fg ? 1.0f : 0.f;
  }
  }  // t < total

  if constexpr (LOSS != 0) {  // fixed-order workgroup reduction -> one partial row per workgroup
    __shared__ float red[kMatchThreads / 64][3];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      s_cls += __shfl_xor(s_cls, o);
      s_loc += __shfl_xor(s_loc, o);
      s_fg += __shfl_xor(s_fg, o);
    }
    if (lane == 0) {
      red[wave][0] = s_cls;
      red[wave][1] = s_loc;
      red[wave][2] = s_fg;
    }
    __syncthreads();
    if (tid < 3) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < kMatchThreads / 64; ++w) acc += red[w][tid];
      p.partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + tid] = acc;
    }
  }
}

// sums the per-workgroup rows in index order (one workgroup, fixed tree): sums[0..2] = cls, loc, #foreground
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* partial, int rows, float* sums) {
  __shared__ float red[4][3];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int r = (int)tid; r < rows; r += 256) {
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += partial[(size_t)r * 3 + k];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] += __shfl_xor(acc[k], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (tid < 3) sums[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}


// ---- MultiBoxLoss hard-negative mining (criterion.py:58-71) -----------------------------------------------------------
// The reference sorts the hardness of all A*H*W anchors of an image twice (sort, then sort of the permutation) to get every
// anchor's rank and keeps rank < num_neg = min(negpos_ratio * #positives, A*H*W - 1).  Only the num_neg-th largest key is
// needed: one workgroup per image finds it with a 4 x 8-bit radix select over the keys match_kernel<2> left behind, then
// turns the keys into 0/1 flags in place (key > T, or key == T for the first `need` ties in index order; torch's sort is
// not stable, so which of several EQUAL negatives it keeps is unspecified there too).
constexpr int kMineThreads = 1024;

__global__ __launch_bounds__(kMineThreads) void mine_select_kernel(u32* keys, int N, const float* partial, int rows_per_image,
                                                                   float negpos_ratio) {
  __shared__ u32 hist[256];
  __shared__ float s_red[kMineThreads / 64];
  __shared__ u32 s_wave[kMineThreads / 64];
  __shared__ u32 s_pick[2];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.x;
  u32* key = keys + (size_t)b * N;

  // #positives of this image = the foreground column of its partial rows (exact: counts stay far below 2^24)
  float np = 0.f;
  for (int r = (int)tid; r < rows_per_image; r += kMineThreads) np += partial[((size_t)b * rows_per_image + r) * 3 + 2];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) np += __shfl_xor(np, o);
  if (lane == 0) s_red[wave] = np;
  __syncthreads();
  np = 0.f;
#pragma unroll
  for (int w = 0; w < kMineThreads / 64; ++w) np += s_red[w];
  // ranks r = 0, 1, ... with r < min(ratio * #pos, N - 1)   (:66-68; the ratio may be fractional)
  const float lim = tmin(negpos_ratio * np, (float)(N - 1));
  u32 remaining = lim > 0.f ? (u32)ceilf(lim) : 0u;

  u32 prefix = 0, mask = 0;
  if (remaining != 0) {
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = (int)tid; i < N; i += kMineThreads) {
        const u32 k = key[i];
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (wave == 0) {  // bins 255 .. 0 in four rounds of 64: the first bin where the running count reaches `remaining`
        u32 above = 0;
        bool done = false;
        for (int round = 0; round < 4 && !done; ++round) {
          const u32 bin = 255u - (u32)(round * 64) - lane;
          const u32 c = hist[bin];
          u32 inc = c;  // inclusive prefix over lanes (descending bins)
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const u32 v = __shfl_up(inc, o);
            if ((int)lane >= o) inc += v;
          }
          const u64 hit = __ballot(above + inc >= remaining);
          if (hit != 0) {
            const u32 first = (u32)__ffsll((long long)hit) - 1u;
            if (lane == first) {
              s_pick[0] = bin;
              s_pick[1] = remaining - (above + inc - c);
            }
            done = true;
          }
          above += __shfl(inc, 63);
        }
      }
      __syncthreads();
      prefix |= s_pick[0] << shift;
      mask |= 255u << shift;
      remaining = s_pick[1];
      __syncthreads();
    }
  }
  const u32 T = prefix, need = remaining;  // remaining == 0 with prefix == 0: nothing is mined

  u32 running = 0;
  for (int base = 0; base < N; base += kMineThreads) {
    const int i = base + (int)tid;
    const u32 k = i < N ? key[i] : 0u;
    const bool live = k != 0u && (remaining != 0 || prefix != 0);
    const bool tie = live && k == T;
    const u64 m = __ballot(tie);
    if (lane == 0) s_wave[wave] = (u32)__popcll(m);
    __syncthreads();
    u32 before = running, all = 0;
#pragma unroll
    for (int w = 0; w < kMineThreads / 64; ++w) {
      const u32 c = s_wave[w];
      before += w < (int)wave ? c : 0u;
      all += c;
    }
    const u32 rank = before + mbcnt(m);
    if (i < N) key[i] = (live && (k > T || (tie && rank < need))) ? 1u : 0u;
    running += all;
    __syncthreads();
  }
}

// the mined negatives' terms: BCE with logits against the anchor's one-hot target (criterion.py:56, 69-71) -- all zeros for a
// background anchor (L = softplus(z), dL/dz = sigmoid(z)), one-hot at `lab` for an anchor that matched a box but fell outside
// the centre-sampling region (depth 0 with a class target: box.py:183-207)
template <int DT>
__global__ __launch_bounds__(kMatchThreads) void mine_apply_kernel(const MatchParams p, float* partial) {
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 b = blockIdx.y;
  const u32 HW = (u32)(p.H * p.W);
  const u32 total = (u32)p.A * HW;
  const u32 t = blockIdx.x * kMatchThreads + tid;
  float s_cls = 0.f;
  if (t < total && p.keys[(size_t)b * total + t] != 0u) {
    const u32 a = t / HW, yx = t % HW;
    const size_t cls_i = (((size_t)b * p.A + a) * p.C) * HW + yx;
    const int lab = (int)p.labs[(size_t)b * total + t];
    for (int c = 0; c < p.C; ++c) {
      const size_t i = cls_i + (size_t)c * HW;
      const float z = ld_elem<DT>(p.conf, i);
      const float e = expf(-fabsf(z));
      const float inv = 1.0f / (1.0f + e);
      s_cls += tmax(z, 0.f) - (c == lab ? z : 0.f) + log1pf(e);
      st_elem<DT>(p.d_conf, i, (z >= 0.f ? inv : e * inv) - (c == lab ? 1.0f : 0.f));
    }
  }
  __shared__ float red[kMatchThreads / 64];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s_cls += __shfl_xor(s_cls, o);
  if (lane == 0) red[wave] = s_cls;
  __syncthreads();
  if (tid < 3) {
    float acc = 0.f;
    if (tid == 0) {
#pragma unroll
      for (int w = 0; w < kMatchThreads / 64; ++w) acc += red[w];
    }
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + tid] = acc;
  }
}

}  // namespace ssdk

namespace ssdk {
static int launch_match(const char* what, int by_scale, const float* targets, int B, int G, const float* anchors,
                        int A, int C, int H, int W, int stride, float hi, float lo, float radius,
                        float* cls_target, float* box_target, float* depth, void* stream) {
  if (!targets || !anchors || !cls_target || !box_target || !depth) {
    set_error("%s: null pointer", what);
    return SSDK_E_BADARG;
  }
  if (B < 1 || G < 0 || G > SSDK_MAX_GT || A < 1 || A > SSDK_MAX_ANCHORS || C < 1 || H < 1 || W < 1 ||
      stride < 1) {
    set_error("%s: bad dims B=%d G=%d (<=%d) A=%d (<=%d) C=%d H=%d W=%d stride=%d", what, B, G,
              SSDK_MAX_GT, A, SSDK_MAX_ANCHORS, C, H, W, stride);
    return SSDK_E_BADARG;
  }
  MatchParams p;
  memset(&p, 0, sizeof(p));
  p.targets = targets;
  p.G = G;
  p.A = A;
  p.C = C;
  p.H = H;
  p.W = W;
  p.stride = stride;
  p.hi = hi;
  p.lo = lo;
  p.by_scale = by_scale;
  p.use_radius = radius > 0.f;
  p.radius_px = (float)((double)stride * (double)radius);
  memcpy(p.anchors, anchors, sizeof(float) * 4 * A);
  p.cls_target = cls_target;
  p.box_target = box_target;
  p.depth = depth;
  const unsigned total = (unsigned)A * H * W;
  dim3 grid((total + kMatchThreads - 1) / kMatchThreads, (unsigned)B);
  hipLaunchKernelGGL((match_kernel<0, SSDK_F32>), grid, dim3(kMatchThreads), 0, (hipStream_t)stream, p);
  return check_launch("match_kernel");
}
}  // namespace ssdk

extern "C" int ssdk_match_targets(const float* targets, int B, int G, const float* anchors, int A, int C,
                                  int H, int W, int stride, float match_threshold, float unmatch_threshold,
                                  float center_sampling_radius, float* cls_target, float* box_target,
                                  float* depth, void* stream) {
  return ssdk::launch_match("match_targets", 0, targets, B, G, anchors, A, C, H, W, stride, match_threshold,
                            unmatch_threshold, center_sampling_radius, cls_target, box_target, depth, stream);
}

extern "C" int ssdk_match_targets_by_scale(const float* targets, int B, int G, const float* anchors, int A,
                                           int C, int H, int W, int stride, float lower_scale,
                                           float upper_scale, int center_sampling, float* cls_target,
                                           float* box_target, float* depth, void* stream) {
  // the reference calls get_sample_region with its default radius (1.5) whatever the configured value is
  return ssdk::launch_match("match_targets_by_scale", 1, targets, B, G, anchors, A, C, H, W, stride,
                            upper_scale, lower_scale, center_sampling ? 1.5f : 0.f, cls_target, box_target,
                            depth, stream);
}

// ---- fused target assignment + losses -----------------------------------------------------------
namespace ssdk {
static size_t match_rows(int B, int A, int H, int W) {
  return (((size_t)A * H * W + kMatchThreads - 1) / kMatchThreads) * (size_t)B;
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// cls_kind 0: FocalLoss(alpha, gamma); 1: MultiBoxLoss(negpos_ratio = alpha)
static int run_match_loss(const char* what, int cls_kind, const float* targets, int B, int G, const float* anchors, int A,
                          int C, int H, int W, int stride, int by_scale, float thr_a, float thr_b, float radius,
                          const void* conf, const void* loc, int dtype, float alpha, float gamma, float beta, int loc_loss,
                          void* d_conf, void* d_loc, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
  if (!targets || !anchors || !conf || !loc || !d_conf || !d_loc || !sums || !workspace) {
    set_error("%s: null pointer", what);
    return SSDK_E_BADARG;
  }
  if (B < 1 || G < 0 || G > SSDK_MAX_GT || A < 1 || A > SSDK_MAX_ANCHORS || C < 1 || H < 1 || W < 1 ||
      stride < 1 || loc_loss < 0 || loc_loss > 4 || (loc_loss == 0 && !(beta > 0.f))) {
    set_error("%s: bad dims B=%d G=%d (<=%d) A=%d (<=%d) C=%d H=%d W=%d stride=%d beta=%g loc_loss=%d", what, B, G,
              SSDK_MAX_GT, A, SSDK_MAX_ANCHORS, C, H, W, stride, (double)beta, loc_loss);
    return SSDK_E_BADARG;
  }
  if (cls_kind == 1 && !(alpha >= 0.f)) {
    set_error("%s: negpos_ratio %g", what, (double)alpha);
    return SSDK_E_BADARG;
  }
  if (dtype != SSDK_F32 && dtype != SSDK_BF16 && dtype != SSDK_F16) {
    set_error("%s: dtype %d not supported", what, dtype);
    return SSDK_E_BADARG;
  }
  if (cls_kind == 1 && C > 65534) {  // (the mined pass reads every anchor's target label from a 16-bit side array)
    set_error("%s: C=%d classes (<= 65534)", what, C);
    return SSDK_E_BADARG;
  }
  const size_t need = cls_kind == 1 ? ssdk_match_multibox_loss_workspace_bytes(B, A, H, W)
                                    : ssdk_match_loss_workspace_bytes(B, A, H, W);
  if (workspace_bytes < need) {
    set_error("%s: workspace too small (%zu < %zu)", what, workspace_bytes, need);
    return SSDK_E_BADARG;
  }
  MatchParams p;
  memset(&p, 0, sizeof(p));
  p.targets = targets;
  p.G = G;
  p.A = A;
  p.C = C;
  p.H = H;
  p.W = W;
  p.stride = stride;
  p.by_scale = by_scale ? 1 : 0;
  if (by_scale) {  // thr_a / thr_b = lower / upper scale multipliers, radius != 0 = centre sampling (radius 1.5)
    p.lo = thr_a;
    p.hi = thr_b;
    radius = radius > 0.f ? 1.5f : 0.f;
  } else {  // thr_a / thr_b = match / unmatch IoU thresholds
    p.hi = thr_a;
    p.lo = thr_b;
  }
  p.use_radius = radius > 0.f;
  p.radius_px = (float)((double)stride * (double)radius);
  memcpy(p.anchors, anchors, sizeof(float) * 4 * A);
  p.conf = conf;
  p.loc = loc;
  p.d_conf = d_conf;
  p.d_loc = d_loc;
  p.partial = (float*)workspace;
  p.alpha = alpha;
  p.gamma = gamma;
  p.beta = beta;
  p.loc_loss = loc_loss;
  const unsigned total = (unsigned)A * H * W;
  dim3 grid((total + kMatchThreads - 1) / kMatchThreads, (unsigned)B);
  const size_t rows = (size_t)grid.x * grid.y;
  hipStream_t st = (hipStream_t)stream;
  if (cls_kind == 0) {
    if (dtype == SSDK_BF16) hipLaunchKernelGGL((match_kernel<1, SSDK_BF16>), grid, dim3(kMatchThreads), 0, st, p);
    else if (dtype == SSDK_F16) hipLaunchKernelGGL((match_kernel<1, SSDK_F16>), grid, dim3(kMatchThreads), 0, st, p);
    else hipLaunchKernelGGL((match_kernel<1, SSDK_F32>), grid, dim3(kMatchThreads), 0, st, p);
    int rc = check_launch("match_loss_kernel");
    if (rc != SSDK_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)p.partial, (int)rows, sums);
    return check_launch("loss_finalize_kernel");
  }
  // MultiBoxLoss: rows [0, rows) of the partial table come from the match pass (positives, loc, #fg), rows [rows, 2 rows)
  // from the mined negatives; the keys follow
  float* partial2 = p.partial + rows * 3;
  p.keys = (u32*)((char*)workspace + align256(2 * rows * 3 * sizeof(float)));
  p.labs = (u16*)((char*)p.keys + align256((size_t)B * total * sizeof(u32)));
  if (dtype == SSDK_BF16) hipLaunchKernelGGL((match_kernel<2, SSDK_BF16>), grid, dim3(kMatchThreads), 0, st, p);
  else if (dtype == SSDK_F16) hipLaunchKernelGGL((match_kernel<2, SSDK_F16>), grid, dim3(kMatchThreads), 0, st, p);
  else hipLaunchKernelGGL((match_kernel<2, SSDK_F32>), grid, dim3(kMatchThreads), 0, st, p);
  int rc = check_launch("match_multibox_kernel");
  if (rc != SSDK_OK) return rc;
  hipLaunchKernelGGL(mine_select_kernel, dim3((unsigned)B), dim3(kMineThreads), 0, st, p.keys, (int)total,
                     (const float*)p.partial, (int)grid.x, alpha);
  rc = check_launch("mine_select_kernel");
  if (rc != SSDK_OK) return rc;
  if (dtype == SSDK_BF16) hipLaunchKernelGGL((mine_apply_kernel<SSDK_BF16>), grid, dim3(kMatchThreads), 0, st, p, partial2);
  else if (dtype == SSDK_F16) hipLaunchKernelGGL((mine_apply_kernel<SSDK_F16>), grid, dim3(kMatchThreads), 0, st, p, partial2);
  else hipLaunchKernelGGL((mine_apply_kernel<SSDK_F32>), grid, dim3(kMatchThreads), 0, st, p, partial2);
  rc = check_launch("mine_apply_kernel");
  if (rc != SSDK_OK) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)p.partial, (int)(2 * rows), sums);
  return check_launch("loss_finalize_kernel");
}
}  // namespace ssdk

extern "C" size_t ssdk_match_loss_workspace_bytes(int B, int A, int H, int W) {
  if (B < 1 || A < 1 || H < 1 || W < 1) return 0;
  return ssdk::match_rows(B, A, H, W) * 3 * sizeof(float);
}

extern "C" size_t ssdk_match_multibox_loss_workspace_bytes(int B, int A, int H, int W) {
  if (B < 1 || A < 1 || H < 1 || W < 1) return 0;
  return ssdk::align256(2 * ssdk::match_rows(B, A, H, W) * 3 * sizeof(float)) +
         ssdk::align256((size_t)B * A * H * W * sizeof(unsigned)) + (size_t)B * A * H * W * sizeof(unsigned short);
}

extern "C" int ssdk_match_loss(const float* targets, int B, int G, const float* anchors, int A, int C, int H,
                               int W, int stride, int by_scale, float thr_a, float thr_b, float radius,
                               const void* conf, const void* loc, int dtype, float alpha, float gamma,
                               float beta, int loc_loss, void* d_conf, void* d_loc, float* sums, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return ssdk::run_match_loss("match_loss", 0, targets, B, G, anchors, A, C, H, W, stride, by_scale, thr_a, thr_b, radius,
                              conf, loc, dtype, alpha, gamma, beta, loc_loss, d_conf, d_loc, sums, workspace,
                              workspace_bytes, stream);
}

extern "C" int ssdk_match_multibox_loss(const float* targets, int B, int G, const float* anchors, int A, int C, int H,
                                        int W, int stride, int by_scale, float thr_a, float thr_b, float radius,
                                        const void* conf, const void* loc, int dtype, float negpos_ratio, float beta,
                                        int loc_loss, void* d_conf, void* d_loc, float* sums, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return ssdk::run_match_loss("match_multibox_loss", 1, targets, B, G, anchors, A, C, H, W, stride, by_scale, thr_a, thr_b,
                              radius, conf, loc, dtype, negpos_ratio, 0.f, beta, loc_loss, d_conf, d_loc, sums, workspace,
                              workspace_bytes, stream);
}
