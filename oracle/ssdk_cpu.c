/* oracle/ssdk_cpu.c -- the box math of the detection hot path behind the SAME C-ABI symbols as libssdk.so, in plain C
 * on the CPU (last sentence: "same symbols compiled for CPU (libssdk_cpu.so) for tests").
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_build/libssdk_cpu.so and loaded by
 * tests/test_oracle_c.py; nothing under ssds.pytorch_amd/ links, loads or calls it, and the product has no CPU
 * fallback.  It follows the reference line by line (ssds/modeling/layers/box.py:46-58, 61-87, 90-113, 116-226,
 * 229-359, 362-405, 408-477, 480-546; decoder.py:25-49), each function citing the lines it restates; parity is pinned
 * by the reference-generated fixtures in tests/golden/ (the same ones that pin oracle/box_oracle.py) and by seeded
 * comparisons with the numpy oracle.
 *
 * Conventions (shared with the numpy oracle and the HIP kernels): IEEE fp32 arithmetic in the reference's operation
 * order, no FMA contraction (-ffp-contract=off); bf16 / fp16 heads are widened to fp32 first; top-k and sort ties
 * resolve as (score descending, flat index ascending); max over ground-truth boxes resolves to the first maximum;
 * min / max / clip propagate NaN the way torch and numpy do.
 * Every pointer is HOST memory here and `stream` / `workspace` are ignored (workspace queries return 0). */
#include "../include/ssdk.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int ssdk_version(void) { return SSDK_VERSION; }
const char* ssdk_last_error(void) { return g_err; }

/* ---- scalar helpers ------------------------------------------------------------------------------------------- */
static float np_min(float a, float b) { return a != a ? a : (b != b ? b : (a < b ? a : b)); } /* NaN propagates */
static float np_max(float a, float b) { return a != a ? a : (b != b ? b : (a > b ? a : b)); }

static float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        ++e;
        man <<= 1;
      } while (!(man & 0x400u));
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | man << 13;
  } else {
    bits = sign | (exp + 127 - 15) << 23 | man << 13;
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static float load_as_f32(const void* p, int dtype, size_t i) {
  if (dtype == SSDK_F32) return ((const float*)p)[i];
  const uint16_t h = ((const uint16_t*)p)[i];
  if (dtype == SSDK_BF16) {
    const uint32_t bits = (uint32_t)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
  }
  return half_to_float(h);
}

/* ---- box.py:46-58 generate_anchors ------------------------------------------------------------------------------ */
int ssdk_generate_anchors(int stride, const float* ratios, int nr, const float* scales, int ns, float* out) {
  if (!ratios || !scales || !out || nr < 1 || ns < 1 || nr * ns > SSDK_MAX_ANCHORS || stride < 1)
    return fail(SSDK_E_BADARG, "generate_anchors: bad arguments");
  const float wh = (float)stride; /* :53 */
  for (int s = 0; s < ns; ++s)    /* scale-major, ratio-minor (:49-51) */
    for (int r = 0; r < nr; ++r) {
      const float ratio = ratios[r], scale = scales[s];
      const float ws = rintf(sqrtf(wh * wh / ratio)); /* :54 torch.round = half to even */
      const float hs = rintf(ws * ratio);             /* :55 */
      float* o = out + (size_t)(s * nr + r) * 4;
      o[0] = 0.5f * (wh - ws * scale); /* :56 */
      o[1] = 0.5f * (wh - hs * scale);
      o[2] = 0.5f * (wh + ws * scale) - 1.0f; /* :57 */
      o[3] = 0.5f * (wh + hs * scale) - 1.0f;
    }
  return SSDK_OK;
}

/* ---- box.py:61-71 box2delta / box.py:74-87 delta2box (one box) ----------------------------------------------------- */
static void box2delta(const float* box, const float* anc, float* d) {
  const float aw = anc[2] - anc[0] + 1.0f, ah = anc[3] - anc[1] + 1.0f;
  const float acx = anc[0] + 0.5f * aw, acy = anc[1] + 0.5f * ah;
  const float bw = box[2] - box[0] + 1.0f, bh = box[3] - box[1] + 1.0f;
  const float bcx = box[0] + 0.5f * bw, bcy = box[1] + 0.5f * bh;
  d[0] = (bcx - acx) / aw;
  d[1] = (bcy - acy) / ah;
  d[2] = logf(bw / aw);
  d[3] = logf(bh / ah);
}

static float clampf(float t, float hi) { return np_max(0.0f, np_min(t, hi)); } /* :83 clamp(min=0, max=size*stride-1) */

static void delta2box(const float* d, const float* anc, int W, int H, int stride, float* box) {
  const float aw = anc[2] - anc[0] + 1.0f, ah = anc[3] - anc[1] + 1.0f;
  const float cx = anc[0] + 0.5f * aw, cy = anc[1] + 0.5f * ah;
  const float pcx = d[0] * aw + cx, pcy = d[1] * ah + cy;
  const float pw = expf(d[2]) * aw, ph = expf(d[3]) * ah;
  const float mx = (float)W * (float)stride - 1.0f, my = (float)H * (float)stride - 1.0f;
  box[0] = clampf(pcx - 0.5f * pw, mx);
  box[1] = clampf(pcy - 0.5f * ph, my);
  box[2] = clampf(pcx + 0.5f * pw - 1.0f, mx);
  box[3] = clampf(pcy + 0.5f * ph - 1.0f, my);
}

/* ---- box.py:408-477 decode, one level -------------------------------------------------------------------------------- */
typedef struct {
  float score;
  int64_t idx;
} cand_t;

static int cand_cmp(const void* a, const void* b) { /* score descending, flat index ascending */
  const cand_t* x = (const cand_t*)a;
  const cand_t* y = (const cand_t*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* out_*: rows of `row_stride` slots per image, this level's top_n slots starting at `col0` (zero filled by the caller) */
static int decode_level(const ssdk_level* lv, int B, int dtype, float thr, int top_n, int rescore, float* scores,
                        float* boxes, float* classes, int row_stride, int col0) {
  const int A = lv->A, C = lv->C, H = lv->H, W = lv->W, stride = lv->stride;
  const int64_t n = (int64_t)A * C * H * W;
  cand_t* cand = (cand_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(cand_t));
  if (!cand) return fail(SSDK_E_WORKSPACE, "decode: out of memory");
  for (int b = 0; b < B; ++b) { /* :435 */
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
      const float s = load_as_f32(lv->cls, dtype, (size_t)b * n + i);
      if (s >= thr) { /* :440 (NaN fails the comparison) */
        cand[m].score = s;
        cand[m].idx = i;
        ++m;
      }
    }
    if (m == 0) continue;
    qsort(cand, (size_t)m, sizeof(cand_t), cand_cmp); /* :446 topk */
    const int k = (int)(m < top_n ? m : top_n);
    for (int j = 0; j < k; ++j) {
      const int64_t i = cand[j].idx;
      const int c = (int)((i / W / H) % C); /* :448 */
      const int x = (int)(i % W);           /* :452 */
      const int y = (int)((i / W) % H);
      const int a = (int)(i / C / H / W);
      float d[4], g[4], bx[4];
      for (int t = 0; t < 4; ++t) /* :455-456 box.view(A, 4, H, W)[a, :, y, x] */
        d[t] = load_as_f32(lv->box, dtype, ((size_t)b * A * 4 + (size_t)a * 4 + t) * H * W + (size_t)y * W + x);
      const float fx = (float)x * (float)stride, fy = (float)y * (float)stride; /* :459-462 */
      g[0] = fx + lv->anchors[a * 4 + 0];
      g[1] = fy + lv->anchors[a * 4 + 1];
      g[2] = fx + lv->anchors[a * 4 + 2];
      g[3] = fy + lv->anchors[a * 4 + 3];
      delta2box(d, g, W, H, stride, bx);
      float s = cand[j].score;
      if (rescore) { /* :464-471 */
        const float gcx = (g[0] + g[2]) / 2.0f, gcy = (g[1] + g[3]) / 2.0f;
        const float l = fabsf(gcx - bx[0]), t = fabsf(gcy - bx[1]), r = fabsf(bx[2] - gcx), bt = fabsf(bx[3] - gcy);
        const float rx = np_min(l, r) / np_max(l, r), ry = np_min(t, bt) / np_max(t, bt);
        s = s * sqrtf(rx * ry);
      }
      const size_t o = (size_t)b * row_stride + col0 + j;
      scores[o] = s;
      classes[o] = (float)c;
      memcpy(boxes + o * 4, bx, 4 * sizeof(float));
    }
  }
  free(cand);
  return SSDK_OK;
}

static int level_ok(const ssdk_level* lv) {
  return lv && lv->cls && lv->box && lv->A >= 1 && lv->A <= SSDK_MAX_ANCHORS && lv->C >= 1 && lv->H >= 1 && lv->W >= 1 && lv->stride >= 1;
}

size_t ssdk_decode_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype, int top_n) {
  (void)levels, (void)L, (void)B, (void)dtype, (void)top_n;
  return 0;
}

int ssdk_decode(const ssdk_level* level, int B, int dtype, float threshold, int top_n, int rescore, float* scores,
                float* boxes, float* classes, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace, (void)workspace_bytes, (void)stream;
  if (!level_ok(level) || !scores || !boxes || !classes || B < 1 || top_n < 1 || top_n > SSDK_MAX_TOPN || dtype < SSDK_F32 || dtype > SSDK_F16)
    return fail(SSDK_E_BADARG, "decode: bad arguments");
  memset(scores, 0, (size_t)B * top_n * sizeof(float)); /* :430-432 zero padded */
  memset(boxes, 0, (size_t)B * top_n * 4 * sizeof(float));
  memset(classes, 0, (size_t)B * top_n * sizeof(float));
  return decode_level(level, B, dtype, threshold, top_n, rescore, scores, boxes, classes, top_n, 0);
}

/* ---- box.py:480-546 nms --------------------------------------------------------------------------------------------- */
typedef struct {
  float score, cls, box[4], area;
  int order;
} det_t;

static int det_cmp(const void* a, const void* b) { /* :505 sort descending (stable: original position breaks ties) */
  const det_t* x = (const det_t*)a;
  const det_t* y = (const det_t*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

size_t ssdk_nms_workspace_bytes(int B, int N, int ndetections) {
  (void)B, (void)N, (void)ndetections;
  return 0;
}

int ssdk_nms(const float* scores, const float* boxes, const float* classes, int B, int N, float nms_threshold,
             int ndetections, int using_diou, float* out_scores, float* out_boxes, float* out_classes, void* workspace,
             size_t workspace_bytes, void* stream) {
  (void)workspace, (void)workspace_bytes, (void)stream;
  if (!scores || !boxes || !classes || !out_scores || !out_boxes || !out_classes || B < 1 || N < 1 || N > SSDK_MAX_NMS_N ||
      ndetections < 1 || ndetections > SSDK_MAX_NDET)
    return fail(SSDK_E_BADARG, "nms: bad arguments");
  const float eps = 1e-7f;
  memset(out_scores, 0, (size_t)B * ndetections * sizeof(float)); /* :489-491 */
  memset(out_boxes, 0, (size_t)B * ndetections * 4 * sizeof(float));
  memset(out_classes, 0, (size_t)B * ndetections * sizeof(float));
  det_t* d = (det_t*)malloc((size_t)N * sizeof(det_t));
  if (!d) return fail(SSDK_E_WORKSPACE, "nms: out of memory");
  for (int b = 0; b < B; ++b) {
    int m = 0;
    for (int j = 0; j < N; ++j) { /* :496 keep scores > 0 (drops NaN too) */
      const float s = scores[(size_t)b * N + j];
      if (s > 0.0f) {
        d[m].score = s;
        d[m].cls = classes[(size_t)b * N + j];
        memcpy(d[m].box, boxes + ((size_t)b * N + j) * 4, 4 * sizeof(float));
        d[m].order = m;
        ++m;
      }
    }
    if (m == 0) continue; /* :501-502 */
    qsort(d, (size_t)m, sizeof(det_t), det_cmp);
    for (int j = 0; j < m; ++j) /* :507 */
      d[j].area = (d[j].box[2] - d[j].box[0] + 1.0f) * (d[j].box[3] - d[j].box[1] + 1.0f);
    int i;
    for (i = 0; i < ndetections; ++i) { /* :512 */
      if (i >= m) {                     /* :513-515 */
        i -= 1;
        break;
      }
      const det_t p = d[i];
      int w = 0;
      for (int j = 0; j < m; ++j) {
        const det_t q = d[j];
        const float x1 = np_max(q.box[0], p.box[0]), y1 = np_max(q.box[1], p.box[1]); /* :518-519 */
        const float x2 = np_min(q.box[2], p.box[2]), y2 = np_min(q.box[3], p.box[3]);
        const float dx = np_max(x2 - x1 + 1.0f, 0.0f), dy = np_max(y2 - y1 + 1.0f, 0.0f);
        const float inter = dx * dy;
        float iou = inter / (q.area + p.area - inter + eps); /* :521 */
        if (using_diou) {                                    /* :523-530: distance of the TOP-LEFT corners */
          const float olx = np_min(q.box[0], p.box[0]), oly = np_min(q.box[1], p.box[1]);
          const float orx = np_max(q.box[2], p.box[2]), ory = np_max(q.box[3], p.box[3]);
          const float dlx = q.box[0] - p.box[0], dly = q.box[1] - p.box[1];
          const float inter_diag = dlx * dlx + dly * dly;
          const float dox = orx - olx, doy = ory - oly;
          const float outer_diag = (dox * dox + doy * doy) + eps;
          iou = np_max(-1.0f, np_min(iou - inter_diag / outer_diag, 1.0f));
        }
        const int keep = j == i || q.score > p.score || iou <= nms_threshold || q.cls != p.cls; /* :532-533 */
        if (keep) d[w++] = q; /* :536-539 compaction */
      }
      m = w;
    }
    if (i == ndetections) i = ndetections - 1; /* the loop variable after `for i in range(ndetections)` ran out */
    const int cnt = i + 1 < m ? i + 1 : m;     /* :542-544 */
    for (int j = 0; j < cnt; ++j) {
      out_scores[(size_t)b * ndetections + j] = d[j].score;
      out_classes[(size_t)b * ndetections + j] = d[j].cls;
      memcpy(out_boxes + ((size_t)b * ndetections + j) * 4, d[j].box, 4 * sizeof(float));
    }
  }
  free(d);
  return SSDK_OK;
}

/* ---- decoder.py:25-49 Decoder.__call__ --------------------------------------------------------------------------------- */
size_t ssdk_decode_nms_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype, int top_n_per_level, int ndetections) {
  (void)levels, (void)L, (void)B, (void)dtype, (void)top_n_per_level, (void)ndetections;
  return 0;
}

int ssdk_decode_nms(const ssdk_level* levels, int L, int B, int dtype, float threshold, int top_n_per_level, int rescore,
                    float nms_threshold, int ndetections, int using_diou, float* out_scores, float* out_boxes,
                    float* out_classes, float* mid_scores, float* mid_boxes, float* mid_classes, void* workspace,
                    size_t workspace_bytes, void* stream) {
  (void)workspace, (void)workspace_bytes;
  if (!levels || L < 1 || B < 1 || top_n_per_level < 1 || top_n_per_level > SSDK_MAX_TOPN || (long)L * top_n_per_level > SSDK_MAX_NMS_N ||
      dtype < SSDK_F32 || dtype > SSDK_F16)
    return fail(SSDK_E_BADARG, "decode_nms: bad arguments");
  for (int l = 0; l < L; ++l)
    if (!level_ok(levels + l)) return fail(SSDK_E_BADARG, "decode_nms: bad level %d", l);
  const int N = L * top_n_per_level;
  float* ms = mid_scores ? mid_scores : (float*)malloc((size_t)B * N * sizeof(float));
  float* mb = mid_boxes ? mid_boxes : (float*)malloc((size_t)B * N * 4 * sizeof(float));
  float* mc = mid_classes ? mid_classes : (float*)malloc((size_t)B * N * sizeof(float));
  int rc = (ms && mb && mc) ? SSDK_OK : fail(SSDK_E_WORKSPACE, "decode_nms: out of memory");
  if (rc == SSDK_OK) {
    memset(ms, 0, (size_t)B * N * sizeof(float));
    memset(mb, 0, (size_t)B * N * 4 * sizeof(float));
    memset(mc, 0, (size_t)B * N * sizeof(float));
    for (int l = 0; l < L && rc == SSDK_OK; ++l) /* decoder.py:36-47 per level, :48 torch.cat(dim=1) */
      rc = decode_level(levels + l, B, dtype, threshold, top_n_per_level, rescore, ms, mb, mc, N, l * top_n_per_level);
    if (rc == SSDK_OK) /* decoder.py:49 */
      rc = ssdk_nms(ms, mb, mc, B, N, nms_threshold, ndetections, using_diou, out_scores, out_boxes, out_classes, NULL, 0, stream);
  }
  if (!mid_scores) free(ms);
  if (!mid_boxes) free(mb);
  if (!mid_classes) free(mc);
  return rc;
}

/* ---- box.py:362-405 extract_targets over box.py:116-226 / 229-359 --------------------------------------------------------- */
/* box.py:90-113 get_sample_region for one (point, box): the point lies inside the box AND inside the square of
 * half-side stride*radius around the box centre */
static int in_sample_region(const float* box, float half, float px, float py) {
  const float cx = (box[0] + box[2]) / 2.0f, cy = (box[1] + box[3]) / 2.0f;
  const float l = px - np_max(cx - half, box[0]), t = py - np_max(cy - half, box[1]);
  const float r = np_min(cx + half, box[2]) - px, b = np_min(cy + half, box[3]) - py;
  return np_min(np_min(l, t), np_min(r, b)) > 0.0f;
}

typedef struct {
  int by_scale;
  float match_thr, unmatch_thr, radius; /* IoU matching */
  float lower_scale, upper_scale;       /* scale-range matching */
  int center_sampling;
} match_cfg;

static int match_impl(const match_cfg* cfg, const float* targets, int B, int G, const float* anchors, int A, int C, int H, int W,
                      int stride, float* cls_target, float* box_target, float* depth) {
  if (!targets || !anchors || !cls_target || !box_target || !depth || B < 1 || G < 0 || G > SSDK_MAX_GT || A < 1 ||
      A > SSDK_MAX_ANCHORS || C < 1 || H < 1 || W < 1 || stride < 1)
    return fail(SSDK_E_BADARG, "match_targets: bad arguments");
  const size_t HW = (size_t)H * W;
  float(*box)[4] = (float(*)[4])malloc((size_t)(G > 0 ? G : 1) * sizeof *box);
  float* lab = (float*)malloc((size_t)(G > 0 ? G : 1) * sizeof(float));
  float* sarea = (float*)malloc((size_t)(G > 0 ? G : 1) * sizeof(float));
  if (!box || !lab || !sarea) {
    free(box), free(lab), free(sarea);
    return fail(SSDK_E_WORKSPACE, "match_targets: out of memory");
  }
  const float pt_off = (float)(stride / 2); /* anchor point = cell corner + stride // 2 */
  for (int b = 0; b < B; ++b) {
    float* ct = cls_target + (size_t)b * A * C * HW;
    float* bt = box_target + (size_t)b * A * 4 * HW;
    float* dp = depth + (size_t)b * A * HW;
    memset(ct, 0, (size_t)A * C * HW * sizeof(float));
    memset(bt, 0, (size_t)A * 4 * HW * sizeof(float));
    memset(dp, 0, (size_t)A * HW * sizeof(float));
    int g = 0;
    for (int j = 0; j < G; ++j) { /* :375 rows with label <= -1 are padding */
      const float* t = targets + ((size_t)b * G + j) * 5;
      if (t[4] > -1.0f) {
        box[g][0] = t[0]; /* :162 / :284 xywh -> inclusive ltrb */
        box[g][1] = t[1];
        box[g][2] = t[0] + t[2] - 1.0f;
        box[g][3] = t[1] + t[3] - 1.0f;
        lab[g] = t[4];
        const float bw = box[g][2] - box[g][0] + 1.0f, bh = box[g][3] - box[g][1] + 1.0f;
        sarea[g] = cfg->by_scale ? sqrtf(bw * bh) : bw * bh; /* :285 sqrt-area | :165 area */
        ++g;
      }
    }
    if (g == 0) continue; /* :133-146 / :246-260 all zero */
    for (int a = 0; a < A; ++a) {
      const float* an = anchors + (size_t)a * 4;
      const float aw = an[2] - an[0] + 1.0f, ah = an[3] - an[1] + 1.0f;
      const float asize = sqrtf(aw * ah); /* :265-268 */
      const float lower = np_max(cfg->lower_scale * asize, -1.0f), upper = cfg->upper_scale * asize;
      for (int ix = 0; ix < W; ++ix)
        for (int iy = 0; iy < H; ++iy) {
          const float fx = (float)(ix * stride), fy = (float)(iy * stride);
          const float ga[4] = {fx + an[0], fy + an[1], fx + an[2], fy + an[3]}; /* :151-159 */
          const float px = fx + pt_off, py = fy + pt_off;
          int best = 0, matched = 0;
          float best_ov = 0.0f;
          if (!cfg->by_scale) {
            const float garea = (ga[2] - ga[0] + 1.0f) * (ga[3] - ga[1] + 1.0f);
            for (int j = 0; j < g; ++j) { /* :163-171 IoU (+1 convention, no epsilon), first maximum */
              const float x1 = np_max(ga[0], box[j][0]), y1 = np_max(ga[1], box[j][1]);
              const float x2 = np_min(ga[2], box[j][2]), y2 = np_min(ga[3], box[j][3]);
              const float dx = np_max(x2 - x1 + 1.0f, 0.0f), dy = np_max(y2 - y1 + 1.0f, 0.0f);
              const float inter = dx * dy;
              const float ov = inter / (garea + sarea[j] - inter);
              if (j == 0 || ov > best_ov || (ov != ov && best_ov == best_ov)) { /* max: the first NaN wins like torch.max */
                best_ov = ov;
                best = j;
              }
            }
          } else {
            float best_area = 100000.0f; /* :5 INF */
            for (int j = 0; j < g; ++j) { /* :288-321 candidates; the smallest sqrt-area wins, first on ties */
              int cared, inside;
              if (cfg->center_sampling) {
                cared = sarea[j] >= lower && sarea[j] <= upper;
                inside = in_sample_region(box[j], (float)((double)stride * 1.5), px, py);
              } else {
                const float l = px - box[j][0], t = py - box[j][1], r = box[j][2] - px, bb = box[j][3] - py;
                const float mx = np_max(np_max(l, t), np_max(r, bb));
                cared = mx >= lower && mx <= upper;
                inside = np_min(np_min(l, t), np_min(r, bb)) > 0.0f;
              }
              const int ok = cared && inside;
              const float ar = ok ? sarea[j] : 100000.0f;
              matched |= ok;
              if (j == 0 || ar < best_area) {
                best_area = ar;
                best = j;
              }
            }
          }
          const size_t pos = (size_t)iy * W + ix;
          float d4[4];
          box2delta(box[best], ga, d4); /* :172 / :323 */
          for (int t = 0; t < 4; ++t) bt[((size_t)a * 4 + t) * HW + pos] = d4[t];
          float dv;
          int onehot;
          if (!cfg->by_scale) { /* :177-182 */
            dv = -1.0f;
            if (best_ov < cfg->unmatch_thr) dv = 0.0f;
            if (best_ov >= cfg->match_thr) dv = lab[best] + 1.0f;
            if (cfg->radius > 0.0f) { /* :184-191 */
              int any = 0;
              const float half = (float)((double)stride * (double)cfg->radius);
              for (int j = 0; j < g && !any; ++j) any = in_sample_region(box[j], half, px, py);
              dv = np_min(dv, any ? 1.0f : 0.0f);
            }
            onehot = !(best_ov < cfg->unmatch_thr); /* :195-207 background column dropped */
          } else {                                    /* :328-346 */
            dv = matched ? lab[best] + 1.0f : 0.0f;
            onehot = matched;
          }
          dp[(size_t)a * HW + pos] = dv;
          if (onehot) {
            const long c = (long)lab[best];
            if (c >= 0 && c < C) ct[((size_t)a * C + c) * HW + pos] = 1.0f;
          }
        }
    }
  }
  free(box), free(lab), free(sarea);
  return SSDK_OK;
}

int ssdk_match_targets(const float* targets, int B, int G, const float* anchors, int A, int C, int H, int W, int stride,
                       float match_threshold, float unmatch_threshold, float center_sampling_radius, float* cls_target,
                       float* box_target, float* depth, void* stream) {
  (void)stream;
  match_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.match_thr = match_threshold;
  cfg.unmatch_thr = unmatch_threshold;
  cfg.radius = center_sampling_radius;
  return match_impl(&cfg, targets, B, G, anchors, A, C, H, W, stride, cls_target, box_target, depth);
}

int ssdk_match_targets_by_scale(const float* targets, int B, int G, const float* anchors, int A, int C, int H, int W, int stride,
                                float lower_scale, float upper_scale, int center_sampling, float* cls_target, float* box_target,
                                float* depth, void* stream) {
  (void)stream;
  match_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.by_scale = 1;
  cfg.lower_scale = lower_scale;
  cfg.upper_scale = upper_scale;
  cfg.center_sampling = center_sampling;
  return match_impl(&cfg, targets, B, G, anchors, A, C, H, W, stride, cls_target, box_target, depth);
}
