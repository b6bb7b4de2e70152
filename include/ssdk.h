/*
 * ssdk.h -- C-ABI of the MI355X-native detection hot path of ssds.pytorch.
 *
 * The reference (ShuangXieIrene/ssds.pytorch v1.5) is pure Python; the only native interface it
 * names is the phantom plugin `ssds._C` that it imports but never ships:
 *     ssds/modeling/layers/box.py:3-4        from ssds._C import decode / nms   (commented out)
 *     ssds/modeling/layers/box.py:419-421    decode_cuda(cls_head, box_head, anchors.view(-1).tolist(),
 *                                                        stride, threshold, top_n)
 *     ssds/modeling/layers/box.py:483-485    nms_cuda(scores, boxes, classes, nms, ndetections)
 *     ssds/utils/export.py:134-153           ssds._C (TensorRT; out of scope)
 * This header is what a maintainer's `ssds._C` (ctypes / pybind stub) binds.
 * Every entry point below cites the Python function it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  `stream` is a hipStream_t passed as void*.
 *   - the caller owns every buffer (inputs, outputs, workspace); kernels never allocate, never
 *     synchronise, and fully write their outputs (zero padding included), so every call is
 *     hipGraph-capturable on the caller's stream.
 *   - return value: 0 = ok, negative = error (SSDK_E_*); text via ssdk_last_error() (thread local).
 *   - head tensors are NCHW contiguous, channel = a*C + c (conf) / a*4 + k (loc), exactly the layout
 *     the reference's heads produce (ssd.py:68-70); base pointers must be 16-byte aligned.
 *   - tie order (unspecified in the reference's topk/sort): score descending, flat index ascending.
 *   - all box arithmetic is IEEE fp32 in the reference's operation order without FMA contraction,
 *     inputs of any dtype are upcast to fp32 first; outputs are always fp32 (box.py:430-432).
 *   - descriptor structs (ssdk_conv_desc, ssdk_mbconv_desc, ssdk_xpair_desc, ssdk_op, ...) are ZERO-INITIALISED by the
 *     caller before it sets the fields it knows.  New fields are only ever appended to a struct and 0 / NULL selects
 *     the behaviour from before the field existed; entry points never change their signature (new ones are added and
 *     SSDK_VERSION is raised).
 *   - ABI: a descriptor that GROWS changes sizeof(ssdk_op), which is the array stride of ssdk_run_ops -- so every
 *     SSDK_VERSION step that appends a descriptor field (210, 220) is ABI-BREAKING for callers of the descriptor entry points:
 *     a caller must be compiled against the header of the library it loads.  The library cannot see the caller's layout
 *     through a pointer, so the caller proves it once: ssdk_abi_check(SSDK_VERSION, sizeof(ssdk_op)) right after loading
 *     (0 = the caller's header and the library's agree; the Python host does this in ssds/_native.py and refuses to run
 *     otherwise), or per struct with ssdk_struct_size().  The box entry points (plain pointers and scalars) are unaffected.
 */
#ifndef SSDK_H_
#define SSDK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDK_VERSION 244 /* 0.2.4.4: ssdk_im2col3x3_folded / ssdk_col2im3x3_folded; 0.2.4.3: ssdk_stem3x3s2_fwd / _wgrad (the first convolution of the training step); 0.2.4.2: ssdk_concat_nchw_to_nhwc; 0.2.4.1: ssdk_pack_conv3x3[_dgrad]; 0.2.4: ssdk_abi_check, ssdk_pw_* (1x1 convolutions of the training step: forward / input gradient / weight gradient on NCHW tensors); 0.2.3: ssdk_struct_size; 0.2.2: ssdk_mbconv_desc.image_nw / w_image / w_image_bytes (ssdk_mbk.hip), larger ssdk_match_multibox_loss workspace; 0.2.1: ssdk_match_multibox_loss, ssdk_op.lane == 2; fields appended to descriptors since 200 (zero = old behaviour) */

#define SSDK_MAX_LEVELS 8    /* feature-map levels per decode_nms call            */
#define SSDK_MAX_ANCHORS 16  /* anchors per location (A)                          */
#define SSDK_MAX_TOPN 1024   /* top_n per level (box.decode) / candidates kept    */
#define SSDK_MAX_NDET 1024   /* ndetections (box.nms)                             */
#define SSDK_MAX_NMS_N 8192  /* candidates per image entering nms                 */
#define SSDK_MAX_GT 256      /* ground-truth rows per image (extract_targets)     */

enum {
  SSDK_OK = 0,
  SSDK_E_BADARG = -1,    /* null / misaligned pointer, size out of range          */
  SSDK_E_WORKSPACE = -2, /* workspace too small (see *_workspace_bytes)           */
  SSDK_E_LAUNCH = -3,    /* HIP launch error                                      */
  SSDK_E_NODEVICE = -4   /* no HIP device / wrong architecture                    */
};

typedef enum { SSDK_F32 = 0, SSDK_BF16 = 1, SSDK_F16 = 2, SSDK_U8 = 3 /* ssdk_preprocess source only */ } ssdk_dtype;
typedef enum {
  SSDK_ACT_NONE = 0,
  SSDK_ACT_RELU = 1,
  SSDK_ACT_RELU6 = 2,
  SSDK_ACT_SILU = 3,
  SSDK_ACT_SIGMOID = 4
} ssdk_act;

/* One feature-map level of the multibox head (decoder.py:36-47 zips loc, conf, anchors.items()). */
typedef struct ssdk_level {
  const void* cls; /* device, [B, A*C, H, W] scores (sigmoid probabilities in eval, ssd.py:72-73) */
  const void* box; /* device, [B, A*4, H, W] deltas                                               */
  int32_t A, C, H, W;
  int32_t stride;                       /* key of the anchors OrderedDict (model_builder.py:41)  */
  float anchors[SSDK_MAX_ANCHORS * 4];  /* host values, generate_anchors(stride, ...) ltrb       */
} ssdk_level;

int ssdk_version(void);
/* sizeof() of a descriptor struct AS THE LIBRARY WAS BUILT (version 230): a binding checks its own layout against it at
 * load time instead of trusting the version number alone -- a caller compiled against an older header passes a shorter
 * struct, and the library would read past its end.  which: */
#define SSDK_SIZEOF_LEVEL 0
#define SSDK_SIZEOF_CONV_DESC 1
#define SSDK_SIZEOF_MBCONV_DESC 2
#define SSDK_SIZEOF_FUSE_DESC 3
#define SSDK_SIZEOF_STEM_DESC 4
#define SSDK_SIZEOF_POOL_DESC 5
#define SSDK_SIZEOF_XPAIR_DESC 6
#define SSDK_SIZEOF_OP 7
size_t ssdk_struct_size(int which); /* 0 for an unknown `which` */
/* The caller's view of the ABI against the library's: header_version = the SSDK_VERSION the caller was compiled with,
 * sizeof_op = its sizeof(ssdk_op).  0 if header_version / 10 == the library's version / 10 (the last digit counts additions
 * that leave every layout alone) and the sizes agree; SSDK_E_BADARG (text in ssdk_last_error) otherwise.  (version 240) */
int ssdk_abi_check(int header_version, size_t sizeof_op);
const char* ssdk_last_error(void);
/* name of the kernel the calling thread launched last (which variant a layer was dispatched to; tests, tools) */
const char* ssdk_last_kernel(void);

/* Device facts the host side needs for roofline accounting (bench.py). Returns 0 / SSDK_E_NODEVICE. */
int ssdk_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* arch, int arch_len);

/* box.py:46-58 generate_anchors(stride, ratio_vals, scales_vals) -> out[nr*ns*4] (HOST memory, fp32).
 * Ordering: scale-major, ratio-minor.  Pure host function (runs once per model/image size). */
int ssdk_generate_anchors(int stride, const float* ratios, int nr, const float* scales, int ns,
                          float* out);

/* box.py:408-477 decode(all_cls_head, all_box_head, stride, threshold, top_n, anchors, rescore)
 * for ONE level.  Replaces the phantom ssds._C.decode (box.py:419-421), extended by `rescore` and
 * a dtype.  scores[B*top_n], boxes[B*top_n*4], classes[B*top_n] device fp32, zero padded. */
size_t ssdk_decode_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype, int top_n);
int ssdk_decode(const ssdk_level* level, int B, int dtype, float threshold, int top_n, int rescore,
                float* scores, float* boxes, float* classes, void* workspace,
                size_t workspace_bytes, void* stream);

/* box.py:480-546 nms(all_scores, all_boxes, all_classes, nms, ndetections, using_diou).
 * Replaces the phantom ssds._C.nms (box.py:483-485), extended by `using_diou`.
 * in: scores[B*N], boxes[B*N*4], classes[B*N]; out: [B*ndet], [B*ndet*4], [B*ndet]; device fp32. */
size_t ssdk_nms_workspace_bytes(int B, int N, int ndetections);
int ssdk_nms(const float* scores, const float* boxes, const float* classes, int B, int N,
             float nms_threshold, int ndetections, int using_diou, float* out_scores,
             float* out_boxes, float* out_classes, void* workspace, size_t workspace_bytes,
             void* stream);

/* Contexts.  The two entry points that are more than one launch (ssdk_decode_nms: scan + tail; ssdk_run_ops: a whole
 * recorded network with an optional side lane) need a few HIP objects -- a side stream, fork/join events, optional
 * profiling events.  They live in a caller-owned context, never in library globals: a context belongs to the device
 * that was current at ssdk_ctx_create() and is used by one host thread at a time; two threads, or two devices, use two
 * contexts and share nothing (the library is re-entrant; its only other state is the per-thread error text).  The
 * entry points WITHOUT a context argument use a context that belongs to the calling thread and the current device. */
typedef struct ssdk_ctx ssdk_ctx;
ssdk_ctx* ssdk_ctx_create(void);       /* NULL + ssdk_last_error() when there is no HIP device */
void ssdk_ctx_destroy(ssdk_ctx* ctx);  /* the caller has synchronised the streams that used it */

/* decoder.py:25-49 Decoder.__call__: decode every level -> concat -> nms, nothing returns to the host in between.
 * Three launches: scan16_kernel (one HBM pass over the conf tensors: threshold + exact top-k per scan unit, unsorted;
 * scan_kernel for fp32 heads, thresholds <= 0 and top_n > 512), levelsel_kernel (per (image, level): select the level's
 * top_n of its units' keys, order them, delta2box + centre rescoring) and nmswalk_kernel (per image: order by rescored
 * score in rounds, greedy class-aware (D)IoU NMS).  Geometries whose per-level lists do not fit the LDS run round 1's
 * level_kernel + nms_kernel behind the scan instead (same results; SSDK_DECODE_FUSED=0 forces it).
 * mid_* (optional, may be NULL): the concatenated per-level decode output [B, L*top_n(,4)] that the reference
 * materialises with torch.cat (decoder.py:48); when NULL the library keeps them in the workspace. */
size_t ssdk_decode_nms_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype,
                                       int top_n_per_level, int ndetections);
int ssdk_decode_nms(const ssdk_level* levels, int L, int B, int dtype, float threshold,
                    int top_n_per_level, int rescore, float nms_threshold, int ndetections,
                    int using_diou, float* out_scores, float* out_boxes, float* out_classes,
                    float* mid_scores, float* mid_boxes, float* mid_classes, void* workspace,
                    size_t workspace_bytes, void* stream);
int ssdk_decode_nms_ctx(ssdk_ctx* ctx, const ssdk_level* levels, int L, int B, int dtype, float threshold,
                        int top_n_per_level, int rescore, float nms_threshold, int ndetections,
                        int using_diou, float* out_scores, float* out_boxes, float* out_classes,
                        float* mid_scores, float* mid_boxes, float* mid_classes, void* workspace,
                        size_t workspace_bytes, void* stream);

/* (new) Tail stream of ssdk_decode_nms[_ctx] (NULL = off, the default): when set, the scan still runs on the
 * `stream` argument but everything behind it is enqueued on the tail stream, ordered after the scan by an event -- the
 * latency-bound end of the stage then overlaps the next batch's forward pass.  The caller owns the consequences: the
 * outputs are complete on the TAIL stream, the workspace and the box tensors must stay untouched until the tail work
 * has finished (alternate two workspaces and make the main stream wait for the tail event of the call that last used
 * one; host mirror: ssds/modeling/layers/decoder.py). */
int ssdk_ctx_set_tail_stream(ssdk_ctx* ctx, void* stream);
int ssdk_set_decode_tail_stream(void* stream); /* the calling thread's default context */

/* (new, debug) SSDK_LDS_POISON=1 in the environment fills every CU's LDS with NaN patterns in front of each kernel of
 * ssdk_run_ops / ssdk_conv / ssdk_mbconv / ssdk_decode_nms, so that a kernel reading LDS it did not write shows up
 * as NaNs in the parity tests instead of depending on the previous tenant of its CU.  ssdk_debug_lds_probe poisons
 * once and returns in *count (device, zeroed by the caller) how many poisoned words a kernel that never wrote its LDS
 * can see. */
int ssdk_debug_lds_probe(unsigned* count, void* stream);

/* Optional per-kernel timing for roofline accounting (bench.py): when enabled, ssdk_decode_nms[_ctx] records
 * hipEvents around its launches into the context's ring of 256 slots (no synchronisation inside the timed region).
 * enable = 1: one event interval per launch -- ssdk_[ctx_]get_timings(back, ms, 3): ms[0] = scan, ms[1] = levelsel (or
 * level_kernel on the round-1 path), ms[2] = nmswalk (nms_kernel) of the call `back` calls before the most recent one;
 * enable = 2: ONE interval around the stage's launches (every extra event costs ~4.6 us of GPU time and separates the
 * kernels it sits between) -- ms[0] = the whole stage, ms[1] = ms[2] = 0.  It synchronises on that call's last event. */
int ssdk_ctx_set_profiling(ssdk_ctx* ctx, int enable);
int ssdk_ctx_get_timings(ssdk_ctx* ctx, int back, float* ms, int n);
int ssdk_set_profiling(int enable);
int ssdk_get_timings(int back, float* ms, int n);
/* (debug) SSDK_TAIL_STAMPS=1: clock stamps of the most recent decode stage -- workgroup 0's phase boundaries in
 * levelsel_kernel / nmswalk_kernel / scan16_kernel (out[0..23]; layout: tools/scan_probe.py) and, from out[48] on, the
 * wall-clock start / end of every scan workgroup (2 x min(workgroups, 4096) words); n >= 48; synchronises the device. */
int ssdk_ctx_get_tail_stamps(ssdk_ctx* ctx, unsigned long long* out, int n);
/* Per-op timing of ssdk_run_ops[_ctx]: when enabled, an event is recorded before every op (and after the last) on the
 * caller's stream.  After synchronising, ssdk_[ctx_]get_op_timings fills ms[i] / kernels[i] (kernel name, may be
 * NULL) for the ops of the most recent profiled call and returns their count. */
int ssdk_ctx_set_op_profiling(ssdk_ctx* ctx, int enable);
int ssdk_ctx_get_op_timings(ssdk_ctx* ctx, float* ms, const char** kernels, int n_max);
int ssdk_set_op_profiling(int enable);
int ssdk_get_op_timings(float* ms, const char** kernels, int n_max);
/* Side lane of ssdk_run_ops_ctx: 1 on, 0 off (everything in line on the caller's stream), -1 = the environment's
 * choice (SSDK_SIDE_STREAM, default on). */
int ssdk_ctx_set_side_lane(ssdk_ctx* ctx, int enable);

/* box.py:362-405 extract_targets + box.py:116-226 snap_to_anchors_by_iou for ONE level and the whole
 * batch in one launch.  targets[B*G*5] device fp32 (x, y, w, h, label), rows with label <= -1 are
 * padding (box.py:375).  anchors[A*4] HOST fp32.  Outputs device fp32, fully written:
 * cls_target[B,A,C,H,W], box_target[B,A,4,H,W], depth[B,A,1,H,W]. */
int ssdk_match_targets(const float* targets, int B, int G, const float* anchors, int A, int C, int H,
                       int W, int stride, float match_threshold, float unmatch_threshold,
                       float center_sampling_radius, float* cls_target, float* box_target,
                       float* depth, void* stream);

/* box.py:229-359 snap_to_anchors_by_scale (FCOS-style scale-range assignment, reached from
 * extract_targets box.py:388-400 when `match[0]` is a list) for ONE level and the whole batch.  A box is a
 * candidate for grid point (x, y) of anchor a when the point lies strictly inside it and
 * lower_scale*sqrt(area_a) (clamped at -1) <= max(l,t,r,b) <= upper_scale*sqrt(area_a); with
 * center_sampling != 0 the point must lie in the box's centre region (radius 1.5 strides, the reference's
 * fixed default) and the box's sqrt-area is compared instead.  The smallest candidate wins; depth is
 * label+1 or 0 (no ignore band).  Same buffers and layout as ssdk_match_targets. */
int ssdk_match_targets_by_scale(const float* targets, int B, int G, const float* anchors, int A, int C,
                                int H, int W, int stride, float lower_scale, float upper_scale,
                                int center_sampling, float* cls_target, float* box_target, float* depth,
                                void* stream);

/* Target assignment fused with the training losses of ONE level: replaces, for the whole batch,
 * extract_targets (box.py:362-405) + FocalLoss (core/criterion.py:74-108) + SmoothL1Loss (criterion.py:111-151) +
 * the depth masks and sums of ModelWithLossBasic.forward (pipeline/pipeline_anchor_apex.py:48-66).  The three
 * target tensors are never materialised: a thread matches its anchor exactly as ssdk_match_targets[_by_scale]
 * does, reads the C class logits and 4 box regressions of that anchor, and writes
 *   d_conf = d(sum of masked focal terms)/d conf,  d_loc = d(sum of masked smooth-L1 terms)/d loc
 * (same shape/dtype as conf [B, A*C, H, W] / loc [B, A*4, H, W], NCHW contiguous; fp32 arithmetic), and
 *   sums[0] = sum_{depth>=0} focal,  sums[1] = sum_{depth>0} smooth-L1,  sums[2] = #(depth>0)
 * reduced in a fixed order (bit-reproducible).  by_scale = 0: thr_a/thr_b = match/unmatch IoU thresholds and
 * `radius` the centre-sampling radius; by_scale = 1: thr_a/thr_b = lower/upper scale multipliers and radius != 0
 * turns centre sampling on.  The caller divides by the foreground count summed over levels (:69-71) and scales
 * the gradients by the incoming scalar.  dtype = SSDK_F32 | SSDK_BF16 | SSDK_F16 of conf/loc/d_conf/d_loc.
 * loc_loss: 0 = SmoothL1Loss(beta); 1..4 = IOULoss 'iou' | 'giou' | 'diou' | 'ciou' (criterion.py:154-293): one term per
 * foreground anchor on the boxes the predicted / target deltas encode, differentiated in forward mode inside the
 * kernel with torch's rules for max/min ties and clamp. */
size_t ssdk_match_loss_workspace_bytes(int B, int A, int H, int W);
int ssdk_match_loss(const float* targets, int B, int G, const float* anchors, int A, int C, int H, int W,
                    int stride, int by_scale, float thr_a, float thr_b, float radius, const void* conf,
                    const void* loc, int dtype, float alpha, float gamma, float beta, int loc_loss,
                    void* d_conf, void* d_loc, float* sums, void* workspace, size_t workspace_bytes, void* stream);

/* The same per-level body with MultiBoxLoss as the class criterion (core/criterion.py:43-71; cfg MATCHER.NEGPOS_RATIO):
 * sigmoid cross entropy on the positives plus the hardest negpos_ratio x #positives negatives (depth == 0) of each image
 * of THIS level, hardness = an anchor's largest per-class term (:59-61), count clamped to A*H*W - 1 (:67).  Three launches
 * instead of the reference's two full sorts per level: match + positives' terms + hardness keys, a per-image radix select
 * of the num_neg-th largest key (ties between EQUAL keys are kept in index order; torch's unstable sort leaves that
 * choice unspecified), the mined negatives' terms and gradients; then the fixed-order reduction.  sums / d_conf / d_loc /
 * loc_loss / by_scale / thr_* / radius exactly as ssdk_match_loss; sums[0] = sum over positives and mined negatives.
 * A mined negative's term is BCE against the anchor's OWN one-hot target: all zeros for a background anchor, one-hot for an
 * anchor that matched a box but lies outside the centre-sampling region (depth 0 with a class target, box.py:183-207). */
size_t ssdk_match_multibox_loss_workspace_bytes(int B, int A, int H, int W);
int ssdk_match_multibox_loss(const float* targets, int B, int G, const float* anchors, int A, int C, int H, int W,
                             int stride, int by_scale, float thr_a, float thr_b, float radius, const void* conf,
                             const void* loc, int dtype, float negpos_ratio, float beta, int loc_loss, void* d_conf,
                             void* d_loc, float* sums, void* workspace, size_t workspace_bytes, void* stream);

/* VOC-style mAP bookkeeping of the eval epoch: MeanAveragePrecision.__call__
 * (core/evaluation_metrics.py:15-61, called per batch at pipeline/pipeline_anchor_basic.py:161-176) for one batch
 * of decoder outputs.  scores [B, D], boxes [B, D, 4] (ltrb), classes [B, D] fp32 as returned by ssdk_decode_nms;
 * targets [B, G, 5] fp32 (ltrb + label, label < 0 = padding; the caller has already converted xywh -> ltrb like
 * pipeline_anchor_basic.py:175).  Per image and class, a detection with score > conf_threshold takes the
 * same-class box of highest IoU (first maximum; IoU without the +1 of box.py, evaluation_metrics.py:16-26) and is
 * a true positive iff that IoU >= iou_threshold and no earlier detection claimed the box (:51-56).
 * Writes one record per detection slot: keys[b, i] = (class << 32) | ~ordered(score) (ascending key = class
 * ascending, score descending; slots that are padding / below threshold / out-of-range class get class =
 * num_classes) and tp[b, i]; adds the number of ground-truth boxes of each class to npos[num_classes] (int32,
 * zeroed by the caller before the first batch).  D <= 2048, G <= SSDK_MAX_GT. */
int ssdk_map_match(const float* scores, const float* boxes, const float* classes, int B, int D,
                   const float* targets, int G, int num_classes, float conf_threshold, float iou_threshold,
                   long long* keys, unsigned char* tp, int* npos, void* stream);

/* MeanAveragePrecision.get_results (evaluation_metrics.py:63-142) on the records of a whole epoch sorted by key
 * (stable): tp_sorted [N], seg_offsets[c] .. seg_offsets[c+1] = the records of class c (int64 [num_classes+1]).
 * ap[c] (fp64) = area under the precision envelope in recall (VOC definition :100-112); NaN when npos[c] == 0
 * (:124-128), 0 when the class has ground truth but no detections (:97-98). */
int ssdk_map_average_precision(const unsigned char* tp_sorted, const long long* seg_offsets, const int* npos,
                               int num_classes, double* ap, void* stream);

/* Fused convolution + folded BatchNorm + activation (+ residual) for the detector network:
 * basic_layers.py:5-57 (SepConvBNReLU / ConvBNReLU / ConvBNReLUx2), the MobileNetV2 blocks behind
 * nets/mobilenet.py:56-99, and the bare multibox head convs ssd.py:100-103 / fpn.py:10-18.
 *   y = act(conv(x, w) * scale[c] + bias[c]) (+ residual)
 * Dispatch:  groups == 1, Cin % 8 == 0  -> MFMA implicit GEMM (1x1 / 3x3, stride 1|2, pad k/2)
 *            groups == Cin == Cout      -> depthwise 3x3 (HBM-bound, NHWC)
 *            Cin <= 4 (image stem)      -> direct 3x3; w is fp32 with the BN scale folded in, scale = NULL
 * Layouts:   x NHWC (= torch channels_last) except the stem which also takes NCHW;
 *            w KRSC [Cout][kh][kw][Cin/groups] in the activation dtype;  scale (may be NULL = 1), bias fp32;
 *            y NHWC, or NCHW for the multibox heads (the [B, A*C, H, W] layout decode consumes); with NCHW
 *            output channels >= split go to y2 with activation act2 (loc | conf of one SSD level as ONE GEMM).
 */
enum { SSDK_LAYOUT_NCHW = 0, SSDK_LAYOUT_NHWC = 1 };
typedef struct ssdk_conv_desc {
  const void* x;
  const void* w;
  const float* scale;
  const float* bias;
  const void* residual; /* optional, NHWC: y = act(conv) + residual, or act(conv + residual) with res_mode bit 1 */
  void* y;
  void* y2;             /* optional second output (NCHW split) */
  int32_t N, Cin, H, W, Cout, k, stride, groups; /* groups: 1 dense | Cin depthwise (k=3) | Cin/16 grouped (k=3) */
  int32_t act, act2, split;
  int32_t dtype;        /* SSDK_BF16 | SSDK_F16 (input, weights and output) */
  int32_t in_layout, out_layout;
  int32_t res_mode;     /* bit 0: `residual` is half resolution [N][H/2][W/2][Cout], added nearest-x2-upsampled (FPN
                           top-down path, fpn.py:80-87); bit 1: activation applied AFTER the add (ResNet blocks) */
  const void* w_frag;   /* optional (NULL: not given): the same weights in FRAGMENT-MAJOR order, see ssdk_weight_frag_bytes.
                           Kernels that stream weights straight into MFMA operand registers (small maps: the weights ARE the
                           traffic) read it instead of `w`: 1 KiB contiguous per wave instruction instead of 16 rows x 64 B,
                           measured 40-50 instead of 14 B/clk per CU (tools/micro/wstream.hip) */
} ssdk_conv_desc;
/* Fragment-major image of a row-major weight matrix w[rows][K] (a KRSC conv weight: rows = Cout, K = k*k*Cin), K % 32 == 0:
 *   frag[g][ks][fg][fr][j] = w[16*g + fr][32*ks + 8*fg + j]     g < ceil(rows/16), ks < K/32, fg < 4, fr < 16, j < 8
 * rows past `rows` are zero.  One (g, ks) block = 1 KiB = the A operand of one v_mfma_f32_16x16x32 k-step of 16 output
 * channels, lane (16*fg + fr) owning 16 contiguous bytes.  Pure layout: the caller builds it once per model (the Python
 * host does it in ConvPack.frag()). */
size_t ssdk_weight_frag_bytes(int rows, int K);
size_t ssdk_conv_workspace_bytes(int N, int Cin, int H, int W, int Cout, int k, int stride, int dtype);
int ssdk_conv(const ssdk_conv_desc* desc, void* workspace, size_t workspace_bytes, void* stream);
/* a whole pre-planned network: descs[0..n) launched in order on `stream` (one host call per forward) */
int ssdk_conv_sequence(const ssdk_conv_desc* descs, int n, void* workspace, size_t workspace_bytes,
                       void* stream);

/* One MobileNetV2 inverted-residual block (the torchvision InvertedResidual behind nets/mobilenet.py:56,
 * 84-89) as ONE kernel: 1x1 expand + BN + ReLU6 -> 3x3 depthwise (stride 1|2) + BN + ReLU6 -> 1x1 linear
 * projection + BN (+ x), the expanded tensor never leaves the chip.  NHWC in/out, bf16 | f16.
 * The expanded and depthwise tensors are internal (LDS only) and held in fp16 whatever the model dtype:
 *   w_expand [Chid][Cin] activation dtype, scale_expand / bias_expand fp32 [Chid] (folded BN);
 *   w_dw fp16 [3][3][Chid] with the BN scale folded in, bias_dw fp16 [Chid];
 *   w_project fp16 [Cout][Chid], scale_project / bias_project fp32 [Cout].
 * Limits: Cin <= 160, Cout <= 320, channels % 8 == 0; residual needs stride 1, Cin == Cout. */
typedef struct ssdk_mbconv_desc {
  const void* x;
  void* y;
  const void* w_expand;
  const float* scale_expand;
  const float* bias_expand;
  const void* w_dw;
  const void* bias_dw;
  const void* w_project;
  const float* scale_project;
  const float* bias_project;
  int32_t N, H, W, Cin, Chid, Cout, stride, residual, dtype;
  int32_t stem; /* 0: x is the NHWC block input.  1 | 2: x is the NCHW | NHWC IMAGE [N,Cin<=3,H,W] and the
                   "expand" conv is the network stem (3x3, stride 2, pad 1, BN, ReLU6; w_expand
                   [Chid][3][8][4] = (ky, kx zero-padded to 8, ci zero-padded to 4)): stem + first depthwise-separable block
                   (mobilenet.py:78-89, expand_ratio 1) in one launch */
  int32_t variant; /* Two kernels implement ssdk_mbconv: the LDS-tiled one (every geometry) and, for the high-resolution
                      blocks (Cin <= 32, hidden in {96, 144, 192}, Cout <= 64, or the stem block), a register-flow one
                      (ssdk_mbflow.hip).  0: automatic (register-flow where the map is large enough to pay), 1: register-flow
                      wherever it exists, -1: LDS-tiled only (tests, A/B runs).  Part of the descriptor: the library keeps
                      no process-wide switch.  ssdk_last_kernel() names the kernel that ran.
                      2: ssdk_mbsplit.hip wherever it exists, 3: ssdk_mbk.hip wherever it exists (tests). */
  /* appended in 0.2.2 (zero = absent).  OPTIONAL image of the block's 1x1 weights and per-channel constants for ssdk_mbk.hip
   * (the blocks of MobileNetV2 on 16- and 32-pixel-wide maps, mobilenet.py:56, 84-89: 64 -> 384 -> 64 | 96 and 96 -> 576 -> 96
   * @32x32, 96 -> 576 -> 160 / stride 2 @32x32, 160 -> 960 -> 160 | 320 @16x16), built once per model by the host
   * (ssds/modeling/layers/fused_conv.py MbPack.image); without it those blocks run on the LDS-tiled kernel.
   *   image_nw  slices of the hidden channels = waves per work item the image is built for (4);
   *             NCHW = ceil(Chid / 16 / image_nw) chunks of 16 per slice, NP = ceil(NCHW / 2) chunk pairs, KS = Cin / 32,
   *             NFO = column fragments per half (ssdk_mbk_image_bytes reports it), halves = Cout / (16 NFO)
   *   layout    1. weights [halves][image_nw slices][NP pairs][2 KS + NFO fragments of 1 KiB]; a fragment = 64 lanes x 8
   *                16-bit elements, lane l:
   *                  expand (chunk cc of the pair, k-step ks): w_expand[16 c + (l & 15)][32 ks + 8 (l >> 4) + 0..7], c = slice *
   *                    NCHW + 2 pair + cc, in the activation dtype (BN scale folded in, like w_expand);
   *                  project (column fragment f of half h): element j = w_project[16 NFO h + 16 f + (l & 15)][16 (slice * NCHW
   *                    + 2 pair + j / 4) + 4 (l >> 4) + j % 4], fp16;
   *                  zeros where the chunk lies beyond the slice (an odd NCHW) or beyond Chid;
   *             2. constants [image_nw slices][ceil(NCHW * 384 / 1024) KiB]: per slice, for its chunks c (channel = 16 (slice *
   *                NCHW + c) + 4 g + q): bias_expand [NCHW][4 g][4 q] fp32 | w_dw [NCHW][9 taps][4 g][4 q] fp16 |
   *                bias_dw / 6 [NCHW][4 g][4 q] fp16 (fp32 multiply by 1/6, rounded to fp16), zero-padded to the KiB;
   *             3. projection BN [halves][2 KiB]: [NFO f][4 g][scale_project * 6 (4 q fp32) | bias_project (4 q fp32)], channel
   *                16 NFO h + 16 f + 4 g + q, zero-padded.
   *   w_image_bytes >= ssdk_mbk_image_bytes(Cin, Chid, Cout, stride, output width, image_nw, &NFO) (0: no instance of the
   *   kernel takes the block at that width). */
  int32_t image_nw;
  const void* w_image;
  size_t w_image_bytes;
} ssdk_mbconv_desc;
int ssdk_mbconv(const ssdk_mbconv_desc* desc, void* stream);
size_t ssdk_mbk_image_bytes(int Cin, int Chid, int Cout, int stride, int out_width, int image_nw, int* nfo);

/* Weighted feature fusion of the BiFPN (bifpn.py:41-62), NHWC, one launch:
 *   y = w0 * a + w1 * R_b(b) [+ w2 * R_c(c)]      a, y: [N][H][W][C]
 * R = SAME (source [N][H][W][C]), UP2 (nearest x2 upsample of [N][H/2][W/2][C], F.interpolate scale_factor=2) or
 * POOL2 (F.max_pool2d(kernel_size=2) of [N][hb][wb][C], floor(hb/2) == H).  fp32 accumulate, one rounding.  c may be NULL. */
enum { SSDK_FUSE_SAME = 0, SSDK_FUSE_UP2 = 1, SSDK_FUSE_POOL2 = 2 };
typedef struct ssdk_fuse_desc {
  const void* a;
  const void* b;
  const void* c;
  void* y;
  float w0, w1, w2;
  int32_t mode_b, mode_c;
  int32_t N, H, W, C;
  int32_t hb, wb, hc, wc; /* source dims of b / c; only read for POOL2 (floor mode: 2H or 2H+1) */
  int32_t dtype;
} ssdk_fuse_desc;
int ssdk_fuse(const ssdk_fuse_desc* desc, void* stream);

/* Detector front door (SSDDetector.__call__, ssds.py:47-57): transpose + `(x - mean) / std` + cast in one pass.
 *   x  raw image batch on the device: [N,H,W,C] (src_layout NHWC = 1) or [N,C,H,W] (NCHW = 0), C <= 4,
 *      src_dtype SSDK_U8 | SSDK_F32 | SSDK_BF16 | SSDK_F16
 *   mean, std  HOST fp32 [C];   y  [N,C,H,W] of dst_dtype (SSDK_F32 | SSDK_BF16 | SSDK_F16)
 * fp32 arithmetic in the reference's order (subtract, divide), one rounding to dst_dtype. */
int ssdk_preprocess(const void* x, int src_dtype, int src_layout, int N, int H, int W, int C, const float* mean,
                    const float* std, void* y, int dst_dtype, void* stream);

/* Dense 1x1 / stride-1 convolutions of the TRAINING step on NCHW tensors (version 240; csrc/ssdk_pwtrain.hip): the pointwise
 * convolutions of MobileNetV2's inverted-residual blocks (nets/mobilenet.py:56, 78 through torchvision's InvertedResidual /
 * ConvBNReLU) and of the SSD extras (layers/basic_layers.py:40-57), forward and backward in the reference's DDP step
 * (pipeline/pipeline_anchor_apex.py:103-130).  16-bit tensors (SSDK_BF16 | SSDK_F16), fp32 accumulation, HW = H * W, tensors
 * contiguous, pointers need the alignment of their element type only (the weight matrix: 16 bytes).
 *   ssdk_pw_prepare   w32 [Cout, Cin] fp32 (the master weights)  ->  w16 [Cout, Cin] and wt16 [Cin, Cout] in `dtype`
 *   ssdk_pw_forward   y[b] = a x[b] (+ bias):  x [B, K, HW], a [M, K] row-major 16 bit, bias fp32 [M] or NULL  ->  y [B, M, HW]
 *                     forward:         a = w16  (K = Cin,  M = Cout)
 *                     input gradient:  a = wt16 (K = Cout, M = Cin), x = dy, bias = NULL
 *                     K must be a multiple of 8.
 *   ssdk_pw_wgrad     dw [Cout, Cin] fp32 = sum_b dy[b] x[b]^T  (dy [B, Cout, HW], x [B, Cin, HW]); per-wave fp32 partial tiles
 *                     through `workspace` (ssdk_pw_wgrad_workspace_bytes), added in index order: bit-reproducible.
 *   ssdk_pw_forward_stats   ssdk_pw_forward + sums [M][2] = per output channel (sum y, sum y^2) over B * HW of its outputs (fp32
 *                     accumulators, before the store's rounding): the statistics of the BatchNorm that follows the convolution (ssdk_bn_act_train_fwd_sums), without a
 *                     pass over y.  Per-wave partials through `workspace` (ssdk_pw_stats_workspace_bytes, 16-byte aligned), added
 *                     in index order: bit-reproducible. */
int ssdk_pw_prepare(const float* w32, void* w16, void* wt16, int Cout, int Cin, int dtype, void* stream);
size_t ssdk_pw_stats_workspace_bytes(int B, int K, int M, int HW);
int ssdk_pw_forward_stats(const void* x, const void* a, const float* bias, void* y, float* sums, void* workspace,
                          size_t workspace_bytes, int B, int K, int M, int HW, int dtype, void* stream);
int ssdk_pw_forward(const void* x, const void* a, const float* bias, void* y, int B, int K, int M, int HW, int dtype,
                    void* stream);
size_t ssdk_pw_wgrad_workspace_bytes(int B, int Cout, int Cin, int HW);
int ssdk_pw_wgrad(const void* dy, const void* x, float* dw, void* workspace, size_t workspace_bytes, int B, int Cout, int Cin,
                  int HW, int dtype, void* stream);

/* Dense 3x3 / pad 1 / stride 1 | 2 convolutions of the TRAINING step (stem mobilenet.py:78, extras basic_layers.py:40-57, heads
 * ssd.py:100-103; version 240): y[b] = W2 col[b] with W2 = weight.view(Cout, Cin * 9) (k = ci * 9 + ky * 3 + kx) padded to
 * Kp = Cin * 9 rounded up to 8 columns, so forward / input gradient / weight gradient are ssdk_pw_forward(col, W2),
 * ssdk_pw_forward(dy, W2^T) + ssdk_col2im3x3, ssdk_pw_wgrad(dy, col).  16-bit NCHW tensors, Ho = (H - 1) / stride + 1.
 *   ssdk_im2col3x3   x [B, C, H, W] -> col [B, Kp, Ho * Wo]   (zero outside the plane and in the rows C * 9 .. Kp)
 *   ssdk_col2im3x3   dcol [B, Kp, Ho * Wo] -> dx [B, C, H, W]  (per input pixel a gather of <= 9 terms, fp32 sum in tap order) */
int ssdk_im2col3x3(const void* x, void* col, int B, int C, int H, int W, int stride, int dtype, void* stream);
int ssdk_col2im3x3(const void* dcol, void* dx, int B, int C, int H, int W, int stride, int dtype, void* stream);
/* The same two with the batch folded into the pixel dimension: col / dcol are [Kp, B, Ho * Wo], i.e. ONE image of B * Ho * Wo pixels
 * for ssdk_pw_forward / ssdk_pw_wgrad (called with B = 1, HW = B * Ho * Wo), which tile the pixels of one image in groups of 128:
 * the layers with few pixels per image (the extras of /root/reference/ssds/modeling/ssds/ssd.py:63-65: 64 / 16 / 4 / 1). */
int ssdk_im2col3x3_folded(const void* x, void* col, int B, int C, int H, int W, int stride, int dtype, void* stream);
int ssdk_col2im3x3_folded(const void* dcol, void* dx, int B, int C, int H, int W, int stride, int dtype, void* stream);

/* The network's FIRST convolution inside the training step (3x3 / stride 2 / pad 1, Cin <= 3 image channels, Cout <= 32, bias-free:
 * torchvision MobileNetV2 features[0][0] behind /root/reference/ssds/modeling/nets/mobilenet.py:180-192; the reference trains it
 * through cuDNN, pipeline_anchor_apex.py:37-72) on its own kernels (csrc/ssdk_stemtrain.hip) -- an image has no input gradient:
 *   ssdk_stem3x3s2_fwd     x [N, Cin, H, W] 16 bit, w [Cout, Cin, 3, 3] fp32 master weights (rounded to the tensor dtype inside,
 *                          like autocast's cast) -> y [N, Cout, Ho, Wo] 16 bit, fp32 accumulation in tap order; Ho = (H - 1) / 2 + 1
 *   ssdk_stem3x3s2_wgrad   x, dy [N, Cout, Ho, Wo] 16 bit -> dw [Cout, Cin, 3, 3] fp32 (matrix cores over pixels, workgroup partials
 *                          added in index order: bit-reproducible)
 * workspace of the weight gradient: ssdk_stem3x3s2_wgrad_workspace_bytes(N, H) bytes, 16-byte aligned. */
size_t ssdk_stem3x3s2_wgrad_workspace_bytes(int N, int H);
int ssdk_stem3x3s2_fwd(const void* x, const float* w, void* y, int N, int Cin, int H, int W, int Cout, int dtype, void* stream);
int ssdk_stem3x3s2_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Cin, int H, int W,
                         int Cout, int dtype, void* stream);

/* Weights of a 3x3 layer -- or of the loc | conf PAIR of an SSD level (reference ssd.py:100-103), w2 / b2 / n2 = NULL / NULL / 0
 * for a single layer -- from the fp32 master tensors [n, Cin, 3, 3] into the layouts ssdk_conv reads, in one launch (the head
 * convolutions of the TRAINING step run on the inference kernels: weights change every step): krsc = 16-bit [n1 + n2][3][3][Cin];
 * frag (may be NULL) = the fragment-major image of ssdk_weight_frag_bytes(n1 + n2, 9 * Cin), rows past the last channel zero;
 * bias = fp32 [n1 + n2] (a NULL b1 / b2 contributes zeros).  Cin % 8 == 0; frag needs (9 * Cin) % 32 == 0. */
int ssdk_pack_conv3x3(const float* w1, const float* b1, int n1, const float* w2, const float* b2, int n2, int cin, void* krsc, void* frag,
                      float* bias, int dtype, void* stream);
/* The same layer (pair) as the weights of its INPUT-GRADIENT convolution (stride 1, pad 1): dx = conv3x3(dy, W'),
 * W'[ci][ky][kx][o] = W[o][ci][2 - ky][2 - kx], o zero-padded to opad >= n1 + n2 channels (opad % 8 == 0; dy is handed to ssdk_conv
 * with opad channels).  krsc = 16-bit [Cin][3][3][opad]; frag (may be NULL; opad % 32 == 0) = its fragment-major image. */
int ssdk_pack_conv3x3_dgrad(const float* w1, int n1, const float* w2, int n2, int cin, int opad, void* krsc, void* frag, int dtype,
                            void* stream);
/* a [N, c1, HW] | b [N, c2, HW] (16-bit NCHW planes; b / c2 may be NULL / 0) -> out [N, HW, cpad] (NHWC), channels past c1 + c2
 * zero; cpad even.  The concatenated output gradient of a pair in the layout ssdk_conv reads. */
int ssdk_concat_nchw_to_nhwc(const void* a, int c1, const void* b, int c2, void* out, int cpad, int N, int HW, int dtype, void* stream);

/* SGD with momentum / weight decay / Nesterov over ALL parameter tensors of a model (version 240; csrc/ssdk_sgd.hip): the
 * optimizer.step() of the reference's loop (pipeline_anchor_apex.py:128-130 on core/optimizer.py:73-134's torch.optim.SGD) with
 * the reference's NaN/Inf skip (:110-111, 126-127) taken on the device.  fp32 tensors.  params / grads / momentum_bufs / numel
 * are HOST arrays of n entries (device pointers, element counts); momentum_bufs may be NULL when momentum == 0.
 *   g = grad + weight_decay p;  buf = momentum buf + g;  p -= lr (nesterov ? g + momentum buf : buf)
 * lr_dev (device float, may be NULL -> lr) is read by the kernel: a captured hipGraph sees later changes.  found_inf (device
 * float, may be NULL): non-zero = no tensor is touched. */
int ssdk_sgd_step(int n, void* const* params, const void* const* grads, void* const* momentum_bufs, const int64_t* numel,
                  const float* lr_dev, float lr, float momentum, float weight_decay, int nesterov, const float* found_inf,
                  void* stream);

/* Depthwise 3x3 convolution (pad 1, stride 1|2) for the TRAINING step: forward, input gradient and weight gradient,
 * NCHW contiguous, dtype SSDK_F32 | SSDK_BF16 | SSDK_F16, fp32 accumulation (replaces MIOpen's naive_conv_* kernels
 * behind torch.nn.functional.conv2d(groups = C) in the DDP step, pipeline_anchor_apex.py:75-171).
 *   fwd:        x [N,C,H,W], w [C,1,3,3] -> y [N,C,Ho,Wo]
 *   bwd_data:   dy [N,C,Ho,Wo], w -> dx [N,C,H,W]           (H, W are the INPUT dims in all three calls)
 *   bwd_weight: x, dy -> dw fp32 [C,1,3,3]; two-stage fixed-order reduction (bit-reproducible) through `workspace`
 * Any plane size; pointers need only the alignment of their element type (rows are read and written as unaligned
 * 16-byte runs). */
int ssdk_dwconv_fwd(const void* x, const void* w, void* y, int N, int C, int H, int W, int stride, int dtype, void* stream);
/* (version 240) forward + sums [C][2] = per channel (sum y, sum y^2) over N * Ho * Wo of its outputs (fp32 accumulators, before the
 * store's rounding): the batch statistics of the BatchNorm behind the convolution (ssdk_bn_act_train_fwd_sums), without a pass over
 * y.  Per-workgroup partials through `workspace` (16-byte aligned), added in index order.  ssdk_dwconv_fwd_stats_workspace_bytes
 * returns 0 where the geometry runs on the tiled fallback kernels (rows too wide): call ssdk_dwconv_fwd there. */
size_t ssdk_dwconv_fwd_stats_workspace_bytes(int N, int C, int H, int W, int stride, int dtype);
/* (version 240) the depthwise convolution behind a DEFERRED BatchNorm: x is the BatchNorm's INPUT, coef [C][4] / act come from
 * ssdk_bn_act_train_stats; the kernels stage act(a x + b) rounded to the dtype -- bit for bit what the BatchNorm's apply pass would
 * have stored.  16 bit, whole-row kernels only: ssdk_dwconv_affine_supported tells (1 | 0).  sums (optional, with `workspace` of
 * ssdk_dwconv_fwd_stats_workspace_bytes): the statistics of y for the NEXT BatchNorm. */
int ssdk_dwconv_affine_supported(int N, int C, int H, int W, int stride, int dtype);
int ssdk_dwconv_fwd_affine(const void* x, const float* coef, int act, const void* w, void* y, float* sums, void* workspace,
                           size_t workspace_bytes, int N, int C, int H, int W, int stride, int dtype, void* stream);
int ssdk_dwconv_bwd_weight_affine(const void* x, const float* coef, int act, const void* dy, float* dw, void* workspace,
                                  size_t workspace_bytes, int N, int C, int H, int W, int stride, int dtype, void* stream);
int ssdk_dwconv_fwd_stats(const void* x, const void* w, void* y, float* sums, void* workspace, size_t workspace_bytes, int N, int C,
                          int H, int W, int stride, int dtype, void* stream);
int ssdk_dwconv_bwd_data(const void* dy, const void* w, void* dx, int N, int C, int H, int W, int stride, int dtype,
                         void* stream);
size_t ssdk_dwconv_bwd_weight_workspace_bytes(int N, int C, int H, int W, int stride);
int ssdk_dwconv_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int C,
                           int H, int W, int stride, int dtype, void* stream);

/* How the whole-row depthwise kernels cut one pass into workgroups (host only: no launch, usable without a device; the
 * CPU tests check the plan's invariants with it).  pass 0 forward | 1 input gradient | 2 weight gradient.
 * out[12] = images per workgroup, bands per plane, rows per band, 8-pixel segments per row, LDS row stride, staged rows
 * per piece, 16-byte chunks per staged row, units per piece, workgroups, partial-sum groups of the weight gradient, LDS
 * bytes, rows of the thread space.  Returns 0, or 1 when the tiled kernels take the pass (rows too wide for LDS). */
int ssdk_dwconv_plan(int pass, int N, int C, int H, int W, int stride, int dtype, int* out);


/* BatchNorm2d with batch statistics for the TRAINING step (replaces MIOpenBatchNorm{Fwd,Bwd}Spatial), NCHW
 * contiguous data of dtype SSDK_F32 | SSDK_BF16 | SSDK_F16, fp32 parameters / statistics, HW = H*W.
 *   fwd: y = (x - mean_c) * invstd_c * weight_c + bias_c;  save_mean / save_invstd fp32 [C] (for backward);
 *        running_mean / running_var (may both be NULL) updated like torch: momentum, unbiased variance
 *   bwd: dx, dweight = sum(dy * xhat), dbias = sum(dy)   (dweight / dbias may be NULL)
 * Reductions run in a fixed order (bit-reproducible).  workspace: ssdk_bn_workspace_bytes(N, C), 16-byte aligned. */
size_t ssdk_bn_workspace_bytes(int N, int C);
int ssdk_bn_train_fwd(const void* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                      void* y, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, int N, int C,
                      int HW, float momentum, float eps, int dtype, void* stream);
int ssdk_bn_train_bwd(const void* x, const void* dy, const float* weight, const float* save_mean, const float* save_invstd,
                      void* dx, float* dweight, float* dbias, void* workspace, size_t workspace_bytes, int N, int C,
                      int HW, int dtype, void* stream);

/* The same with the activation that follows the BatchNorm in every Conv-BN-ReLU6 block of the backbone
 * (nets/mobilenet.py:24-33; basic_layers.py:5-57) folded in: act = 0 none | 1 ReLU6 | 2 ReLU.
 *   fwd: y = act(bn(x));  bwd: dy is the gradient w.r.t. act(bn(x)) -- it is masked where the pre-activation value (as
 *        the forward pass rounded it to `dtype`) lies outside the open pass-through interval, exactly what
 *        hardtanh_backward / threshold_backward do on the stored tensor; nothing extra is saved.  `bias` is needed
 *        by the backward pass to rebuild the pre-activation. */
int ssdk_bn_act_train_fwd(const void* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                          void* y, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, int N,
                          int C, int HW, float momentum, float eps, int act, int dtype, void* stream);
/* ... forward with the batch statistics PROVIDED: sums [C][2] = (sum x, sum x^2) over N * HW, computed by the kernel that produced
 * x (ssdk_pw_forward_stats).  No reduction pass: finalize + apply only.  (version 240) */
int ssdk_bn_act_train_fwd_sums(const void* x, const float* sums, const float* weight, const float* bias, float* running_mean,
                               float* running_var, void* y, float* save_mean, float* save_invstd, void* workspace,
                               size_t workspace_bytes, int N, int C, int HW, float momentum, float eps, int act, int dtype,
                               void* stream);
/* ... the forward pass WITHOUT its apply pass (version 240): statistics (from `sums`, or NULL: the reduction over x), running
 * statistics, save_mean / save_invstd and coef_out [C][4] = (a, b, 0, 0) with y = act(a x + b).  The depthwise convolution behind the
 * BatchNorm applies the coefficients while it stages x (ssdk_dwconv_fwd_affine / ssdk_dwconv_bwd_weight_affine below): the
 * BatchNorm's output -- the 6 x expanded tensor of an inverted-residual block (mobilenet.py:56) -- is never written. */
int ssdk_bn_act_train_stats(const void* x, const float* sums, const float* weight, const float* bias, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd, float* coef_out, void* workspace,
                            size_t workspace_bytes, int N, int C, int HW, float momentum, float eps, int dtype, void* stream);
int ssdk_bn_act_train_bwd(const void* x, const void* dy, const float* weight, const float* bias, const float* save_mean,
                          const float* save_invstd, void* dx, float* dweight, float* dbias, void* workspace,
                          size_t workspace_bytes, int N, int C, int HW, int act, int dtype, void* stream);

/* ResNet stem (nets/resnet.py:41-46): 7x7 / stride 2 / pad 3 convolution on the 3-channel image + folded BN +
 * activation -> NHWC, and the 3x3 / stride 2 / pad 1 max pooling (NHWC -> NHWC, -inf padding like torch).
 *   x  image [N,3,H,W] (in_layout NCHW) or [N,H,W,3] (NHWC), activation dtype
 *   w  [Cout][7][8][4] activation dtype: (ky, kx padded to 8, ci padded to 4), zeros in the padding slots
 *   scale / bias  fp32 [Cout] folded BN;  Cout in {32, 64} */
typedef struct ssdk_stem_desc {
  const void* x;
  const void* w;
  const float* scale;
  const float* bias;
  void* y;
  int32_t N, H, W, Cin, Cout, act, dtype, in_layout;
} ssdk_stem_desc;
int ssdk_conv_stem7(const ssdk_stem_desc* desc, void* stream);
typedef struct ssdk_pool_desc {
  const void* x;
  void* y;
  int32_t N, H, W, C, dtype, pad;
} ssdk_pool_desc;
int ssdk_maxpool3x3s2(const ssdk_pool_desc* desc, void* stream);

/* One SSD "extra" layer on a small map (ssd.py:88-99 / basic_layers.py:40-57: Conv 1x1 + BN + act, then Conv 3x3 / stride
 * 2 / pad 1 + BN + act) as ONE launch with the intermediate map in LDS: on the 8x8 ... 2x2 maps the two convolutions
 * are microseconds of work behind two launches, split-K fences and latency-bound k-loops.
 *   x [N][H][W][Cin] NHWC, w1 [Cmid][Cin] (KRSC 1x1), w2 [Cout][3][3][Cmid], scale / bias fp32 (folded BN),
 *   y [N][Ho][Wo][Cout] NHWC with Ho = (H - 1) / 2 + 1; dtype SSDK_BF16 | SSDK_F16; act SSDK_ACT_NONE | RELU | RELU6.
 * Covered: H*W <= 16 or == 64, Cin % 128 == 0, Cmid 64 | 128, Cout 128 | 256 (anything else: SSDK_E_BADARG -- run the
 * two convolutions through ssdk_conv). */
typedef struct ssdk_xpair_desc {
  const void* x;
  void* y;
  const void* w1;
  const float* scale1;
  const float* bias1;
  const void* w2;
  const float* scale2;
  const float* bias2;
  int32_t N, H, W, Cin, Cmid, Cout, act1, act2, dtype, pad;
  const void* w1_frag;  /* optional fragment-major images of w1 [Cmid][Cin] and w2 [Cout][9*Cmid] (ssdk_weight_frag_bytes); */
  const void* w2_frag;  /* both or neither */
} ssdk_xpair_desc;
int ssdk_xpair(const ssdk_xpair_desc* desc, void* stream);

/* Plan executor: a recorded forward as a list of tagged ops (topological order), replayed with one host call.
 * lane 0 ops run in order on the caller's stream.  lane 1 / lane 2 ops are forked onto the context's side stream (when
 * the side lane is on, ssdk_ctx_set_side_lane) and run concurrently with the following lane 0 ops, in list order among
 * themselves; a run of consecutive side ops forks ONCE, behind everything listed before its first op (at most 32 runs per
 * call, else everything runs in line); they use the upper half of the workspace, and everything is joined back onto
 * the caller's stream before the call returns.  lane 1: leaves (the multibox heads) -- next to the main chain an
 * underfilled grid is free, so the kernel choice may differ from the in-line one.  lane 2 (version 210): chains (the
 * towers + heads of the small pyramid levels next to the big levels' launches) -- the kernel choice for an op that may
 * underfill the chip (no split-K), made from the tag alone, so the outputs do not depend on whether the side lane is on.
 * Buffers read or written by side ops must not be reused by later ops of the list. */
enum { SSDK_OP_CONV = 0, SSDK_OP_MBCONV = 1, SSDK_OP_FUSE = 2, SSDK_OP_STEM7 = 3, SSDK_OP_POOL = 4, SSDK_OP_XPAIR = 5 };
typedef struct ssdk_op {
  int32_t kind, lane;
  ssdk_conv_desc conv;
  ssdk_mbconv_desc mb;
  ssdk_fuse_desc fuse;
  ssdk_stem_desc stem;
  ssdk_pool_desc pool;
  ssdk_xpair_desc xpair;
} ssdk_op;
int ssdk_run_ops(const ssdk_op* ops, int n, void* workspace, size_t workspace_bytes, void* stream);
int ssdk_run_ops_ctx(ssdk_ctx* ctx, const ssdk_op* ops, int n, void* workspace, size_t workspace_bytes, void* stream);
/* convenience wrapper: dense conv, NHWC in/out, single output */
int ssdk_conv_bn_act(const void* x, const void* w, const float* scale, const float* bias, int N,
                     int Cin, int H, int W, int Cout, int k, int stride, int act, int dtype,
                     int out_dtype, void* y, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDK_H_ */
