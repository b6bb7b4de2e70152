"""CPU oracle for the ssds.pytorch detection hot path (box math).

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``ssds.pytorch_amd/``) never imports anything from ``oracle/``
and has no CPU fallback.

It is a plain-numpy fp32 restatement of ``ssds/modeling/layers/box.py`` and
``ssds/modeling/layers/decoder.py`` of the reference (ShuangXieIrene/ssds.pytorch
v1.5).  Every function cites the reference lines it follows.  Parity is PINNED:
``tests/golden/*.npz`` were produced by importing the reference's own functions in
the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks this restatement against them, plus the
known-answer vectors of SURVEY.md section 4.

``oracle/ssdk_cpu.c`` is a second, independent restatement of the same functions in C behind the C-ABI's own
prototypes (``tests/test_oracle_c.py`` pins it with the same fixtures and cross-checks the two).

Conventions shared with the HIP kernels (the "contract"):

* all arithmetic is IEEE fp32 in the operation order of the reference, no FMA
  contraction (numpy never fuses);
* heads of any dtype (bf16/fp16) are upcast to fp32 first (SURVEY.md section 4:
  the reference returns fp32 outputs for every input dtype);
* ``torch.topk`` / ``torch.sort`` tie order is unspecified in the reference;
  the contract is *(score descending, flat index ascending)*, i.e. a stable sort.
  ``torch.max(dim)`` ties resolve to the first (lowest) index, as the CPU
  reference does.
"""
from collections import OrderedDict

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# anchors
# --------------------------------------------------------------------------- #
def configure_ratio_scale(num_featmaps, ratios, scales):
    """box.py:8-43 -- normalise cfg ASPECT_RATIOS / SIZES to per-level lists."""
    if len(scales) != num_featmaps:
        raise ValueError(
            "cfg.SIZES is not correct,"
            "the len of cfg.SIZES should equal to num layers({}) or 2, but it is {}".format(
                num_featmaps, len(scales)
            )
        )
    scales = list(scales)
    for i in range(num_featmaps):
        if not isinstance(scales[i], list):
            scales[i] = [scales[i]]
    if isinstance(ratios[0], list):
        if len(ratios) != num_featmaps:
            raise ValueError(
                "When cfg.ASPECT_RATIOS contains list for each layer,"
                "Len of cfg.ASPECT_RATIOS should equal to num layers({}), but it is {}".format(
                    num_featmaps, len(ratios)
                )
            )
    else:
        ratios = [ratios for _ in range(num_featmaps)]
    return ratios, scales


def generate_anchors(stride, ratio_vals, scales_vals):
    """box.py:46-58 -- A = len(ratios)*len(scales) base boxes (ltrb), scale-major."""
    nr, ns = len(ratio_vals), len(scales_vals)
    scales = np.repeat(np.asarray(scales_vals, F32), nr).reshape(-1, 1)  # :49-50
    ratios = np.tile(np.asarray(ratio_vals, F32), ns)  # :51
    wh = np.full((nr * ns, 2), F32(stride), F32)  # :53
    ws = np.round(np.sqrt(wh[:, 0] * wh[:, 1] / ratios))  # :54 (half-to-even)
    dwh = np.stack([ws, np.round(ws * ratios)], 1)  # :55
    xy1 = F32(0.5) * (wh - dwh * scales)  # :56
    xy2 = F32(0.5) * (wh + dwh * scales) - F32(1)  # :57
    return np.concatenate([xy1, xy2], 1).astype(F32)


# --------------------------------------------------------------------------- #
# encode / decode of one box
# --------------------------------------------------------------------------- #
def box2delta(boxes, anchors):
    """box.py:61-71."""
    boxes = np.asarray(boxes, F32)
    anchors = np.asarray(anchors, F32)
    anchors_wh = anchors[:, 2:] - anchors[:, :2] + F32(1)
    anchors_ctr = anchors[:, :2] + F32(0.5) * anchors_wh
    boxes_wh = boxes[:, 2:] - boxes[:, :2] + F32(1)
    boxes_ctr = boxes[:, :2] + F32(0.5) * boxes_wh
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.concatenate(
            [(boxes_ctr - anchors_ctr) / anchors_wh, np.log(boxes_wh / anchors_wh)], 1
        ).astype(F32)


def delta2box(deltas, anchors, size, stride):
    """box.py:74-87.  ``size`` = [W, H] of the feature map."""
    deltas = np.asarray(deltas, F32)
    anchors = np.asarray(anchors, F32)
    anchors_wh = anchors[:, 2:] - anchors[:, :2] + F32(1)
    ctr = anchors[:, :2] + F32(0.5) * anchors_wh
    pred_ctr = deltas[:, :2] * anchors_wh + ctr
    with np.errstate(over="ignore"):
        pred_wh = np.exp(deltas[:, 2:]) * anchors_wh
    m = np.zeros([2], F32)
    M = np.asarray([size], F32) * F32(stride) - F32(1)

    def clamp(t):
        return np.maximum(m, np.minimum(t, M))

    return np.concatenate(
        [clamp(pred_ctr - F32(0.5) * pred_wh), clamp(pred_ctr + F32(0.5) * pred_wh - F32(1))], 1
    ).astype(F32)


# --------------------------------------------------------------------------- #
# target assignment
# --------------------------------------------------------------------------- #
def get_sample_region(boxes, stride, anchor_points, radius=1.5):
    """box.py:90-113 -- ATSS centre-sampling mask [W, H, G]."""
    stride = F32(stride * radius)
    center = (boxes[:, :2] + boxes[:, 2:]) / F32(2)
    center_boxes = np.concatenate((center - stride, center + stride), -1)
    lt = anchor_points[:, :, None, :] - np.maximum(center_boxes[:, :2], boxes[:, :2])[None, None]
    rb = np.minimum(center_boxes[:, 2:], boxes[:, 2:])[None, None] - anchor_points[:, :, None, :]
    cb = np.concatenate((lt, rb), -1)
    return cb.min(-1) > 0


def snap_to_anchors_by_iou(
    boxes, size, stride, anchors, num_classes, match, center_sampling_radius=0
):
    """box.py:116-226 (``is_centerness=False`` branch).

    boxes [G,5] = (x, y, w, h, label) abs pixels; size = [W*stride, H*stride].
    Returns cls_target [A,C,H,W], box_target [A,4,H,W], depth [A,1,H,W] (fp32).
    """
    anchors = np.asarray(anchors, F32)
    A = anchors.shape[0]
    width, height = int(size[0] / stride), int(size[1] / stride)  # :131
    boxes = np.asarray(boxes, F32).reshape(-1, 5)
    if boxes.size == 0:  # :133-146
        return (
            np.zeros([A, num_classes, height, width], F32),
            np.zeros([A, 4, height, width], F32),
            np.zeros([A, 1, height, width], F32),
        )
    boxes, classes = boxes[:, :4], boxes[:, 4:]  # :148
    match_threshold, unmatch_threshold = F32(match[0]), F32(match[1])  # :149

    # grid anchors, idx = (a*W + ix)*H + iy   (:151-159, meshgrid 'ij' -> x major)
    xs = np.arange(0, size[0], stride, dtype=F32)
    ys = np.arange(0, size[1], stride, dtype=F32)
    x, y = np.meshgrid(xs, ys, indexing="ij")  # [W, H]
    xyxy = np.stack((x, y, x, y), 2)[None]  # [1, W, H, 4]
    ganchors = (xyxy + anchors.reshape(-1, 1, 1, 4)).reshape(-1, 4)

    # IoU (+1 pixel convention, no epsilon)   :162-168
    boxes = np.concatenate([boxes[:, :2], boxes[:, :2] + boxes[:, 2:] - F32(1)], 1)
    xy1 = np.maximum(ganchors[:, None, :2], boxes[:, :2])
    xy2 = np.minimum(ganchors[:, None, 2:], boxes[:, 2:])
    d = np.clip(xy2 - xy1 + F32(1), 0, None)
    inter = d[..., 0] * d[..., 1]
    bwh = boxes[:, 2:] - boxes[:, :2] + F32(1)
    boxes_area = bwh[:, 0] * bwh[:, 1]
    awh = ganchors[:, 2:] - ganchors[:, :2] + F32(1)
    anchors_area = awh[:, 0] * awh[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        overlap = inter / (anchors_area[:, None] + boxes_area - inter)

    # best GT per anchor (first max wins)     :171-175
    indices = overlap.argmax(1)
    overlap = overlap[np.arange(overlap.shape[0]), indices]
    box_target = box2delta(boxes[indices], ganchors)
    box_target = box_target.reshape(A, width, height, 4).transpose(0, 3, 2, 1)  # [A,4,H,W]

    # depth: -1 ignore / 0 background / label+1 foreground   :177-182
    depth = np.full_like(overlap, -1, dtype=F32)
    depth[overlap < unmatch_threshold] = 0
    fg = overlap >= match_threshold
    depth[fg] = classes[indices][fg].reshape(-1) + F32(1)
    depth = depth.reshape(A, width, height)
    if center_sampling_radius > 0:  # :184-191
        anchor_points = np.stack((x, y), 2) + F32(stride // 2)
        inside = (
            get_sample_region(boxes, stride, anchor_points, center_sampling_radius)
            .astype(F32)
            .max(-1)
        )
        depth = np.minimum(depth, inside[None])
    depth = depth.transpose(0, 2, 1)  # [A,H,W]

    # one-hot classes, background column dropped   :195-207
    cls = classes[indices].astype(np.int64).reshape(-1)
    cls[overlap < unmatch_threshold] = num_classes
    cls_target = np.zeros((ganchors.shape[0], num_classes + 1), F32)
    cls_target[np.arange(cls.shape[0]), cls] = 1
    cls_target = cls_target[:, :num_classes].reshape(A, width, height, num_classes)
    cls_target = cls_target.transpose(0, 3, 2, 1)  # [A,C,H,W]

    return (
        np.ascontiguousarray(cls_target, F32).reshape(A, num_classes, height, width),
        np.ascontiguousarray(box_target, F32).reshape(A, 4, height, width),
        np.ascontiguousarray(depth, F32).reshape(A, 1, height, width),
    )


INF = 100000  # box.py:5


def snap_to_anchors_by_scale(
    boxes, size, stride, anchors, num_classes, match, center_sampling_radius=0
):
    """box.py:229-359 (``is_centerness=False`` branch; the other branch reads an undefined name and
    cannot run).  match = [lower, upper] multipliers of sqrt(anchor area) for this level.
    Returns cls_target [A,C,H,W], box_target [A,4,H,W], depth [A,1,H,W] (fp32)."""
    anchors = np.asarray(anchors, F32)
    A = anchors.shape[0]
    width, height = int(size[0] / stride), int(size[1] / stride)  # :244
    boxes = np.asarray(boxes, F32).reshape(-1, 5)
    if boxes.size == 0:  # :246-260
        return (
            np.zeros([A, num_classes, height, width], F32),
            np.zeros([A, 4, height, width], F32),
            np.zeros([A, 1, height, width], F32),
        )
    boxes, classes = boxes[:, :4], boxes[:, 4:]  # :262

    # per-anchor size range   :265-268
    anchors_wh = anchors[:, 2:] - anchors[:, :2] + F32(1)
    anchors_size = np.sqrt(anchors_wh[:, 0] * anchors_wh[:, 1]).astype(F32)[:, None, None]
    lower = np.maximum(F32(match[0]) * anchors_size, F32(-1))
    upper = F32(match[1]) * anchors_size

    # grid anchors (x major) and anchor points   :271-281
    xs = np.arange(0, size[0], stride, dtype=F32)
    ys = np.arange(0, size[1], stride, dtype=F32)
    x, y = np.meshgrid(xs, ys, indexing="ij")  # [W, H]
    xyxy = np.stack((x, y, x, y), 2)[None]
    ganchors = (xyxy + anchors.reshape(-1, 1, 1, 4)).reshape(-1, 4)
    anchor_points = np.stack((x, y), 2) + F32(stride // 2)  # [W, H, 2]

    boxes = np.concatenate([boxes[:, :2], boxes[:, :2] + boxes[:, 2:] - F32(1)], 1)  # :284
    bwh = boxes[:, 2:] - boxes[:, :2] + F32(1)
    boxes_area = np.sqrt(bwh[:, 0] * bwh[:, 1]).astype(F32)  # :285 (a sqrt-area)

    G = boxes.shape[0]
    if center_sampling_radius > 0:  # :288-299; get_sample_region runs with its default radius 1.5
        cared = (boxes_area >= lower) & (boxes_area <= upper)  # [A,1,G]
        inside = get_sample_region(boxes, stride, anchor_points).reshape(-1, G)  # [W*H, G]
    else:  # :300-311
        ap = anchor_points.reshape(-1, 2)
        lt = ap[:, None, :] - boxes[:, :2]
        rb = boxes[:, 2:] - ap[:, None, :]
        reg = np.concatenate([lt, rb], -1)  # [W*H, G, 4]
        mx = reg.max(-1)
        cared = (mx >= lower) & (mx <= upper)  # [A, W*H, G]
        inside = reg.min(-1) > 0
    mask = (cared & inside).reshape(-1, G)  # :315  [A*W*H, G]
    area = np.tile(boxes_area, (mask.shape[0], 1))
    area[~mask] = INF
    anymask = mask.any(1)
    indices = area.argmin(1)  # first minimum wins

    box_target = box2delta(boxes[indices], ganchors)  # :323
    box_target = box_target.reshape(A, width, height, 4).transpose(0, 3, 2, 1)

    depth = np.zeros(mask.shape[0], F32)  # :328-331
    depth[anymask] = classes[indices][anymask].reshape(-1) + F32(1)
    depth = depth.reshape(A, width, height).transpose(0, 2, 1)

    cls = classes[indices].astype(np.int64).reshape(-1)  # :334-346
    cls[~anymask] = num_classes
    cls_target = np.zeros((ganchors.shape[0], num_classes + 1), F32)
    cls_target[np.arange(cls.shape[0]), cls] = 1
    cls_target = cls_target[:, :num_classes].reshape(A, width, height, num_classes).transpose(0, 3, 2, 1)

    return (
        np.ascontiguousarray(cls_target, F32).reshape(A, num_classes, height, width),
        np.ascontiguousarray(box_target, F32).reshape(A, 4, height, width),
        np.ascontiguousarray(depth, F32).reshape(A, 1, height, width),
    )


def extract_targets(
    targets, anchors, classes, stride, size, match=(0.5, 0.4), center_sampling_radius=0
):
    """box.py:362-405.  ``match[0]`` float -> IoU matching; list -> one [lower, upper] scale range per
    level, selected by the position of ``stride`` among the anchor keys (:389).

    targets [B,G,5] padded with label -1; anchors = OrderedDict{stride: [A,4]};
    size = (h, w) of the level's feature map.
    """
    by_scale = isinstance(match[0], (list, tuple))
    if not by_scale and not isinstance(match[0], float):
        raise ValueError("unvalidate match param")
    targets = np.asarray(targets, F32)
    outs = ([], [], [])
    for target in targets:
        target = target[target[:, -1] > -1]  # :375
        if by_scale:
            snapped = snap_to_anchors_by_scale(
                target,
                [s * stride for s in size[::-1]],
                stride,
                anchors[stride],
                classes,
                match[list(anchors).index(stride)],
                center_sampling_radius,
            )
        else:
            snapped = snap_to_anchors_by_iou(
                target,
                [s * stride for s in size[::-1]],  # :379
                stride,
                anchors[stride],
                classes,
                match,
                center_sampling_radius,
            )
        for lst, s in zip(outs, snapped):
            lst.append(s)
    return tuple(np.stack(o) for o in outs)


# --------------------------------------------------------------------------- #
# decode (threshold + top-k + box decode + centre rescoring)
# --------------------------------------------------------------------------- #
def decode(
    all_cls_head, all_box_head, stride=1, threshold=0.05, top_n=1000, anchors=None, rescore=True
):
    """box.py:408-477.  Returns fp32 (scores [B,top_n], boxes [B,top_n,4], classes [B,top_n])."""
    cls_all = np.asarray(all_cls_head, F32)
    box_all = np.asarray(all_box_head, F32)
    anchors = np.asarray(anchors, F32)
    A = anchors.shape[0]
    C = cls_all.shape[1] // A
    H, W = cls_all.shape[-2:]
    B = cls_all.shape[0]
    out_scores = np.zeros((B, top_n), F32)
    out_boxes = np.zeros((B, top_n, 4), F32)
    out_classes = np.zeros((B, top_n), F32)
    thr = F32(threshold)

    for b in range(B):  # :435
        cls_head = cls_all[b].reshape(-1)
        keep = np.nonzero(cls_head >= thr)[0]  # :440
        if keep.size == 0:
            continue
        scores = cls_head[keep]
        k = min(top_n, keep.size)
        order = np.argsort(-scores, kind="stable")[:k]  # :446 (score desc, index asc)
        scores = scores[order]
        indices = keep[order]
        classes = (indices // W // H) % C  # :448
        x = indices % W  # :452
        y = (indices // W) % H
        a = indices // C // H // W
        boxes = box_all[b].reshape(A, 4, H, W)[a, :, y, x]  # :455-456

        grid = np.stack([x, y, x, y], 1).astype(F32) * F32(stride) + anchors[a, :]  # :459-462
        boxes = delta2box(boxes, grid, [W, H], stride)
        if rescore:  # :464-471
            grid_center = (grid[:, :2] + grid[:, 2:]) / F32(2)
            lt = np.abs(grid_center - boxes[:, :2])
            rb = np.abs(boxes[:, 2:] - grid_center)
            with np.errstate(divide="ignore", invalid="ignore"):
                r = np.minimum(lt, rb) / np.maximum(lt, rb)
                # torch.min/max propagate NaN the same way np.minimum/maximum do
                centerness = np.sqrt(r[:, 0] * r[:, 1])
            scores = scores * centerness

        out_scores[b, :k] = scores
        out_boxes[b, :k] = boxes
        out_classes[b, :k] = classes.astype(F32)
    return out_scores, out_boxes, out_classes


# --------------------------------------------------------------------------- #
# nms (greedy, class aware, optional DIoU with top-left distance)
# --------------------------------------------------------------------------- #
def nms(all_scores, all_boxes, all_classes, nms=0.5, ndetections=100, using_diou=True):
    """box.py:480-546, restated literally (compaction every iteration)."""
    all_scores = np.asarray(all_scores, F32)
    all_boxes = np.asarray(all_boxes, F32)
    all_classes = np.asarray(all_classes, F32)
    B = all_scores.shape[0]
    out_scores = np.zeros((B, ndetections), F32)
    out_boxes = np.zeros((B, ndetections, 4), F32)
    out_classes = np.zeros((B, ndetections), F32)
    thr = F32(nms)
    eps = F32(1e-7)

    for b in range(B):
        keep = np.nonzero(all_scores[b].reshape(-1) > 0)[0]  # :496 (drops NaN too)
        scores = all_scores[b, keep]
        boxes = all_boxes[b, keep, :].reshape(-1, 4)
        classes = all_classes[b, keep]
        if scores.size == 0:
            continue
        order = np.argsort(-scores, kind="stable")  # :505
        scores, boxes, classes = scores[order], boxes[order], classes[order]
        areas = (boxes[:, 2] - boxes[:, 0] + F32(1)) * (boxes[:, 3] - boxes[:, 1] + F32(1))  # :507

        i = 0
        for i in range(ndetections):  # :512
            if i >= scores.size:  # :513 (keep bookkeeping is harmless)
                i -= 1
                break
            xy1 = np.maximum(boxes[:, :2], boxes[i, :2])
            xy2 = np.minimum(boxes[:, 2:], boxes[i, 2:])
            d = np.clip(xy2 - xy1 + F32(1), 0, None)
            inter = d[:, 0] * d[:, 1]
            iou = inter / (areas + areas[i] - inter + eps)  # :521
            if using_diou:  # :523-530 (top-left corner distance, not centre)
                outer_lt = np.minimum(boxes[:, :2], boxes[i, :2])
                outer_rb = np.maximum(boxes[:, 2:], boxes[i, 2:])
                dl = boxes[:, :2] - boxes[i, :2]
                inter_diag = dl[:, 0] * dl[:, 0] + dl[:, 1] * dl[:, 1]
                do = outer_rb - outer_lt
                outer_diag = (do[:, 0] * do[:, 0] + do[:, 1] * do[:, 1]) + eps
                iou = np.clip(iou - inter_diag / outer_diag, F32(-1.0), F32(1.0))
            criterion = (scores > scores[i]) | (iou <= thr) | (classes != classes[i])  # :532
            criterion[i] = True
            scores = scores[criterion]
            boxes = boxes[criterion]
            classes = classes[criterion]
            areas = areas[criterion]
        out_scores[b, : i + 1] = scores[: i + 1]
        out_boxes[b, : i + 1] = boxes[: i + 1]
        out_classes[b, : i + 1] = classes[: i + 1]
    return out_scores, out_boxes, out_classes


# --------------------------------------------------------------------------- #
# Decoder (decode every level -> concat -> nms)
# --------------------------------------------------------------------------- #
class Decoder(object):
    """decoder.py:15-49."""

    def __init__(self, conf_threshold, nms_threshold, top_n, top_n_per_level, rescore, use_diou):
        self.conf_threshold = conf_threshold
        self.nms_threshold = nms_threshold
        self.top_n = top_n
        self.top_n_per_level = top_n_per_level
        self.rescore = rescore
        self.use_diou = use_diou

    def decode_levels(self, loc, conf, anchors):
        decoded = [
            decode(c, l, stride, self.conf_threshold, self.top_n_per_level, anchor, rescore=self.rescore)
            for l, c, (stride, anchor) in zip(loc, conf, anchors.items())
        ]
        return [np.concatenate(t, 1) for t in zip(*decoded)]

    def __call__(self, loc, conf, anchors):
        decoded = self.decode_levels(loc, conf, anchors)
        return nms(*decoded, self.nms_threshold, self.top_n, using_diou=self.use_diou)


def make_anchors(strides, ratios, scales):
    """model_builder.py:43-49 -- OrderedDict{stride: anchors[A,4]} in level order."""
    return OrderedDict(
        (strides[i], generate_anchors(strides[i], ratios[i], scales[i])) for i in range(len(strides))
    )


# --------------------------------------------------------------------------- #
# detector front door (ssds/ssds.py:47-57)
# --------------------------------------------------------------------------- #
def preprocess(imgs, mean, std):
    """HWC -> CHW (when ``shape[3] == 3``), float32, ``(x - mean) / std`` in that order -> fp32 [N,C,H,W]."""
    imgs = np.asarray(imgs)
    if imgs.shape[3] == 3:  # ssds.py:53-54
        imgs = imgs.transpose(0, 3, 1, 2)
    x = imgs.astype(F32)
    c = x.shape[1]
    m = np.broadcast_to(np.asarray(mean, F32).reshape(-1), (c,)).reshape(1, c, 1, 1)
    s = np.broadcast_to(np.asarray(std, F32).reshape(-1), (c,)).reshape(1, c, 1, 1)
    return ((x - m) / s).astype(F32)
