"""Seeded synthetic inputs for the golden fixtures (numpy only, no torch, no reference).

Shared by ``make_golden.py`` (which feeds these inputs to the *reference* functions
imported from /root/reference and stores the outputs in ``*.npz``) and by the tests
(which regenerate the same inputs and compare the oracle / the HIP kernels with the
stored reference outputs).  Inputs are regenerated from seeds (``RandomState`` streams
are stable) so the fixtures hold only outputs + an input checksum.
"""
import zlib
from collections import OrderedDict

import numpy as np

F32 = np.float32


def checksum(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def make_unique_above(arr, thr):
    """Nudge duplicate values >= thr upward by ulps until all passing values are distinct
    (the reference's topk/sort tie order is unspecified; golden inputs avoid ties)."""
    flat = arr.reshape(-1).copy()
    thr = F32(thr)
    for _ in range(64):
        idx = np.nonzero(flat >= thr)[0]
        vals = flat[idx]
        order = np.argsort(vals, kind="stable")
        sv = vals[order]
        dup = np.nonzero(sv[1:] == sv[:-1])[0] + 1
        if dup.size == 0:
            return flat.reshape(arr.shape)
        flat[idx[order[dup]]] = np.nextafter(sv[dup], F32(np.inf))
    raise RuntimeError("could not de-duplicate")


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> returns (uint16 bits, fp32 upcast)."""
    u = np.ascontiguousarray(x, F32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    b = ((u + r) >> 16).astype(np.uint16)
    return b, (b.astype(np.uint32) << 16).view(F32)


def f16_round(x):
    h = np.ascontiguousarray(x, F32).astype(np.float16)
    return h.view(np.uint16), h.astype(F32)


# ----------------------------------------------------------------------------- decode
# name: (seed, B, A, C, H, W, stride, thr, top_n, rescore, kind)
DECODE_CASES = OrderedDict(
    [
        ("small", (11, 2, 3, 5, 7, 9, 8, 0.05, 50, True, "uniform4")),
        ("small_norescore", (12, 2, 3, 5, 7, 9, 8, 0.05, 50, False, "uniform4")),
        ("ragged_fewpass", (13, 3, 2, 4, 5, 3, 16, 0.6, 40, True, "fewpass")),
        ("topn_gt_cand", (14, 1, 1, 2, 4, 4, 32, 0.3, 1000, True, "uniform4")),
        ("ssd300_l0", (15, 2, 6, 80, 19, 19, 15, 0.01, 300, True, "logit")),
        ("ssd300_l1", (16, 2, 6, 80, 10, 10, 30, 0.01, 300, True, "logit")),
        ("ssd300_l5", (17, 2, 6, 80, 1, 1, 300, 0.01, 300, True, "logit")),
        ("ssd512_l0", (18, 1, 6, 80, 32, 32, 16, 0.01, 300, True, "logit")),
        ("fpn_a9", (19, 1, 9, 20, 12, 20, 8, 0.05, 100, True, "logit")),
        ("bf16_l1", (20, 2, 6, 80, 16, 16, 32, 0.01, 300, True, "bf16")),
        ("f16_l2", (21, 2, 6, 80, 8, 8, 64, 0.01, 300, True, "f16")),
        ("allpass", (22, 1, 2, 3, 6, 6, 8, 0.0, 20, True, "uniform4")),
    ]
)

SSD_RATIOS = [1, 2, 0.5]
SSD_SCALES = [2.0, 2.828]


def anchors_for(A, stride, gen):
    """Pick a ratio/scale set giving A anchors; ``gen`` is a generate_anchors callable."""
    if A == 6:
        return gen(stride, SSD_RATIOS, SSD_SCALES)
    if A == 9:
        return gen(stride, [1, 2, 0.5], [2.0, 2.52, 3.175])
    if A == 3:
        return gen(stride, [1, 2, 0.5], [2.0])
    if A == 2:
        return gen(stride, [1], [2.0, 1.0])
    if A == 1:
        return gen(stride, [1], [2.0])
    raise ValueError(A)


def decode_inputs(name):
    """-> dict(cls fp32 (the upcast values), box fp32, raw (uint16 bits or None), dtype, params)"""
    seed, B, A, C, H, W, stride, thr, top_n, rescore, kind = DECODE_CASES[name]
    rs = np.random.RandomState(seed)
    n = B * A * C * H * W
    dtype = "f32"
    raw_cls = raw_box = None
    if kind == "uniform4":
        cls = (rs.random_sample(n) ** 4).astype(F32)
    elif kind == "fewpass":
        cls = (rs.random_sample(n) * 0.5).astype(F32)
        hot = rs.choice(n // B * 2, size=7, replace=False)  # only images 0,1 get candidates
        cls[hot] = (0.6 + 0.39 * rs.random_sample(7)).astype(F32)
    elif kind in ("logit", "bf16", "f16"):
        cls = sigmoid(rs.standard_normal(n).astype(F32) * F32(1.5) - F32(4.6))
    else:
        raise ValueError(kind)
    box = (rs.standard_normal(B * A * 4 * H * W) * 0.5).astype(F32)
    if kind == "bf16":
        dtype = "bf16"
        # distinct bf16 values above threshold: keep first occurrence of each passing value
        bits, up = bf16_round(cls)
        order = rs.permutation(n)
        seen = set()
        for i in order:
            if up[i] >= F32(thr):
                if bits[i] in seen:
                    bits[i] = 0x3000  # tiny positive, below every threshold used
                else:
                    seen.add(bits[i])
        raw_cls = bits
        cls = (bits.astype(np.uint32) << 16).view(F32)
        raw_box, box = bf16_round(box)
    elif kind == "f16":
        dtype = "f16"
        bits, up = f16_round(cls)
        order = rs.permutation(n)
        seen = set()
        for i in order:
            if up[i] >= F32(thr):
                if bits[i] in seen:
                    bits[i] = 0x0400
                else:
                    seen.add(bits[i])
        raw_cls = bits
        cls = bits.view(np.float16).astype(F32)
        raw_box, box = f16_round(box)
    else:
        cls = make_unique_above(cls, thr)
    cls = cls.reshape(B, A * C, H, W)
    box = box.reshape(B, A * 4, H, W)
    if raw_cls is not None:
        raw_cls = raw_cls.reshape(B, A * C, H, W)
        raw_box = raw_box.reshape(B, A * 4, H, W)
    return dict(
        cls=cls, box=box, raw_cls=raw_cls, raw_box=raw_box, dtype=dtype,
        A=A, C=C, stride=stride, thr=thr, top_n=top_n, rescore=rescore,
    )


# ----------------------------------------------------------------------------- nms
# name: (seed, B, N, nclass, ncluster, thr, ndet, diou, kind)
NMS_CASES = OrderedDict(
    [
        ("clustered_diou", (31, 3, 400, 3, 12, 0.5, 100, True, "cluster")),
        ("clustered_iou", (32, 3, 400, 3, 12, 0.5, 100, False, "cluster")),
        ("dense_trunc", (33, 2, 1800, 80, 40, 0.6, 100, True, "cluster")),
        ("few_survivors", (34, 2, 300, 1, 3, 0.3, 100, True, "cluster")),
        ("ndet_small", (35, 2, 500, 4, 50, 0.6, 10, True, "cluster")),
        ("with_zeros_nan", (36, 4, 256, 2, 8, 0.5, 50, True, "holes")),
        ("single", (37, 1, 1, 1, 1, 0.5, 5, True, "cluster")),
        ("heavy_one_class", (38, 1, 1800, 1, 6, 0.6, 100, True, "cluster")),
    ]
)


def nms_inputs(name):
    seed, B, N, nclass, ncluster, thr, ndet, diou, kind = NMS_CASES[name]
    rs = np.random.RandomState(seed)
    centers = rs.random_sample((B, ncluster, 2)) * 400 + 50
    sizes = rs.random_sample((B, ncluster, 2)) * 120 + 20
    which = rs.randint(0, ncluster, size=(B, N))
    bidx = np.arange(B)[:, None]
    c = centers[bidx, which] + rs.standard_normal((B, N, 2)) * 6
    s = sizes[bidx, which] * (1 + rs.standard_normal((B, N, 2)) * 0.1)
    s = np.abs(s) + 1
    boxes = np.concatenate([c - s / 2, c + s / 2], -1)
    boxes = np.clip(np.round(boxes * 4) / 4, 0, 511).astype(F32)
    scores = rs.random_sample((B, N)).astype(F32)
    scores = np.stack([make_unique_above(scores[b], 0.0) for b in range(B)])
    classes = rs.randint(0, nclass, size=(B, N)).astype(F32)
    if kind == "holes":
        scores[0, ::3] = 0
        scores[1, :] = 0  # an image without candidates
        scores[2, 5] = np.nan
        scores[2, 17] = np.nan
        scores[3, ::2] = -0.25
    return dict(scores=scores, boxes=boxes, classes=classes, thr=thr, ndet=ndet, diou=diou)


# ----------------------------------------------------------------------------- Decoder
# name: (seed, B, image, C, maps, strides, thr, nms, top_n, per_level, rescore, diou)
DECODER_CASES = OrderedDict(
    [
        ("ssd300", (41, 2, 300, 80, [19, 10, 5, 3, 2, 1], [15, 30, 60, 100, 150, 300],
                    0.01, 0.6, 100, 300, True, True)),
        ("ssd300_plain", (42, 1, 300, 20, [19, 10, 5, 3, 2, 1], [15, 30, 60, 100, 150, 300],
                          0.3, 0.45, 20, 50, False, False)),
        ("ssd512", (43, 1, 512, 80, [32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512],
                    0.01, 0.6, 100, 300, True, True)),
    ]
)


def decoder_inputs(name, gen):
    seed, B, image, C, maps, strides, thr, nmsthr, top_n, per_level, rescore, diou = DECODER_CASES[name]
    rs = np.random.RandomState(seed)
    A = 6
    loc, conf = [], []
    for m in maps:
        c = sigmoid(rs.standard_normal(B * A * C * m * m).astype(F32) * F32(1.5) - F32(4.6))
        c = make_unique_above(c, thr).reshape(B, A * C, m, m)
        l = (rs.standard_normal(B * A * 4 * m * m) * 0.5).astype(F32).reshape(B, A * 4, m, m)
        conf.append(c)
        loc.append(l)
    anchors = OrderedDict((s, gen(s, SSD_RATIOS, SSD_SCALES)) for s in strides)
    return dict(loc=loc, conf=conf, anchors=anchors, thr=thr, nms=nmsthr, top_n=top_n,
                per_level=per_level, rescore=rescore, diou=diou)


# ----------------------------------------------------------------------------- extract_targets
# name: (seed, B, maxG, C, H, W, stride, A, match, radius)
MATCH_CASES = OrderedDict(
    [
        ("kat_layout", (0, 1, 2, 7, 2, 3, 8, 2, (0.5, 0.4), 0)),
        ("ssd300_l0", (51, 3, 8, 80, 19, 19, 15, 6, (0.5, 0.4), 0)),
        ("ssd300_l2", (52, 3, 8, 80, 5, 5, 60, 6, (0.5, 0.4), 0)),
        ("nonsquare", (53, 2, 5, 11, 6, 10, 16, 3, (0.5, 0.4), 0)),
        ("lowthr", (54, 2, 12, 4, 8, 8, 32, 6, (0.3, 0.2), 0)),
        ("center_sampling", (55, 2, 6, 5, 8, 8, 16, 3, (0.5, 0.4), 1.5)),
        ("one_by_one", (56, 2, 3, 6, 1, 1, 300, 6, (0.5, 0.4), 0)),
    ]
)


# scale-range assignment (snap_to_anchors_by_scale): match = (lower, upper) multipliers of sqrt(anchor area)
SCALE_MATCH_CASES = OrderedDict(
    [
        ("s_ssd300_l0", (61, 3, 8, 80, 19, 19, 15, 6, (0.5, 4.0), 0)),
        ("s_ssd300_l2", (62, 3, 8, 80, 5, 5, 60, 6, (0.0, 2.0), 0)),
        ("s_nonsquare", (63, 2, 5, 11, 6, 10, 16, 3, (-1.0, 1000.0), 0)),
        ("s_center", (64, 3, 6, 5, 8, 8, 16, 3, (0.5, 4.0), 1.5)),
        ("s_center_r3", (65, 2, 6, 5, 8, 8, 16, 3, (0.2, 8.0), 3.0)),  # radius value is ignored (always 1.5)
        ("s_ties", (66, 2, 6, 9, 8, 8, 16, 2, (0.0, 100.0), 0)),
        ("s_one_by_one", (67, 2, 3, 6, 1, 1, 300, 6, (0.0, 4.0), 0)),
        ("s_single_anchor", (68, 2, 4, 3, 7, 5, 8, 1, (0.0, 6.0), 0)),
    ]
)


def match_inputs(name, gen, table=None):
    table = table or (SCALE_MATCH_CASES if name in SCALE_MATCH_CASES else MATCH_CASES)
    seed, B, maxG, C, H, W, stride, A, match, radius = table[name]
    anchors = anchors_for(A, stride, gen)
    if name == "kat_layout":
        targets = np.array([[[12, 4, 16, 16, 5], [-1, -1, -1, -1, -1]]], F32)
    else:
        rs = np.random.RandomState(seed)
        targets = np.full((B, maxG, 5), -1, F32)
        iw, ih = W * stride, H * stride
        for b in range(B):
            g = rs.randint(0, maxG + 1) if b > 0 else maxG  # image 1.. may be empty/ragged
            if b == B - 1:
                g = 0 if B > 2 else g
            for j in range(g):
                # GT sized around the anchor scale so that matches exist
                w = stride * (1.2 + 2.5 * rs.random_sample())
                h = w * (0.5 + 1.5 * rs.random_sample())
                w, h = min(w, iw), min(h, ih)
                x = rs.random_sample() * max(iw - w, 1)
                y = rs.random_sample() * max(ih - h, 1)
                targets[b, j] = [np.floor(x), np.floor(y), np.ceil(w), np.ceil(h), rs.randint(0, C)]
        if name == "s_ties":  # equal-area boxes at the same place with different labels: the first one wins
            targets[0, 1, :4] = targets[0, 0, :4]
            targets[0, 1, 4] = (targets[0, 0, 4] + 1) % C
            targets[0, 3, 2:4] = targets[0, 2, 3:1:-1]  # transposed box: same area, different shape
    return dict(targets=targets, anchors=anchors, C=C, stride=stride, size=(H, W),
                match=match, radius=radius)


# eval-epoch mAP bookkeeping (MeanAveragePrecision): name -> (seed, batches, B, D, maxG, C, conf_thr, iou_thr)
MAP_CASES = OrderedDict(
    [
        ("m_small", (71, 1, 2, 8, 4, 3, 0.3, 0.5)),
        ("m_coco_like", (72, 3, 4, 100, 24, 80, 0.01, 0.5)),
        ("m_crowded", (73, 2, 3, 64, 12, 2, 0.05, 0.6)),       # many detections per box: only the first is a TP
        ("m_missing_cls", (74, 2, 2, 32, 6, 6, 0.2, 0.45)),    # classes without ground truth / without detections
        ("m_wide", (75, 1, 2, 600, 40, 20, 0.02, 0.5)),        # D > one pass of the workgroup
    ]
)


def map_inputs(name):
    """Per batch: scores [B,D] (descending, zero padded), boxes [B,D,4] ltrb, classes [B,D], targets [B,G,5]
    (ltrb + label, -1 padding) -- detections are jittered copies of the ground truth plus random false positives,
    the shape of a Decoder output.  Scores are distinct within an epoch (ranking ties are outside the contract)."""
    seed, batches, B, D, maxG, C, conf_thr, iou_thr = MAP_CASES[name]
    rs = np.random.RandomState(seed)
    pool = rs.permutation(200000)[: batches * B * D].astype(np.float64)
    all_scores = (0.02 + 0.97 * (pool + 1) / 200001.0).astype(F32)
    assert len(np.unique(all_scores)) == len(all_scores)
    out, k = [], 0
    for _ in range(batches):
        scores = np.zeros((B, D), F32)
        boxes = np.zeros((B, D, 4), F32)
        classes = np.zeros((B, D), F32)
        targets = np.full((B, maxG, 5), -1, F32)
        for b in range(B):
            g = maxG if b == 0 else rs.randint(0, maxG + 1)
            gt_cls_hi = max(1, C - 2) if name == "m_missing_cls" else C  # the last classes never have ground truth
            for j in range(g):
                w, h = 20 + 200 * rs.random_sample(2)
                x, y = rs.random_sample(2) * 300
                targets[b, j] = [x, y, x + w, y + h, rs.randint(0, gt_cls_hi)]
            n = rs.randint(D // 2, D + 1)
            dets = []
            for i in range(n):
                if g > 0 and rs.random_sample() < 0.7:
                    t = targets[b, rs.randint(0, g)]
                    jit = (rs.random_sample(4) - 0.5) * (0.5 if name == "m_crowded" else 0.8) * (t[2] - t[0])
                    bx = t[:4] + jit.astype(F32)
                    cl = t[4] if rs.random_sample() < 0.85 else rs.randint(0, C)
                else:
                    w, h = 20 + 200 * rs.random_sample(2)
                    x, y = rs.random_sample(2) * 300
                    bx = np.array([x, y, x + w, y + h], F32)
                    cl = rs.randint(1 if name == "m_missing_cls" else 0, C)  # class 0 is never detected there
                if name == "m_missing_cls" and cl == 0:
                    cl = 1
                dets.append((all_scores[k], bx, cl))
                k += 1
            dets.sort(key=lambda d: -d[0])
            for i, (s, bx, cl) in enumerate(dets):
                scores[b, i], boxes[b, i], classes[b, i] = s, bx, cl
        if name == "m_small":  # exact duplicates of one box: equal IoU on two same-class targets -> the first wins
            targets[0, 1] = targets[0, 0]
            boxes[0, 0] = targets[0, 0, :4]
            classes[0, 0] = targets[0, 0, 4]
            boxes[0, 1] = targets[0, 0, :4]
            classes[0, 1] = targets[0, 0, 4]
        out.append(dict(scores=scores, boxes=boxes, classes=classes, targets=targets))
    return dict(batches=out, C=C, conf_thr=conf_thr, iou_thr=iou_thr)


def loss_inputs():
    """Seeded inputs of the criteria (core/criterion.py): logits / one-hot target [B,A,C,H,W], depth [B,A,1,H,W]
    with positives, negatives and ignored anchors, predicted / target deltas [B,A,4,H,W]."""
    rs = np.random.RandomState(81)
    B, A, C, H, W = 3, 4, 5, 6, 7
    logits = (rs.standard_normal((B, A, C, H, W)) * 2.5).astype(F32)
    depth = rs.choice([-1.0, 0.0, 0.0, 0.0, 1.0, 2.0, 5.0], size=(B, A, 1, H, W)).astype(F32)
    depth[2] = np.where(depth[2] > 0, 0, depth[2])  # an image without positives
    label = rs.randint(0, C, (B, A, 1, H, W))
    target = ((np.arange(C).reshape(1, 1, C, 1, 1) == label) & (depth > 0)).astype(F32)
    pred = (rs.standard_normal((B, A, 4, H, W)) * 0.4).astype(F32)
    tgt = (rs.standard_normal((B, A, 4, H, W)) * 0.4).astype(F32)
    tgt[0, 0] = pred[0, 0]          # identical boxes
    tgt[0, 1, :2] = pred[0, 1, :2] + 5.0  # disjoint boxes
    return dict(logits=logits, target=target, depth=depth, pred=pred, tgt=tgt)


# ----------------------------------------------------------------------------- whole detector networks
# (a13-a16: SSD / SSDFPN / SSDBiFPN of the reference around its own backbones, make_golden.gen_nets)
# name: (seed, head class, backbone factory | "stub", FEATURE_LAYER, anchors per location, classes, (B, H, W))
NET_CASES = OrderedDict(
    [
        ("ssd_mnv2", (91, "SSD", "MobileNetV2", [[5, 7, "Conv:S", "Conv:S"], [96, 320, 256, 128]], 6, 5, (2, 160, 160))),
        ("fpn_r18", (92, "SSDFPN", "ResNet18", [[3, 4, 5, "Conv:S", "Conv:S"], [128, 256, 512, 512, 256]], 9, 4,
                     (2, 128, 160))),
        ("fpn_r50", (93, "SSDFPN", "ResNet50", [[3, 4, 5, "Conv:S", "Conv:S"], [512, 1024, 2048, 2048, 256]], 9, 3,
                     (2, 128, 128))),
        ("bifpn_regx008", (94, "SSDBiFPN", "RegNetX008", [[2, 3, 4, "Conv:S", "Conv:S"], [128, 288, 672, 672, 256]],
                           9, 3, (2, 128, 160))),
        ("bifpn_regx002_x2", (95, "SSDBiFPN", "RegNetX002", [[2, 3, 4, "Conv:S"], [56, 152, 368, 368], 2], 9, 3,
                              (2, 64, 96))),
        # a stub backbone that returns seeded feature maps: the neck / towers / heads alone, on awkward shapes
        ("ssd_stub", (96, "SSD", "stub", [[0, 1, "Conv:S", "Conv:S"], [64, 136, 128, 64]], 6, 7, (3, 72, 56))),
        ("fpn_stub", (97, "SSDFPN", "stub", [[0, 1, 2, "Conv:S"], [40, 112, 320, 320]], 9, 2, (2, 96, 128))),
        ("bifpn_stub", (98, "SSDBiFPN", "stub", [[0, 1, 2, "Conv:S", "Conv:S"], [48, 96, 200, 200, 256]], 3, 6,
                        (2, 64, 64))),
    ]
)
# stub backbones: feature map l has stride 8 * 2**l and the channel count FEATURE_LAYER names for it


def net_image(name):
    seed, _, _, _, _, _, (B, H, W) = NET_CASES[name]
    return np.random.RandomState(seed).random_sample((B, 3, H, W)).astype(F32)


def stub_features(name):
    seed, _, net, fl, _, _, (B, H, W) = NET_CASES[name]
    assert net == "stub"
    rs = np.random.RandomState(seed + 1000)
    feats = []
    for l, (layer, depth) in enumerate(zip(*fl[:2])):
        if isinstance(layer, int):
            s = 8 << l
            feats.append((rs.standard_normal((B, depth, H // s, W // s)) * 0.7).astype(F32))
    return feats


def seeded_state(spec, seed):
    """Deterministic parameter values for a ``state_dict`` schema ``spec`` = [(key, shape)] (the reference
    model's), independent of key order: conv / linear weights ~ N(0, 1.5/fan_in), biases ~ N(0, 0.1^2), BatchNorm
    weight ~ U(0.5, 1.5), BiFPN fusion weights ~ U(-0.2, 1) (negative ones are cut by the relu of bifpn.py:35-38).
    BatchNorm running statistics are NOT generated here: make_golden.py calibrates them on the reference model
    (one train-mode pass) and stores them in the fixture."""
    keys = [k for k, _ in spec]
    bn = {k[: -len("running_mean")] for k in keys if k.endswith("running_mean")}
    out = OrderedDict()
    for k, shape in spec:
        rs = np.random.RandomState((seed * 7919 + zlib.crc32(k.encode())) & 0x7FFFFFFF)
        shape = tuple(int(s) for s in shape)
        pre = k[: k.rfind(".") + 1]
        if k.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
        elif k.endswith("running_mean"):
            v = np.zeros(shape, F32)
        elif k.endswith("running_var"):
            v = np.ones(shape, F32)
        elif pre in bn and k.endswith("weight"):
            v = rs.uniform(0.5, 1.5, shape).astype(F32)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            v = (rs.standard_normal(shape) * np.sqrt(1.5 / fan_in)).astype(F32)
        elif len(shape) == 2 and (k.endswith("w1") or k.endswith("w2")):
            v = rs.uniform(-0.2, 1.0, shape).astype(F32)
        elif len(shape) == 2:
            v = (rs.standard_normal(shape) * 0.01).astype(F32)
        else:
            v = (rs.standard_normal(shape) * 0.1).astype(F32)
        out[k] = v
    return out
