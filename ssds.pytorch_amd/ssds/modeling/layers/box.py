"""Box math of the detection hot path -- same names/signatures as the reference's
``ssds/modeling/layers/box.py``, executed by hand-written gfx950 kernels through the C-ABI
(``include/ssdk.h``).  No CPU fallback: tensors must live on a HIP device.

Reference lines each function replaces are cited in the docstrings.
"""
import contextlib
import ctypes

import numpy as np
import torch

from ssds import _native as N

INF = 100000


def configure_ratio_scale(num_featmaps, ratios, scales):
    """Normalise cfg.ASPECT_RATIOS / cfg.SIZES to per-level lists (reference box.py:8-43; host logic,
    same errors)."""
    if len(scales) != num_featmaps:
        raise ValueError(
            "cfg.SIZES is not correct,"
            "the len of cfg.SIZES should equal to num layers({}) or 2, but it is {}".format(
                num_featmaps, len(scales)
            )
        )
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
    """Anchor ltrb boxes for one stride (reference box.py:46-58) -> CPU FloatTensor [A,4].
    Computed by ``ssdk_generate_anchors`` (host C, bit-exact with the reference's fp32 tensor ops)."""
    nr, ns = len(ratio_vals), len(scales_vals)
    r = (ctypes.c_float * nr)(*[float(v) for v in ratio_vals])
    s = (ctypes.c_float * ns)(*[float(v) for v in scales_vals])
    out = (ctypes.c_float * (nr * ns * 4))()
    N.check(N.lib.ssdk_generate_anchors(int(stride), r, nr, s, ns, out), "generate_anchors")
    return torch.from_numpy(np.frombuffer(out, dtype=np.float32).reshape(nr * ns, 4).copy())


def box2delta(boxes, anchors):
    """Encode boxes against anchors (reference box.py:61-71); element-wise torch ops on any device."""
    anchors_wh = anchors[:, 2:] - anchors[:, :2] + 1
    anchors_ctr = anchors[:, :2] + 0.5 * anchors_wh
    boxes_wh = boxes[:, 2:] - boxes[:, :2] + 1
    boxes_ctr = boxes[:, :2] + 0.5 * boxes_wh
    return torch.cat([(boxes_ctr - anchors_ctr) / anchors_wh, torch.log(boxes_wh / anchors_wh)], 1)


def delta2box(deltas, anchors, size, stride):
    """Decode deltas against anchors (reference box.py:74-87); element-wise torch ops on any device."""
    anchors_wh = anchors[:, 2:] - anchors[:, :2] + 1
    ctr = anchors[:, :2] + 0.5 * anchors_wh
    pred_ctr = deltas[:, :2] * anchors_wh + ctr
    pred_wh = torch.exp(deltas[:, 2:]) * anchors_wh
    m = torch.zeros([2], device=deltas.device, dtype=deltas.dtype)
    M = torch.tensor([size], device=deltas.device, dtype=deltas.dtype) * stride - 1
    clamp = lambda t: torch.max(m, torch.min(t, M))  # noqa: E731
    return torch.cat([clamp(pred_ctr - 0.5 * pred_wh), clamp(pred_ctr + 0.5 * pred_wh - 1)], 1)


def _anchor_array(anchors, stride=None):
    a = anchors[stride] if isinstance(anchors, dict) else anchors
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def extract_targets(
    targets,
    anchors,
    classes,
    stride,
    size,
    match=[0.5, 0.4],
    center_sampling_radius=0,
    is_centerness=False,
):
    """Snap ground-truth boxes to the anchors of one level for the whole batch in ONE launch
    (reference box.py:362-405 + snap_to_anchors_by_iou box.py:116-226 / snap_to_anchors_by_scale
    box.py:229-359).

    targets [B,G,5] (x, y, w, h, label; label -1 = padding) on a HIP device, anchors =
    OrderedDict{stride: [A,4]}, size = (h, w) of the level.  ``match`` is either the IoU pair
    [match, unmatch] or, for scale-range assignment, one [lower, upper] pair per level (the pair of this
    level is found by the position of ``stride`` among the anchor keys, box.py:389).  Returns fp32
    (cls_target [B,A,C,H,W], box_target [B,A,4,H,W], depth [B,A,1,H,W])."""
    by_scale = isinstance(match[0], list)
    if not by_scale and not isinstance(match[0], float):
        raise ValueError("unvalidate match param")
    if is_centerness:
        # the reference's centerness branch reads an undefined name (box.py:343-344) and cannot run
        raise NotImplementedError("is_centerness targets are not produced by the reference either")
    N.require_device(targets, "extract_targets")
    anc = _anchor_array(anchors, stride)
    A = anc.shape[0]
    t = targets.contiguous().float()
    B, G = int(t.shape[0]), int(t.shape[1])
    H, W = int(size[0]), int(size[1])
    dev = t.device
    cls_t = torch.empty((B, A, classes, H, W), device=dev, dtype=torch.float32)
    box_t = torch.empty((B, A, 4, H, W), device=dev, dtype=torch.float32)
    depth = torch.empty((B, A, 1, H, W), device=dev, dtype=torch.float32)
    anc_p = anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    with torch.cuda.device(dev):
        if by_scale:
            lo, hi = match[list(anchors).index(stride)]  # box.py:389-397
            rc = N.lib.ssdk_match_targets_by_scale(
                t.data_ptr(), B, G, anc_p, A, int(classes), H, W, int(stride), float(lo), float(hi),
                int(center_sampling_radius > 0), cls_t.data_ptr(), box_t.data_ptr(), depth.data_ptr(),
                N.stream_ptr(dev))
        else:
            rc = N.lib.ssdk_match_targets(
                t.data_ptr(), B, G, anc_p, A, int(classes), H, W, int(stride), float(match[0]),
                float(match[1]), float(center_sampling_radius), cls_t.data_ptr(), box_t.data_ptr(),
                depth.data_ptr(), N.stream_ptr(dev))
    N.check(rc, "match_targets")
    return cls_t, box_t, depth


class _TailPipe(object):
    """Tail stream of the decode stage + two alternating workspaces.  A workspace is handed out again only after the
    current stream has been made to wait for the tail work of the call that used it two calls ago."""

    def __init__(self, stream=None):
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.ws = [None, None]
        self.done = [None, None]
        self.i = 0

    def acquire(self, device, nbytes):
        k = self.i & 1
        if self.ws[k] is None or self.ws[k].numel() < nbytes:
            self.ws[k] = torch.empty(int(nbytes) + int(nbytes) // 4, dtype=torch.uint8, device=device)
        if self.done[k] is not None:
            torch.cuda.current_stream(device).wait_event(self.done[k])
        return self.ws[k]

    def release(self):
        k = self.i & 1
        if self.done[k] is None:
            self.done[k] = torch.cuda.Event()
        self.done[k].record(self.stream)
        self.i += 1

    def wait(self):
        """Make the current stream wait for everything enqueued on the tail stream so far."""
        torch.cuda.current_stream().wait_stream(self.stream)


def _heads(all_cls_head, all_box_head):
    N.require_device(all_cls_head, "decode")
    N.require_device(all_box_head, "decode")
    if all_box_head.dtype != all_cls_head.dtype:
        all_box_head = all_box_head.to(all_cls_head.dtype)
    return all_cls_head.contiguous(), all_box_head.contiguous()


def decode(
    all_cls_head,
    all_box_head,
    stride=1,
    threshold=0.05,
    top_n=1000,
    anchors=None,
    rescore=True,
):
    """Box decoding and filtering for one level (reference box.py:408-477): threshold, exact top-n
    (score desc, index asc), delta2box, centre rescoring.  fp32 zero-padded outputs
    (scores [B,top_n], boxes [B,top_n,4], classes [B,top_n])."""
    if anchors is None:
        raise ValueError("decode needs the level's anchors")
    cls, box = _heads(all_cls_head, all_box_head)
    dev = cls.device
    B = int(cls.shape[0])
    dt = N.dtype_code(cls)
    lv = N.make_level(cls, box, stride, anchors)
    top_n = int(top_n)
    scores = torch.empty((B, top_n), device=dev, dtype=torch.float32)
    boxes = torch.empty((B, top_n, 4), device=dev, dtype=torch.float32)
    classes = torch.empty((B, top_n), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        need = N.lib.ssdk_decode_workspace_bytes(ctypes.byref(lv), 1, B, dt, top_n)
        if need == 0:
            N.check(-1, "decode (workspace query)")
        ws = N.workspace(dev, need)
        rc = N.lib.ssdk_decode(ctypes.byref(lv), B, dt, float(threshold), top_n, int(bool(rescore)),
                               scores.data_ptr(), boxes.data_ptr(), classes.data_ptr(),
                               ws.data_ptr(), ws.numel(), N.stream_ptr(dev))
    N.check(rc, "decode")
    return scores, boxes, classes


def nms(all_scores, all_boxes, all_classes, nms=0.5, ndetections=100, using_diou=True):
    """Greedy class-aware (D)IoU non-maximum suppression (reference box.py:480-546).
    fp32 zero-padded outputs ([B,ndetections], [B,ndetections,4], [B,ndetections])."""
    N.require_device(all_scores, "nms")
    s = all_scores.contiguous().float()
    b = all_boxes.contiguous().float()
    c = all_classes.contiguous().float()
    dev = s.device
    B, n = int(s.shape[0]), int(s.shape[1])
    nd = int(ndetections)
    os_ = torch.empty((B, nd), device=dev, dtype=torch.float32)
    ob = torch.empty((B, nd, 4), device=dev, dtype=torch.float32)
    oc = torch.empty((B, nd), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        rc = N.lib.ssdk_nms(s.data_ptr(), b.data_ptr(), c.data_ptr(), B, n, float(nms), nd,
                            int(bool(using_diou)), os_.data_ptr(), ob.data_ptr(), oc.data_ptr(),
                            None, 0, N.stream_ptr(dev))
    N.check(rc, "nms")
    return os_, ob, oc


_default_ctx = {}


def _ctx_for(device):
    """Module-level fallback context per (thread, device) for callers that use ``decode_nms`` without a ``Decoder``."""
    import threading

    key = (threading.get_ident(), device.index)
    c = _default_ctx.get(key)
    if c is None:
        c = _default_ctx[key] = N.Context(device)
    return c


def decode_nms(loc, conf, anchors, threshold, top_n_per_level, rescore, nms_threshold, ndetections,
               using_diou, return_mid=False, tail=None, ctx=None):
    """Decoder.__call__ of the reference (decoder.py:25-49) as one C-ABI call: decode of every level
    (one scan launch + one per-level launch) and NMS, nothing returns to the host in between.

    ``tail`` (a ``_TailPipe``, see ``Decoder.enable_tail_stream``): the latency-bound end of the stage (level
    merge/sort/decode + NMS) runs on the pipe's stream, so it overlaps whatever the caller enqueues next on the
    current stream; the outputs are then complete on that stream (``Decoder.wait()``)."""
    if len(loc) != len(conf) or len(loc) != len(anchors):
        raise ValueError("loc / conf / anchors disagree on the number of levels")
    L = len(loc)
    if L > N.MAX_LEVELS:
        raise N.SsdkError("at most {} levels".format(N.MAX_LEVELS))
    heads = [_heads(c, l) for l, c in zip(loc, conf)]
    dev = heads[0][0].device
    B = int(heads[0][0].shape[0])
    dt = N.dtype_code(heads[0][0])
    levels = (N.Level * L)()
    for i, ((c, l), (stride, anchor)) in enumerate(zip(heads, anchors.items())):
        if N.dtype_code(c) != dt:
            raise N.SsdkError("all levels must share one dtype")
        levels[i] = N.make_level(c, l, stride, anchor)
    K, nd = int(top_n_per_level), int(ndetections)
    # with a tail stream the outputs are written there: allocate them on that stream so that the caching allocator
    # orders their re-use after the tail work
    with torch.cuda.stream(tail.stream) if tail is not None else contextlib.nullcontext():
        os_ = torch.empty((B, nd), device=dev, dtype=torch.float32)
        ob = torch.empty((B, nd, 4), device=dev, dtype=torch.float32)
        oc = torch.empty((B, nd), device=dev, dtype=torch.float32)
        mid = (None, None, None)
        if return_mid:
            mid = (torch.empty((B, L * K), device=dev, dtype=torch.float32),
                   torch.empty((B, L * K, 4), device=dev, dtype=torch.float32),
                   torch.empty((B, L * K), device=dev, dtype=torch.float32))
    if ctx is None:
        ctx = _ctx_for(dev)
    if ctx.device != dev:
        raise N.SsdkError("decode_nms: the context belongs to {}, the heads live on {}".format(ctx.device, dev))
    with torch.cuda.device(dev):
        need = N.lib.ssdk_decode_nms_workspace_bytes(levels, L, B, dt, K, nd)
        if need == 0:
            N.check(-1, "decode_nms (workspace query)")
        if tail is None:
            ws = N.workspace(dev, need + 256)
        else:
            ws = tail.acquire(dev, need + 256)  # the current stream now waits for the tail work that last used it
        ctx.set_tail_stream(tail.stream if tail is not None else None)
        wptr = (ws.data_ptr() + 255) & ~255
        rc = N.lib.ssdk_decode_nms_ctx(
            ctx.ptr, levels, L, B, dt, float(threshold), K, int(bool(rescore)), float(nms_threshold), nd,
            int(bool(using_diou)), os_.data_ptr(), ob.data_ptr(), oc.data_ptr(),
            mid[0].data_ptr() if return_mid else None, mid[1].data_ptr() if return_mid else None,
            mid[2].data_ptr() if return_mid else None, wptr, ws.numel() - (wptr - ws.data_ptr()),
            N.stream_ptr(dev))
        if tail is not None and rc == 0:
            for c, l in heads:  # read by kernels on the tail stream: not to be recycled before those have run
                c.record_stream(tail.stream)
                l.record_stream(tail.stream)
            # the outputs belong to the tail stream's pool but are consumed (after Decoder.wait()) on the caller's
            # stream: their memory must not return to the tail stream while the consumer's kernels still read it
            cur = torch.cuda.current_stream(dev)
            for t in (os_, ob, oc) + tuple(m for m in mid if m is not None):
                t.record_stream(cur)
            tail.release()
    N.check(rc, "decode_nms")
    if return_mid:
        return (os_, ob, oc), mid
    return os_, ob, oc
