"""Target assignment fused with the focal (or MultiBoxLoss) / smooth-L1 (or IoU-family) losses of one level -- the HIP replacement for
the per-level body of the reference's ``ModelWithLossBasic.forward`` (pipeline/pipeline_anchor_apex.py:48-66):
``extract_targets`` (modeling/layers/box.py:362-405), ``FocalLoss`` / ``SmoothL1Loss`` (core/criterion.py:74-151),
the depth masks and the three sums.  One launch per level reads the logits once and writes their gradients; the
one-hot / box / depth target tensors and the element-wise loss tensors are never materialised."""
import ctypes

import torch

from ssds import _native as N
from ssds.modeling.layers.box import _anchor_array


class _MatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conf, loc, targets, anc, classes, stride, by_scale, thr_a, thr_b, radius, alpha, gamma, beta,
                loc_loss, negpos_ratio):
        N.require_device(conf, "match_loss")
        if loc.dtype != conf.dtype:
            loc = loc.to(conf.dtype)
        conf_c, loc_c = conf.contiguous(), loc.contiguous()
        B, _, H, W = conf_c.shape
        A = int(anc.shape[0])
        if conf_c.shape[1] != A * classes or tuple(loc_c.shape) != (B, A * 4, H, W):
            raise ValueError("match_loss: conf %s / loc %s do not fit %d anchors x %d classes" % (
                tuple(conf.shape), tuple(loc.shape), A, classes))
        t = targets.contiguous().float()
        G = int(t.shape[1])
        dev = conf_c.device
        d_conf, d_loc = torch.empty_like(conf_c), torch.empty_like(loc_c)
        sums = torch.empty(3, device=dev, dtype=torch.float32)
        multibox = negpos_ratio is not None
        ws_query = N.lib.ssdk_match_multibox_loss_workspace_bytes if multibox else N.lib.ssdk_match_loss_workspace_bytes
        ws_bytes = int(ws_query(B, A, H, W))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            if multibox:  # MultiBoxLoss: match + positives, per-image select of the hardest negatives, their terms
                rc = N.lib.ssdk_match_multibox_loss(
                    t.data_ptr(), B, G, anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), A, int(classes), H, W,
                    int(stride), int(by_scale), float(thr_a), float(thr_b), float(radius), conf_c.data_ptr(),
                    loc_c.data_ptr(), N.dtype_code(conf_c), float(negpos_ratio), float(beta), int(loc_loss),
                    d_conf.data_ptr(), d_loc.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws_bytes, N.stream_ptr(dev))
            else:
                rc = N.lib.ssdk_match_loss(
                    t.data_ptr(), B, G, anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), A, int(classes), H, W,
                    int(stride), int(by_scale), float(thr_a), float(thr_b), float(radius), conf_c.data_ptr(),
                    loc_c.data_ptr(), N.dtype_code(conf_c), float(alpha), float(gamma), float(beta), int(loc_loss),
                    d_conf.data_ptr(), d_loc.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws_bytes, N.stream_ptr(dev))
        N.check(rc, "match_multibox_loss" if multibox else "match_loss")
        ctx.save_for_backward(d_conf, d_loc)
        ctx.mark_non_differentiable(sums)
        cls_sum, loc_sum = sums[0].clone(), sums[1].clone()
        return cls_sum, loc_sum, sums

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_cls, g_loc, _g_sums):
        d_conf, d_loc = ctx.saved_tensors
        # out of place: the saved gradients stay what the kernel wrote, so backward(retain_graph=True) may run again
        gc = d_conf * g_cls if g_cls is not None else None  # 0-dim fp32 scale, fp32 arithmetic, dtype kept
        gl = d_loc * g_loc if g_loc is not None else None
        return (gc, gl) + (None,) * 13


LOC_LOSS = {"smoothl1": 0, "iou": 1, "giou": 2, "diou": 3, "ciou": 4}


def match_loss(conf, loc, targets, anchors, classes, stride, match, center_sampling_radius=0, alpha=0.25, gamma=2.0,
               beta=0.11, loc_loss="smoothl1", negpos_ratio=None):
    """conf [B, A*C, H, W] logits and loc [B, A*4, H, W] of one level, targets [B, G, 5] (x, y, w, h, label; -1 =
    padding), ``anchors`` / ``match`` as for ``box.extract_targets``; ``loc_loss``: ``smoothl1`` (``beta``) or the
    ``IOULoss`` types ``iou | giou | diou | ciou``.  ``negpos_ratio`` (not None) replaces the focal class term by
    ``MultiBoxLoss(negpos_ratio)`` (criterion.py:43-71: positives + the hardest ratio x #positives negatives of each image
    of this level; ``alpha`` / ``gamma`` unused).  Returns (cls_sum, loc_sum, fg): the masked
    focal sum and smooth-L1 sum (differentiable w.r.t. conf / loc) and the un-clamped foreground count, fp32
    scalars on the device."""
    by_scale = isinstance(match[0], list)
    if not by_scale and not isinstance(match[0], float):
        raise ValueError("unvalidate match param")
    anc = _anchor_array(anchors, stride)
    if by_scale:
        thr_a, thr_b = match[list(anchors).index(stride)]
    else:
        thr_a, thr_b = match[0], match[1]
    cls_sum, loc_sum, sums = _MatchLoss.apply(conf, loc, targets, anc, int(classes), int(stride), by_scale, thr_a,
                                              thr_b, float(center_sampling_radius), alpha, gamma, beta,
                                              LOC_LOSS[loc_loss], negpos_ratio)
    return cls_sum, loc_sum, sums[2]
