"""Losses on the DDP training path (reference ``ssds/core/criterion.py``): ``FocalLoss`` (:74-108),
``SmoothL1Loss`` (:111-151), ``MultiBoxLoss`` (:8-71) and the IoU family (:154-293), element-wise, un-reduced -- the caller masks by ``depth`` and normalises by
the foreground count (pipeline_anchor_apex.py:55-71).  These modules are the plain torch definition
(used on CPU by the gloo tests and as the parity reference); on a HIP device ``ModelWithLossBasic`` replaces
target assignment + both losses + masks + sums by one launch per level (``ssds/core/fused_loss.py``)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalLoss(nn.Module):
    r"""FL(p_t) = -alpha_t (1 - p_t)^gamma log(p_t) on logits (https://arxiv.org/abs/1708.02002).

    Args:
        alpha: weight of the positive class, default 0.25
        gamma: focusing parameter, default 2
    """

    def __init__(self, alpha=0.25, gamma=2, **kwargs):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma

    def forward(self, pred_logits, target, depth=None):
        pred = pred_logits.sigmoid()
        ce = F.binary_cross_entropy_with_logits(pred_logits, target, reduction="none")
        alpha = target * self.alpha + (1.0 - target) * (1.0 - self.alpha)
        pt = torch.where(target == 1, pred, 1 - pred)
        return alpha * (1.0 - pt) ** self.gamma * ce


class SmoothL1Loss(nn.Module):
    r"""Huber-style loss with threshold ``beta``: 0.5 x^2 / beta below beta, x - 0.5 beta above
    (reference criterion.py:140-151; default beta 0.11)."""

    def __init__(self, beta=0.11, **kwargs):
        super().__init__()
        self.beta = beta

    def forward(self, pred, target):
        x = (pred - target).abs()
        l1 = x - 0.5 * self.beta
        l2 = 0.5 * x ** 2 / self.beta
        return torch.where(x >= self.beta, l1, l2)


class MultiBoxLoss(nn.Module):
    r"""Classification part of the SSD MultiBox loss: sigmoid cross entropy on the positives plus the hardest
    ``negpos_ratio`` x #positives negatives of each image (reference criterion.py:8-71).  Shapes as in the pipeline:
    logits / target [B, A, C, H, W], depth [B, A, 1, H, W] (>0 positive, 0 negative, <0 ignored).  Un-reduced; the
    caller masks by ``depth >= 0`` and sums."""

    def __init__(self, negpos_ratio=3, **kwargs):
        super().__init__()
        self.negpos_ratio = negpos_ratio

    def forward(self, pred_logits, target, depth):
        ce = F.binary_cross_entropy_with_logits(pred_logits, target, reduction="none")
        B = ce.shape[0]
        # hardness of an anchor = its largest per-class term; positives and ignored anchors never rank
        hard = ce.max(2)[0].view(B, -1).clone()
        flat_depth = depth.view(B, -1)
        hard[flat_depth != 0] = 0
        order = hard.sort(1, descending=True)[1]
        rank = order.sort(1)[1]  # rank[b, i] = position of anchor i in the descending order
        num_pos = (flat_depth > 0).sum(1, keepdim=True)
        num_neg = torch.clamp(self.negpos_ratio * num_pos, max=flat_depth.shape[1] - 1)
        mined = (rank < num_neg).view_as(depth)
        return ce * ((depth > 0) | mined).expand_as(ce)


class IOULoss(nn.Module):
    r"""1 - IoU-family overlap between predicted and target boxes given as deltas (cx, cy, log w, log h) in anchor
    units, [B, A, 4, H, W] -> [B, A, 1, H, W] (reference criterion.py:154-239).  ``loss_type``: ``iou``, ``giou``
    (enclosing-box penalty), ``diou`` (centre distance over enclosing diagonal), ``ciou`` (diou + aspect-ratio term
    whose weight alpha is a constant for the gradient)."""

    EPS = 1e-7

    def __init__(self, loss_type="iou"):
        super().__init__()
        if loss_type not in ("iou", "giou", "diou", "ciou"):
            raise ValueError("unknown IoU loss type {}".format(loss_type))
        self.loss_type = loss_type

    @staticmethod
    def _corners(d):
        wh = torch.exp(d[:, :, 2:])
        return d[:, :, :2] - 0.5 * wh, d[:, :, :2] + 0.5 * wh, wh

    @staticmethod
    def _area(lt, rb):
        return torch.prod(rb - lt, dim=2) * (lt < rb).all(dim=2)

    def forward(self, pred, target):
        p_lt, p_rb, p_wh = self._corners(pred)
        t_lt, t_rb, t_wh = self._corners(target)
        inter = self._area(torch.max(p_lt, t_lt), torch.min(p_rb, t_rb))
        union = torch.prod(p_wh, dim=2) + torch.prod(t_wh, dim=2) - inter
        iou = (inter + self.EPS) / (union + self.EPS)
        if self.loss_type == "iou":
            return 1 - iou.clamp(0, 1.0).unsqueeze(2)
        o_lt, o_rb = torch.min(p_lt, t_lt), torch.max(p_rb, t_rb)
        if self.loss_type == "giou":
            hull = self._area(o_lt, o_rb) + self.EPS
            return 1 - (iou - (hull - union) / hull).clamp(-1.0, 1.0).unsqueeze(2)
        centre = ((pred[:, :, :2] - target[:, :, :2]) ** 2).sum(dim=2)
        diag = ((o_rb - o_lt) ** 2).sum(dim=2) + self.EPS
        if self.loss_type == "diou":
            return 1 - (iou - centre / diag).clamp(-1.0, 1.0).unsqueeze(2)
        v = (4 / math.pi ** 2) * (torch.atan(t_wh[:, :, 0] / t_wh[:, :, 1]) - torch.atan(p_wh[:, :, 0] / p_wh[:, :, 1])) ** 2
        with torch.no_grad():
            alpha = v / (1 - iou + v)
        return 1 - (iou - (centre / diag + alpha * v)).clamp(-1.0, 1.0).unsqueeze(2)


def GIOULoss():
    return IOULoss("giou")


def DIOULoss():
    return IOULoss("diou")


def CIOULoss():
    return IOULoss("ciou")
