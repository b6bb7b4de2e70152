"""Losses on the DDP training path (reference ``ssds/core/criterion.py``): ``FocalLoss`` (:74-108) and
``SmoothL1Loss`` (:111-151), element-wise, un-reduced -- the caller masks by ``depth`` and normalises by
the foreground count (pipeline_anchor_apex.py:55-71).  These modules are the plain torch definition
(used on CPU by the gloo tests and as the parity reference); on a HIP device ``ModelWithLossBasic`` replaces
target assignment + both losses + masks + sums by one launch per level (``ssds/core/fused_loss.py``)."""
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
