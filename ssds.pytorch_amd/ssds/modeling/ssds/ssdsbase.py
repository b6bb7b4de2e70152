"""Common base of the detector heads (reference ``ssds/modeling/ssds/ssdsbase.py:6-31``): weight
initialisers for extras (Xavier), heads (N(0, 0.01)) and the class-prior bias (pi = 0.01, so that an
untrained conf head outputs sigmoid = 0.01, exactly at the default SCORE_THRESHOLD)."""
import math

import torch
import torch.nn as nn


class SSDSBase(nn.Module):
    def __init__(self, backbone, num_classes):
        super(SSDSBase, self).__init__()
        self.backbone = backbone
        self.num_classes = num_classes

    def initialize_prior(self, layer):
        pi = 0.01
        b = -math.log((1 - pi) / pi)
        nn.init.constant_(layer.bias, b)
        nn.init.normal_(layer.weight, std=0.01)

    def initialize_head(self, layer):
        if isinstance(layer, nn.Conv2d):
            nn.init.normal_(layer.weight, std=0.01)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, val=0)

    def initialize_extra(self, layer):
        if isinstance(layer, nn.Conv2d):
            nn.init.xavier_uniform_(layer.weight)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, val=0)


def drop_child_packs(root):
    """Forget the folded weights cached by the fused blocks below ``root`` (FusedSequentialMixin._ssdk_packs,
    SharedHead._final_pack): the per-layer fallback path must never run on weights older than the parameters."""
    for m in root.modules():
        if "_ssdk_packs" in m.__dict__:
            m.__dict__["_ssdk_packs"] = None
        if "_final_pack" in m.__dict__:
            m.__dict__["_final_pack"] = None


class NeckPlanMixin(object):
    """Eval forward of a detector whose backbone runs on PyTorch-ROCm and whose neck + towers run as one
    recorded plan (``ssds/modeling/layers/planner.py``) on the backbone's feature maps.  ``_build_neck_plan`` is
    provided by the subclass; plans are cached per feature shapes and dropped by train() / .to() /
    load_state_dict()."""

    def invalidate_plans(self):
        self.__dict__["_neck_plans"] = {}
        drop_child_packs(self)

    def train(self, mode=True):
        self.invalidate_plans()
        return super(NeckPlanMixin, self).train(mode)

    def _apply(self, fn, *a, **kw):
        self.invalidate_plans()
        return super(NeckPlanMixin, self)._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate_plans()
        return super(NeckPlanMixin, self).load_state_dict(*a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        # also reached when an ANCESTOR's load_state_dict() runs (model-with-loss wrappers, DDP): see
        # FusedSequentialMixin._load_from_state_dict
        self.invalidate_plans()
        return super(NeckPlanMixin, self)._load_from_state_dict(*a, **kw)

    def _full_native(self, x):
        """Image -> (loc, conf) with backbone, neck and towers as ONE plan, or None (backbone without a planner,
        training mode, CPU tensors ...)."""
        from ssds.modeling.layers import fused_conv as FC
        from ssds.modeling.layers.planner import PlanUnsupported

        if self.training or not FC.fused_enabled() or not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16):
            return None
        plans = self.__dict__.setdefault("_neck_plans", {})
        key = ("image", tuple(x.shape), x.dtype, x.device.index)
        if key not in plans:
            try:
                with torch.no_grad():
                    plans[key] = self._build_neck_plan(None, image=x)
            except PlanUnsupported as e:
                plans[key] = str(e)
        plan = plans[key]
        if isinstance(plan, str):
            return None
        return plan.run(x)

    def _neck_native(self, features):
        """(loc, conf) through the plan, or None when the plan path does not apply."""
        from ssds.modeling.layers import fused_conv as FC
        from ssds.modeling.layers.planner import PlanUnsupported

        f0 = features[0]
        if self.training or not FC.fused_enabled() or not f0.is_cuda or f0.dtype not in (torch.bfloat16, torch.float16):
            return None
        plans = self.__dict__.setdefault("_neck_plans", {})
        key = (tuple(tuple(f.shape) for f in features), f0.dtype, f0.device.index)
        if key not in plans:
            try:
                with torch.no_grad():
                    plans[key] = self._build_neck_plan(features)
            except PlanUnsupported as e:
                plans[key] = str(e)
        plan = plans[key]
        if isinstance(plan, str):
            return None
        return plan.run(*features)
