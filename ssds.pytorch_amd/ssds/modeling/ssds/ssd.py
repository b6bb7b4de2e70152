"""SSD multibox detector head -- same constructor, ``add_extras`` factory, module names (``extras``,
``loc``, ``conf``) and forward contract as the reference's ``ssds/modeling/ssds/ssd.py``:

    forward(x[B,3,H,W]) -> (tuple loc_l [B, A*4, h_l, w_l], tuple conf_l [B, A*C, h_l, w_l])

conf = logits in training mode, sigmoid probabilities in eval mode (reference ssd.py:72-73).

MI355X execution: in eval mode on a HIP device the two bare 3x3 head convs of a level
(reference ssd.py:100-103) run on the MFMA implicit-GEMM kernel with the bias and the sigmoid fused into
the epilogue (``ssdk_conv_bn_act``); the extras (1x1 + 3x3/s2 Conv-BN-ReLU pairs) run on the same kernel
with folded BatchNorm.  Training mode is ordinary autograd; with ``headconv.use_head_pairs`` (ssds/utils/train_ddp.py) the
forward of the twelve head convolutions runs on the same kernels, one split-output launch per level."""
import torch.nn as nn

import torch

from ssds.modeling.layers import fused_conv as FC
from ssds.modeling.layers import headconv as HC
from ssds.modeling.layers.layers_parser import parse_feature_layer
from ssds.modeling.layers.planner import PlanUnsupported, build_ssd_plan

from .ssdsbase import SSDSBase, drop_child_packs


class SSD(SSDSBase):
    r"""SSD: Single Shot MultiBox Detector (https://arxiv.org/abs/1512.02325).

    Args:
        backbone: feature extractor returning a list of feature maps
        extras: extra layers appended after the backbone
        head: (loc conv list, conf conv list), one pair per level
        num_classes: number of classes
    """

    def __init__(self, backbone, extras, head, num_classes):
        super(SSD, self).__init__(backbone, num_classes)
        self.extras = nn.ModuleList(extras)
        self.loc = nn.ModuleList(head[0])
        self.conf = nn.ModuleList(head[1])
        self.initialize()

    def initialize(self):
        self.backbone.initialize()
        self.extras.apply(self.initialize_extra)
        self.loc.apply(self.initialize_head)
        self.conf.apply(self.initialize_head)
        for c in self.conf:
            c.apply(self.initialize_prior)

    # ---- MI355X inference engine: the whole eval forward as one recorded plan ---------------------------
    def invalidate_plans(self):
        """Drop recorded plans / folded weights (called automatically by train(), load_state_dict(), .to());
        call it yourself after editing parameters in place."""
        self._plans = {}
        self._head_packs = None
        drop_child_packs(self)

    def train(self, mode=True):
        self.invalidate_plans()
        return super(SSD, self).train(mode)

    def _apply(self, fn, *a, **kw):
        self.invalidate_plans()
        return super(SSD, self)._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate_plans()
        return super(SSD, self).load_state_dict(*a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        # also reached when an ANCESTOR's load_state_dict() runs (model-with-loss wrappers, DDP)
        self.invalidate_plans()
        return super(SSD, self)._load_from_state_dict(*a, **kw)

    def _native_ok(self, x):
        return (not self.training and FC.fused_enabled() and x.is_cuda
                and x.dtype in (torch.bfloat16, torch.float16))

    def _plan(self, x):
        plans = self.__dict__.setdefault("_plans", {})
        key = (tuple(x.shape), x.dtype, x.device.index)
        if key not in plans:
            try:
                with torch.no_grad():
                    plans[key] = build_ssd_plan(self, x)
            except PlanUnsupported as e:
                plans[key] = str(e)
        return plans[key]

    def _heads_native(self, features):
        packs = self.__dict__.get("_head_packs")
        if packs is None or packs[0] != features[0].dtype:
            packs = (features[0].dtype, [FC.pack_heads(l, c, features[0].dtype) for l, c in zip(self.loc, self.conf)])
            self.__dict__["_head_packs"] = packs
        loc, conf = [], []
        for f, l, pk in zip(features, self.loc, packs[1]):
            y, y2 = FC.conv_native(f, pk, act="none", nchw_out=True, split=l.out_channels, act2="sigmoid")
            loc.append(y)
            conf.append(y2)
        return tuple(loc), tuple(conf)

    def forward(self, x):
        if self._native_ok(x):
            plan = self._plan(x)
            if not isinstance(plan, str):
                return plan.run(x)  # backbone + extras + heads: one C call, one launch per fused layer
        loc, conf = [], []
        features = self.backbone(x)
        for v in self.extras:  # each extra consumes the previous last feature (reference ssd.py:63-65)
            features.append(v(features[-1]))
        if self._native_ok(x) and all(FC.conv_kind(m) == "dense" for m in list(self.loc) + list(self.conf)):
            return self._heads_native(features)  # backbone without a planner: fused heads only
        pair = self.training and self.__dict__.get("_ssdk_head_pair", False)
        for f, l, c in zip(features, self.loc, self.conf):
            if pair and HC.supported(f, l, c):  # training step on a HIP device: loc | conf of the level as one kernel-backed layer
                y, y2 = HC.head_pair(f, l, c)
                loc.append(y)
                conf.append(y2)
                continue
            loc.append(l(f))
            conf.append(c(f))
        if not self.training:
            conf = [c.sigmoid() for c in conf]
        return tuple(loc), tuple(conf)

    @staticmethod
    def add_extras(feature_layer, mbox, num_classes):
        """Declare extras + loc/conf heads from cfg.MODEL.FEATURE_LAYER (reference ssd.py:77-104):
        ints name backbone outputs, strings name extra layers; every level gets a 3x3 loc conv
        (A*4 channels) and a 3x3 conf conv (A*num_classes channels), both with bias."""
        nets_outputs, extra_layers, loc_layers, conf_layers = [], [], [], []
        in_channels = None
        for layer, depth, box in zip(feature_layer[0], feature_layer[1], mbox):
            if isinstance(layer, int):
                nets_outputs.append(layer)
            else:
                extra_layers += parse_feature_layer(layer, in_channels, depth)
            in_channels = depth
            loc_layers += [nn.Conv2d(in_channels, box * 4, kernel_size=3, padding=1)]
            conf_layers += [nn.Conv2d(in_channels, box * num_classes, kernel_size=3, padding=1)]
        return nets_outputs, extra_layers, (loc_layers, conf_layers)
