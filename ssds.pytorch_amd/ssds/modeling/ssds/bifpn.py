"""SSDBiFPN (EfficientDet-style weighted bi-directional FPN + shared towers) -- constructor,
``add_extras`` factory, module/parameter names (``transforms``, ``extras``, ``stack_bifpn``, ``loc``,
``conf``; ``w1``, ``w2``, ``top-down-i``, ``bottom-up-i``) and forward contract of the reference's
``ssds/modeling/ssds/bifpn.py`` (BiFPNModule :10-63, SSDBiFPN :66-196)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ssds.modeling.layers.basic_layers import ConvBNReLU

from .fpn import SharedHead
from .ssdsbase import NeckPlanMixin, SSDSBase


class BiFPNModule(nn.Module):
    """One BiFPN layer over ``levels`` maps of ``channels`` channels with fast-normalised fusion weights
    (relu(w) / (sum relu(w) + 1e-6), reference bifpn.py:35-38)."""

    def __init__(self, channels, levels, init=0.5, block=ConvBNReLU):
        super(BiFPNModule, self).__init__()
        self.levels = levels
        self.w1 = nn.Parameter(torch.Tensor(2, levels).fill_(init))
        self.w2 = nn.Parameter(torch.Tensor(3, levels - 2).fill_(init))
        for i in range(levels - 1, 0, -1):
            self.add_module("top-down-{}".format(i - 1), block(channels, channels))
        for i in range(0, levels - 1, 1):
            self.add_module("bottom-up-{}".format(i + 1), block(channels, channels))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, val=0)

    def forward(self, xx):
        assert len(xx) == self.levels
        n = self.levels
        w1 = F.relu(self.w1)
        w1 = w1 / (torch.sum(w1, dim=0) + 1e-6)
        w2 = F.relu(self.w2)
        w2 = w2 / (torch.sum(w2, dim=0) + 1e-6)
        w1 = w1.to(xx[0].dtype)
        w2 = w2.to(xx[0].dtype)
        skips = [None] + [x for x in xx[1:-1]] + [None]
        for i in range(n - 1, 0, -1):  # top-down (reference bifpn.py:41-46)
            fused = w1[0, i - 1] * xx[i - 1] + w1[1, i - 1] * F.interpolate(xx[i], scale_factor=2, mode="nearest")
            xx[i - 1] = getattr(self, "top-down-{}".format(i - 1))(fused)
        for i in range(0, n - 2, 1):  # bottom-up with skip (reference bifpn.py:49-55)
            fused = w2[0, i] * xx[i + 1] + w2[1, i] * F.max_pool2d(xx[i], kernel_size=2) + w2[2, i] * skips[i + 1]
            xx[i + 1] = getattr(self, "bottom-up-{}".format(i + 1))(fused)
        fused = w1[0, n - 1] * xx[n - 1] + w1[1, n - 1] * F.max_pool2d(xx[n - 2], kernel_size=2)
        xx[n - 1] = getattr(self, "bottom-up-{}".format(n - 1))(fused)  # reference bifpn.py:57-62
        return xx


class SSDBiFPN(NeckPlanMixin, SSDSBase):
    """EfficientDet (https://arxiv.org/abs/1911.09070) head with the reference's ConvBNReLU blocks."""

    def __init__(self, backbone, extras, head, num_classes):
        super(SSDBiFPN, self).__init__(backbone, num_classes)
        self.transforms = nn.ModuleList(extras[0])
        self.extras = nn.ModuleList(extras[1])
        self.stack_bifpn = extras[2]
        self.loc = head[0]
        self.conf = head[1]
        self.initialize()

    def initialize(self):
        self.backbone.initialize()
        self.transforms.apply(self.initialize_extra)
        self.extras.apply(self.initialize_extra)
        self.loc.apply(self.initialize_head)
        self.conf.apply(self.initialize_head)
        self.conf[-1].apply(self.initialize_prior)

    def _build_neck_plan(self, features, image=None):
        from ssds.modeling.layers.planner import build_bifpn_plan

        return build_bifpn_plan(self, features, image=image)

    def forward(self, x):
        out = self._full_native(x)  # planned backbone (MobileNet / ResNet): image -> heads is one plan
        if out is not None:
            return out
        loc, conf = [], []
        features = self.backbone(x)
        out = self._neck_native(features)  # eval on a HIP device: transforms, BiFPN layers, extras, towers = one plan
        if out is not None:
            return out
        x = features[-1]
        n = len(features)
        features = [self.transforms[i](features[i]) for i in range(n)]
        features = self.stack_bifpn(features)
        xx = None
        for i, v in enumerate(self.extras):  # reference bifpn.py:131-138
            if i < n:
                xx = v(features[i])
            elif i == n:
                xx = v(x)
            else:
                xx = v(xx)
            loc.append(self.loc(xx, "none"))
            conf.append(self.conf(xx, "none" if self.training else "sigmoid"))
        return tuple(loc), tuple(conf)

    @staticmethod
    def add_extras(feature_layer, mbox, num_classes):
        """As SSDFPN.add_extras plus ``feature_layer[2]`` stacked BiFPN layers (default 1)
        (reference bifpn.py:144-196)."""
        nets_outputs, transform_layers, extra_layers = [], [], []
        if not all(mbox[i] == mbox[i + 1] for i in range(len(mbox) - 1)):
            raise ValueError("For SSDFPN module, the number of box have to be same in every layer")
        loc_layers = SharedHead(mbox[0] * 4)
        conf_layers = SharedHead(mbox[0] * num_classes)
        for layer, depth in zip(feature_layer[0], feature_layer[1]):
            if isinstance(layer, int):
                nets_outputs.append(layer)
                transform_layers += [nn.Conv2d(depth, 256, 1)]
                extra_layers += [ConvBNReLU(256, 256, 3)]
            elif layer == "Conv:S":
                extra_layers += [ConvBNReLU(depth, 256, 3, stride=2)]
            else:
                raise ValueError(layer + " does not support by SSDFPN")
        num_stack = 1 if len(feature_layer) == 2 else feature_layer[2]
        fpn = nn.Sequential(*[BiFPNModule(256, len(transform_layers)) for _ in range(num_stack)])
        return nets_outputs, (transform_layers, extra_layers, fpn), (loc_layers, conf_layers)
