"""SSDFPN (RetinaNet-style FPN + shared towers) -- constructor, ``add_extras`` factory, module names
(``transforms``, ``extras``, ``loc``, ``conf``) and forward contract of the reference's
``ssds/modeling/ssds/fpn.py`` (SharedHead :10-18, forward :58-101, add_extras :103-147).

MI355X execution (eval, HIP device): the two shared towers -- 4 x (3x3 conv 256->256 + BN + ReLU) and a
final 3x3 conv to A*4 / A*C channels, weights shared by every level -- are the dominant dense
contraction of BASELINE config 3 (110 GFLOP/img); each conv of a tower is one fused MFMA launch per
level (BN folded, ReLU / sigmoid in the epilogue)."""
import torch.nn as nn
import torch.nn.functional as F

from ssds.modeling.layers.basic_layers import ConvBNReLU
from ssds.modeling.layers.fused_conv import FusedSequentialMixin

from .ssdsbase import NeckPlanMixin, SSDSBase


class SharedHead(FusedSequentialMixin, nn.Sequential):
    """4 x ConvBNReLU(256, 256, 3) + Conv2d(256, out_planes, 3) (reference fpn.py:10-18)."""

    def __init__(self, out_planes):
        layers = [ConvBNReLU(256, 256, 3) for _ in range(4)]
        layers += [nn.Conv2d(256, out_planes, 3, padding=1)]
        super(SharedHead, self).__init__(*layers)

    def train(self, mode=True):
        self.__dict__["_final_pack"] = None
        return super(SharedHead, self).train(mode)

    def _apply(self, fn, *a, **kw):
        self.__dict__["_final_pack"] = None
        return super(SharedHead, self)._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self.__dict__["_final_pack"] = None
        return super(SharedHead, self)._load_from_state_dict(*a, **kw)

    def forward(self, x, final_act="none"):
        for m in list(self.children())[:-1]:
            x = m(x)
        last = list(self.children())[-1]
        return _final_conv(self, last, x, final_act)


def _final_conv(owner, conv, x, act):
    """Last conv of a tower: NCHW output (the layout decode consumes), bias + optional sigmoid fused."""
    import torch

    from ssds.modeling.layers import fused_conv as FC

    if (not owner.training and FC.fused_enabled() and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)
            and FC.conv_kind(conv) == "dense"):
        cache = owner.__dict__.get("_final_pack")
        if cache is None or cache[0] != x.dtype:
            cache = (x.dtype, FC.ConvPack(conv, None, "none", x.dtype))
            owner.__dict__["_final_pack"] = cache
        return FC.conv_native(x, cache[1], act=act, nchw_out=True)
    y = conv(x)
    return y.sigmoid() if act == "sigmoid" else y


class SSDFPN(NeckPlanMixin, SSDSBase):
    """RetinaNet (https://arxiv.org/abs/1708.02002) with ConvBNReLU extras/towers like the reference."""

    def __init__(self, backbone, extras, head, num_classes):
        super(SSDFPN, self).__init__(backbone, num_classes)
        self.transforms = nn.ModuleList(extras[0])
        self.extras = nn.ModuleList(extras[1])
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

    def _towers(self, xx, loc, conf):
        loc.append(self.loc(xx, "none"))
        conf.append(self.conf(xx, "none" if self.training else "sigmoid"))

    def _build_neck_plan(self, features, image=None):
        from ssds.modeling.layers.planner import build_fpn_plan

        return build_fpn_plan(self, features, image=image)

    def forward(self, x):
        out = self._full_native(x)  # planned backbone (MobileNet / ResNet): image -> heads is one plan
        if out is not None:
            return out
        loc, conf = [], []
        features = self.backbone(x)
        out = self._neck_native(features)  # eval on a HIP device: laterals, top-down adds, extras, towers = one plan
        if out is not None:
            return out
        x = features[-1]
        n = len(features)
        xx = None
        for i in range(n - 1, -1, -1):  # top-down pathway (reference fpn.py:80-87)
            lateral = self.transforms[i](features[i])
            xx = lateral if i == n - 1 else F.interpolate(xx, scale_factor=2, mode="nearest") + lateral
            features[i] = xx
        for i, v in enumerate(self.extras):  # reference fpn.py:89-97
            if i < n:
                xx = v(features[i])
            elif i == n:
                xx = v(x)
            else:
                xx = v(xx)
            self._towers(xx, loc, conf)
        return tuple(loc), tuple(conf)

    @staticmethod
    def add_extras(feature_layer, mbox, num_classes):
        """ints -> backbone output + 1x1 lateral (bias, no BN) + 3x3 ConvBNReLU; "Conv:S" -> stride-2
        ConvBNReLU on the previous map; one pair of shared towers (reference fpn.py:103-147)."""
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
        return nets_outputs, (transform_layers, extra_layers), (loc_layers, conf_layers)
