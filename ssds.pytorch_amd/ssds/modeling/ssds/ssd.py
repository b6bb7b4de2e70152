"""SSD multibox detector head -- same constructor, ``add_extras`` factory, module names (``extras``,
``loc``, ``conf``) and forward contract as the reference's ``ssds/modeling/ssds/ssd.py``:

    forward(x[B,3,H,W]) -> (tuple loc_l [B, A*4, h_l, w_l], tuple conf_l [B, A*C, h_l, w_l])

conf = logits in training mode, sigmoid probabilities in eval mode (reference ssd.py:72-73).

MI355X execution: in eval mode on a HIP device the two bare 3x3 head convs of a level
(reference ssd.py:100-103) run on the MFMA implicit-GEMM kernel with the bias and the sigmoid fused into
the epilogue (``ssdk_conv_bn_act``); the extras (1x1 + 3x3/s2 Conv-BN-ReLU pairs) run on the same kernel
with folded BatchNorm.  Training mode is ordinary autograd (MIOpen)."""
import torch.nn as nn

from ssds.modeling.layers.fused_conv import conv_bn_act_native, conv_supported, fold_bn, fused_enabled
from ssds.modeling.layers.layers_parser import parse_feature_layer

from .ssdsbase import SSDSBase


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

    def _head(self, x, conv, act):
        if not self.training and fused_enabled() and conv_supported(conv, x):
            scale, bias = fold_bn(conv, None)
            return conv_bn_act_native(x, conv.weight.detach(), None, bias, conv.kernel_size[0],
                                      conv.stride[0], act)
        y = conv(x)
        return y.sigmoid() if act == "sigmoid" else y

    def forward(self, x):
        loc, conf = [], []
        features = self.backbone(x)
        for v in self.extras:  # each extra consumes the previous last feature (reference ssd.py:63-65)
            features.append(v(features[-1]))
        conf_act = "none" if self.training else "sigmoid"
        for f, l, c in zip(features, self.loc, self.conf):
            loc.append(self._head(f, l, "none"))
            conf.append(self._head(f, c, conf_act))
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
