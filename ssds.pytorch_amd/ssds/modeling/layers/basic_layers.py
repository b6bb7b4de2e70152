"""Conv+BN+ReLU building blocks of the SSD/FPN heads and extras -- module/parameter layout identical to
the reference's ``ssds/modeling/layers/basic_layers.py:5-57`` (nn.Sequential indices), so state_dicts are
interchangeable.  In eval mode on a HIP device each (conv, bn, relu) triple runs as ONE fused MFMA
implicit-GEMM launch (``ssdk_conv_bn_act``: BN folded into scale/bias, ReLU in the epilogue); in training
mode the triple is ordinary torch autograd (MIOpen)."""
import torch.nn as nn

from .dwconv import make_conv2d
from .fused_conv import FusedSequentialMixin


class SepConvBNReLU(FusedSequentialMixin, nn.Sequential):
    """depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN + ReLU (reference basic_layers.py:5-25)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, expand_ratio=1):
        padding = (kernel_size - 1) // 2
        super(SepConvBNReLU, self).__init__(
            make_conv2d(in_planes, in_planes, kernel_size, stride, padding, groups=in_planes, bias=False),
            nn.BatchNorm2d(in_planes),
            nn.ReLU(inplace=True),
            nn.Conv2d(in_planes, out_planes, 1, 1, 0, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU(inplace=True),
        )


class ConvBNReLU(FusedSequentialMixin, nn.Sequential):
    """k x k conv + BN + ReLU (reference basic_layers.py:28-37)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1):
        padding = (kernel_size - 1) // 2
        super(ConvBNReLU, self).__init__(
            nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding=padding, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU(inplace=True),
        )


class ConvBNReLUx2(FusedSequentialMixin, nn.Sequential):
    """1x1 (to half width) + BN + ReLU, k x k (stride s) + BN + ReLU: the SSD "Conv:S" extra
    (reference basic_layers.py:40-57)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1):
        padding = (kernel_size - 1) // 2
        super(ConvBNReLUx2, self).__init__(
            nn.Conv2d(in_planes, out_planes // 2, 1, bias=False),
            nn.BatchNorm2d(out_planes // 2),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_planes // 2, out_planes, kernel_size, stride, padding=padding, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU(inplace=True),
        )
