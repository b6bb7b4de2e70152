"""Minimal stand-in for the pieces of ``torchvision.models`` the REFERENCE backbones subclass, so that
``/root/reference/ssds/modeling/nets/{mobilenet,resnet}.py`` can be imported unmodified in the build
container (torchvision is not installed; the reference pins none: README.md:32, Dockerfile:1 -> the
mid-2020 0.7 API with ``mobilenet.ConvBNReLU`` / ``model_urls``).

Used ONLY by ``make_golden.py`` ("a small in-repo torchvision shim ... for the oracle only").
The blocks below restate the published MobileNetV2 / ResNet architectures (Sandler et al. 2018, He et al.
2015) with torchvision 0.7's module and parameter names, which is what fixes the ``state_dict`` keys of the
reference models.  They are NOT part of the product and nothing under ``ssds.pytorch_amd/`` imports them:
backbone parity is therefore "reference wiring around restated standard blocks" and is labelled so in
DESIGN.md.  RegNet needs no shim (``nets/regnet.py`` is self-contained).
"""
import sys
import types

import torch.nn as nn

__version__ = "0.7.0"


# ------------------------------------------------------------------------------------------ mobilenet
def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvBNReLU(nn.Sequential):
    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1):
        super().__init__(
            nn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU6(inplace=True),
        )


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        self.stride = stride
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = self.stride == 1 and inp == oup
        seq = []
        if expand_ratio != 1:
            seq.append(ConvBNReLU(inp, hidden, kernel_size=1))
        seq += [ConvBNReLU(hidden, hidden, stride=stride, groups=hidden),
                nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        y = self.conv(x)
        return x + y if self.use_res_connect else y


# --------------------------------------------------------------------------------------------- resnet
def _conv3x3(i, o, stride=1, groups=1):
    return nn.Conv2d(i, o, 3, stride, 1, groups=groups, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv3x3(width, width, stride, groups)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        self.inplanes = 64
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            seq.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width))
        return nn.Sequential(*seq)


def install():
    """Register ``torchvision``, ``torchvision.models`` and the two sub-modules the reference imports."""
    tv = types.ModuleType("torchvision")
    tv.__version__ = __version__
    models = types.ModuleType("torchvision.models")
    mob = types.ModuleType("torchvision.models.mobilenet")
    mob._make_divisible, mob.ConvBNReLU, mob.InvertedResidual = _make_divisible, ConvBNReLU, InvertedResidual
    mob.model_urls = {"mobilenet_v2": None}
    res = types.ModuleType("torchvision.models.resnet")
    res.ResNet, res.BasicBlock, res.Bottleneck = ResNet, BasicBlock, Bottleneck
    res.model_urls = {k: None for k in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152",
                                        "resnext50_32x4d", "resnext101_32x8d")}
    models.mobilenet, models.resnet = mob, res
    tv.models = models
    for name, m in (("torchvision", tv), ("torchvision.models", models), ("torchvision.models.mobilenet", mob),
                    ("torchvision.models.resnet", res)):
        sys.modules[name] = m
