"""ResNet feature extractors (levels 2..5 = layer1..layer4) -- the interface of the reference's
``ssds/modeling/nets/resnet.py`` (forward :41-56, factories :59-152).  Parameter names follow the
torchvision layout (``conv1, bn1, layer{1..4}.{i}.conv{1,2,3}/bn{1,2,3}/downsample.{0,1}``) so
ImageNet / reference checkpoints load; torchvision itself is not a dependency and the classifier tail is
not instantiated."""
import torch.nn as nn

from .rutils import register


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super(Bottleneck, self).__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    """``forward(x)`` -> list of the maps of the levels in ``outputs`` (level i+2 = layer{i+1})."""

    def __init__(self, layers=[3, 4, 6, 3], bottleneck=Bottleneck, outputs=[5], groups=1,
                 width_per_group=64, url=None):
        super(ResNet, self).__init__()
        self.outputs = outputs
        self.url = url
        self.groups, self.base_width, self.inplanes = groups, width_per_group, 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(bottleneck, 64, layers[0])
        self.layer2 = self._make_layer(bottleneck, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(bottleneck, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(bottleneck, 512, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width))
        return nn.Sequential(*layers)

    def initialize(self):
        """No network on the target systems: load pretrained weights via cfg.RESUME_CHECKPOINT
        (the reference downloads ``self.url`` here, resnet.py:37-39)."""
        return None

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outputs = []
        for i, layer in enumerate([self.layer1, self.layer2, self.layer3, self.layer4]):
            level = i + 2
            if level > max(self.outputs):
                break
            x = layer(x)
            if level in self.outputs:
                outputs.append(x)
        return outputs


@register
def ResNet18(outputs, **kwargs):
    return ResNet(layers=[2, 2, 2, 2], bottleneck=BasicBlock, outputs=outputs)


@register
def ResNet34(outputs, **kwargs):
    return ResNet(layers=[3, 4, 6, 3], bottleneck=BasicBlock, outputs=outputs)


@register
def ResNet50(outputs, **kwargs):
    return ResNet(layers=[3, 4, 6, 3], bottleneck=Bottleneck, outputs=outputs)


@register
def ResNet101(outputs, **kwargs):
    return ResNet(layers=[3, 4, 23, 3], bottleneck=Bottleneck, outputs=outputs)


@register
def ResNet152(outputs, **kwargs):
    return ResNet(layers=[3, 8, 36, 3], bottleneck=Bottleneck, outputs=outputs)


@register
def ResNeXt50_32x4d(outputs, **kwargs):
    return ResNet(layers=[3, 4, 6, 3], bottleneck=Bottleneck, outputs=outputs, groups=32, width_per_group=4)


@register
def ResNeXt101_32x8d(outputs, **kwargs):
    return ResNet(layers=[3, 4, 23, 3], bottleneck=Bottleneck, outputs=outputs, groups=32, width_per_group=8)
