"""MobileNet v1 / v2 feature extractors returning the feature maps named by ``outputs`` -- the
interface of the reference's ``ssds/modeling/nets/mobilenet.py`` (``MobileNetEx.forward`` :180-192,
factories :195-212).  Parameter names follow the reference / torchvision layout (``conv1``,
``layer{1..7}.{i}.conv...``) so reference checkpoints load; the unused classifier tail
(``head_conv``, ``classifier``; reference :91-99) is not instantiated.

torchvision is not a dependency: the (standard) inverted-residual block is defined here.  Depthwise
and pointwise convolutions are HBM-bound and run on PyTorch-ROCm/MIOpen."""
import torch.nn as nn

from ssds.modeling.layers.dwconv import make_conv2d

from .rutils import register


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvBNReLU6(nn.Sequential):
    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1):
        padding = (kernel_size - 1) // 2
        super(ConvBNReLU6, self).__init__(
            make_conv2d(in_planes, out_planes, kernel_size, stride, padding, groups=groups, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU6(inplace=True),
        )


class InvertedResidual(nn.Module):
    """MobileNetV2 block: 1x1 expand -> 3x3 depthwise -> 1x1 linear project (+ residual)."""

    def __init__(self, inp, oup, stride, expand_ratio):
        super(InvertedResidual, self).__init__()
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU6(inp, hidden_dim, kernel_size=1))
        layers.extend([
            ConvBNReLU6(hidden_dim, hidden_dim, stride=stride, groups=hidden_dim),
            nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False),
            nn.BatchNorm2d(oup),
        ])
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class SepConvBNReLU6(nn.Sequential):
    """MobileNetV1 block (reference mobilenet.py:8-28)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, expand_ratio=1):
        padding = (kernel_size - 1) // 2
        super(SepConvBNReLU6, self).__init__(
            make_conv2d(in_planes, in_planes, kernel_size, stride, padding, groups=in_planes, bias=False),
            nn.BatchNorm2d(in_planes),
            nn.ReLU6(inplace=True),
            nn.Conv2d(in_planes, out_planes, 1, 1, 0, bias=False),
            nn.BatchNorm2d(out_planes),
            nn.ReLU6(inplace=True),
        )


_SETTINGS = {
    # t, c, n, s
    "v2": [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
           [6, 320, 1, 1]],
    "v1": [[1, 64, 1, 1], [1, 128, 2, 2], [1, 256, 2, 2], [1, 512, 6, 2], [1, 1024, 2, 2]],
}


class MobileNetEx(nn.Module):
    """``forward(x)`` -> list of the feature maps of the levels in ``outputs`` (level j = ``layer{j}``),
    stopping after the deepest requested level (reference mobilenet.py:180-192)."""

    def __init__(self, width_mult=1.0, version="v1", outputs=[7], url=None, round_nearest=8):
        super(MobileNetEx, self).__init__()
        self.version = version
        self.settings = _SETTINGS[version]
        self.outputs = outputs
        self.url = url
        block = InvertedResidual if version == "v2" else SepConvBNReLU6
        input_channel = _make_divisible(32 * width_mult, round_nearest)
        self.conv1 = ConvBNReLU6(3, input_channel, stride=2)
        self.out_channels = {}
        for j, (t, c, n, s) in enumerate(self.settings):
            output_channel = _make_divisible(c * width_mult, round_nearest)
            layers = []
            for i in range(n):
                layers.append(block(input_channel, output_channel, stride=s if i == 0 else 1, expand_ratio=t))
                input_channel = output_channel
            self.add_module("layer{}".format(j + 1), nn.Sequential(*layers))
            self.out_channels[j + 1] = output_channel
        for m in self.modules():  # reference mobilenet.py:101-112
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def initialize(self):
        """The reference downloads ImageNet weights here (mobilenet.py:131-178).  There is no network on
        the target systems: pretrained weights are loaded explicitly through
        ``ssds.core.checkpoint.resume_checkpoint`` (cfg.RESUME_CHECKPOINT) instead."""
        return None

    def forward(self, x):
        x = self.conv1(x)
        outputs = []
        for j in range(len(self.settings)):
            level = j + 1
            if level > max(self.outputs):
                break
            x = getattr(self, "layer{}".format(level))(x)
            if level in self.outputs:
                outputs.append(x)
        return outputs


@register
def MobileNetV1(outputs, **kwargs):
    return MobileNetEx(width_mult=1.0, version="v1", outputs=outputs)


@register
def MobileNetV2(outputs, **kwargs):
    return MobileNetEx(width_mult=1.0, version="v2", outputs=outputs)
