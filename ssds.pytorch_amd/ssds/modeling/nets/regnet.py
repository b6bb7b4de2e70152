"""RegNetX feature extractors (levels 1..4 = stages s1..s4) -- the interface of the reference's
``ssds/modeling/nets/regnet.py`` (RegNet.forward :270-282, factories from :300).  Widths/depths come from
the published RegNet design-space rule (quantised linear widths, Radosavovic et al. 2020); module names
(``stem.conv/bn``, ``s{i}.b{j}.f.{a,b,c}[_bn]``, ``proj/bn``) follow pycls so its checkpoints load.  The
classification head is not instantiated."""
import math

import numpy as np
import torch.nn as nn

from .rutils import register


class _Stem(nn.Module):
    def __init__(self, w_in, w_out):
        super(_Stem, self).__init__()
        self.conv = nn.Conv2d(w_in, w_out, 3, stride=2, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(w_out)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class _Transform(nn.Module):
    """1x1 -> grouped 3x3 (stride) -> 1x1, each with BN, ReLU after the first two."""

    def __init__(self, w_in, w_out, stride, bm, gw):
        super(_Transform, self).__init__()
        w_b = int(round(w_out * bm))
        self.a = nn.Conv2d(w_in, w_b, 1, bias=False)
        self.a_bn = nn.BatchNorm2d(w_b)
        self.a_relu = nn.ReLU(inplace=True)
        self.b = nn.Conv2d(w_b, w_b, 3, stride=stride, padding=1, groups=w_b // gw, bias=False)
        self.b_bn = nn.BatchNorm2d(w_b)
        self.b_relu = nn.ReLU(inplace=True)
        self.c = nn.Conv2d(w_b, w_out, 1, bias=False)
        self.c_bn = nn.BatchNorm2d(w_out)
        self.c_bn.final_bn = True

    def forward(self, x):
        x = self.a_relu(self.a_bn(self.a(x)))
        x = self.b_relu(self.b_bn(self.b(x)))
        return self.c_bn(self.c(x))


class _Block(nn.Module):
    def __init__(self, w_in, w_out, stride, bm, gw):
        super(_Block, self).__init__()
        self.proj_block = (w_in != w_out) or (stride != 1)
        if self.proj_block:
            self.proj = nn.Conv2d(w_in, w_out, 1, stride=stride, bias=False)
            self.bn = nn.BatchNorm2d(w_out)
        self.f = _Transform(w_in, w_out, stride, bm, gw)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        skip = self.bn(self.proj(x)) if self.proj_block else x
        return self.relu(skip + self.f(x))


class _Stage(nn.Module):
    def __init__(self, w_in, w_out, stride, d, bm, gw):
        super(_Stage, self).__init__()
        for i in range(d):
            self.add_module("b{}".format(i + 1), _Block(w_in if i == 0 else w_out, w_out,
                                                        stride if i == 0 else 1, bm, gw))

    def forward(self, x):
        for blk in self.children():
            x = blk(x)
        return x


def regnet_stages(w_a, w_0, w_m, d, group_w, bot_mul, q=8):
    """-> (widths, depths, group widths) per stage from the RegNet parameters."""
    ws_cont = np.arange(d) * w_a + w_0
    ks = np.round(np.log(ws_cont / w_0) / np.log(w_m))
    ws = (np.round(w_0 * np.power(w_m, ks) / q) * q).astype(int).tolist()
    s_ws, s_ds = [], []
    for w in ws:  # run-length encode the per-block widths into stages
        if s_ws and s_ws[-1] == w:
            s_ds[-1] += 1
        else:
            s_ws.append(w)
            s_ds.append(1)
    # make bottleneck widths divisible by the group width
    w_bots = [int(w * bot_mul) for w in s_ws]
    gs = [min(group_w, wb) for wb in w_bots]
    w_bots = [int(round(wb / g) * g) for wb, g in zip(w_bots, gs)]
    s_ws = [int(wb / bot_mul) for wb in w_bots]
    return s_ws, s_ds, gs


class RegNet(nn.Module):
    """``forward(x)`` -> list of the maps of the stages in ``outputs`` (level i+1 = s{i+1})."""

    def __init__(self, w_a, w_0, w_m, d, group_w, bot_mul, outputs=[4], url=None, **kwargs):
        super(RegNet, self).__init__()
        self.outputs = outputs
        self.url = url
        ws, ds, gs = regnet_stages(w_a, w_0, w_m, d, group_w, bot_mul)
        self.stage_widths = ws
        self.stem = _Stem(3, 32)
        prev = 32
        for i, (dd, w, g) in enumerate(zip(ds, ws, gs)):
            self.add_module("s{}".format(i + 1), _Stage(prev, w, 2, dd, bot_mul, g))
            prev = w
        for m in self.modules():  # pycls initialisation (reference regnet.py:154-166)
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(mean=0.0, std=math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(0.0 if getattr(m, "final_bn", False) else 1.0)
                m.bias.data.zero_()

    def initialize(self):
        """No network on the target systems: load pretrained weights via cfg.RESUME_CHECKPOINT."""
        return None

    def forward(self, x):
        x = self.stem(x)
        outputs = []
        for i, layer in enumerate([self.s1, self.s2, self.s3, self.s4]):
            level = i + 1
            if level > max(self.outputs):
                break
            x = layer(x)
            if level in self.outputs:
                outputs.append(x)
        return outputs


_REGNETX = {  # name: (w_a, w_0, w_m, d, group_w)
    "RegNetX002": (36.44, 24, 2.49, 13, 8),
    "RegNetX004": (24.48, 24, 2.54, 22, 16),
    "RegNetX006": (36.97, 48, 2.24, 16, 24),
    "RegNetX008": (35.73, 56, 2.28, 16, 16),
    "RegNetX016": (34.01, 80, 2.25, 18, 24),
    "RegNetX032": (26.31, 88, 2.25, 25, 48),
    "RegNetX040": (38.65, 96, 2.43, 23, 40),
    "RegNetX064": (60.83, 184, 2.07, 17, 56),
    "RegNetX080": (49.56, 80, 2.88, 23, 120),
}


def _factory(name):
    w_a, w_0, w_m, d, gw = _REGNETX[name]

    def make(outputs, **kwargs):
        return RegNet(w_a=w_a, w_0=w_0, w_m=w_m, d=d, group_w=gw, bot_mul=1, outputs=outputs)

    make.__name__ = name
    make.__module__ = __name__
    make.__doc__ = "{} backbone".format(name)
    return register(make)


for _name in _REGNETX:
    globals()[_name] = _factory(_name)
