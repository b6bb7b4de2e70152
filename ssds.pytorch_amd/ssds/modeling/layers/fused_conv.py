"""Host side of the fused conv + folded-BN + activation kernel (C-ABI ``ssdk_conv_bn_act``).

``fused_conv_bn_act(x, conv, bn, act)`` folds the BatchNorm running statistics into a per-channel
(scale, bias) pair and launches the MFMA implicit-GEMM kernel.  ``FusedSequentialMixin`` lets the
reference-shaped nn.Sequential blocks use it in eval mode without changing their state_dict layout.

Switch: ``SSDK_FUSED_CONV`` = "1" (default: use the HIP kernel for dense 1x1/3x3 convs in eval mode on a
HIP device) or "0" (always torch/MIOpen; used by A/B measurements, never silently).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ssds import _native as N

_ACT_OF = {nn.ReLU: "relu", nn.ReLU6: "relu6", nn.SiLU: "silu", nn.Sigmoid: "sigmoid"}


def fused_enabled():
    return os.environ.get("SSDK_FUSED_CONV", "1") != "0"


def fold_bn(conv, bn):
    """-> (scale[Cout], bias[Cout]) fp32 such that bn(conv(x)) == conv_nobias(x) * scale + bias."""
    cout = conv.out_channels
    dev = conv.weight.device
    if bn is None:
        scale = torch.ones(cout, device=dev, dtype=torch.float32)
        bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout, device=dev)
        return scale, bias
    var = bn.running_var.detach().float()
    mean = bn.running_mean.detach().float()
    gamma = bn.weight.detach().float() if bn.affine else torch.ones_like(var)
    beta = bn.bias.detach().float() if bn.affine else torch.zeros_like(var)
    scale = gamma / torch.sqrt(var + bn.eps)
    b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(mean)
    bias = (b0 - mean) * scale + beta
    return scale, bias


def conv_supported(conv, x):
    k = conv.kernel_size
    return (
        x.is_cuda
        and x.dtype in (torch.bfloat16, torch.float16)
        and conv.groups == 1
        and k[0] == k[1]
        and k[0] in (1, 3)
        and conv.stride[0] == conv.stride[1]
        and conv.stride[0] in (1, 2)
        and conv.padding[0] == k[0] // 2
        and conv.padding[1] == k[0] // 2
        and conv.dilation == (1, 1)
        and conv.padding_mode == "zeros"
    )


def conv_bn_act_native(x, weight, scale, bias, k, stride, act="none", out_dtype=None):
    """y = act(conv(x, weight) * scale + bias) on the MFMA kernel.  x [N,Cin,H,W] NCHW bf16/f16."""
    N.require_device(x, "conv_bn_act")
    x = x.contiguous()
    weight = weight.contiguous().to(x.dtype)
    n, cin, h, w = (int(v) for v in x.shape)
    cout = int(weight.shape[0])
    ho = (h + 2 * (k // 2) - k) // stride + 1
    wo = (w + 2 * (k // 2) - k) // stride + 1
    out_dtype = out_dtype or x.dtype
    y = torch.empty((n, cout, ho, wo), device=x.device, dtype=out_dtype)
    dt = N.dtype_code(x)
    with torch.cuda.device(x.device):
        need = N.lib.ssdk_conv_workspace_bytes(n, cin, h, w, cout, k, stride, dt)
        ws = N.workspace(x.device, need)
        rc = N.lib.ssdk_conv_bn_act(
            x.data_ptr(), weight.data_ptr(), scale.data_ptr() if scale is not None else None,
            bias.data_ptr(), n, cin, h, w, cout, k, stride, N.ACT[act], dt, N.dtype_code(y),
            y.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr(x.device))
    N.check(rc, "conv_bn_act")
    return y


class FusedSequentialMixin(object):
    """nn.Sequential of [Conv2d, (BatchNorm2d), (activation)]* groups: in eval mode on a HIP device run
    each group as one fused launch; otherwise (training / CPU / unsupported conv) the plain torch ops."""

    def _groups(self):
        mods = list(self.children())
        i, out = 0, []
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, nn.Conv2d):
                return None
            bn, act, j = None, "none", i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
                bn, j = mods[j], j + 1
            if j < len(mods) and type(mods[j]) in _ACT_OF:
                act, j = _ACT_OF[type(mods[j])], j + 1
            out.append((conv, bn, act))
            i = j
        return out

    def forward(self, x):
        if self.training or not fused_enabled() or not x.is_cuda:
            return nn.Sequential.forward(self, x)
        groups = self._groups()
        if groups is None:
            return nn.Sequential.forward(self, x)
        for conv, bn, act in groups:
            if conv_supported(conv, x):
                scale, bias = fold_bn(conv, bn)
                x = conv_bn_act_native(x, conv.weight.detach(), scale, bias, conv.kernel_size[0],
                                       conv.stride[0], act)
            else:
                x = conv(x)
                if bn is not None:
                    x = bn(x)
                if act == "relu":
                    x = F.relu(x)
                elif act == "relu6":
                    x = F.relu6(x)
                elif act == "silu":
                    x = F.silu(x)
                elif act == "sigmoid":
                    x = torch.sigmoid(x)
        return x
