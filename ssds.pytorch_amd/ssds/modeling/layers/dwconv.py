"""Depthwise 3x3 convolution with hand-written HIP forward / input-gradient / weight-gradient kernels
(``csrc/ssdk_dwplane.hip``: whole-row bands of the planes; ``csrc/ssdk_dwtrain.hip``: the tiled fallback) behind ``torch.autograd`` -- the training-step replacement for what PyTorch-ROCm
dispatches to MIOpen's ``naive_conv_*`` kernels (more than half of the GPU time of the reference's DDP step on
SSD-MobileNetV2: pipeline_anchor_apex.py:75-171 over torchvision ``InvertedResidual`` blocks, mobilenet.py:56).

``DepthwiseConv2d`` is an ``nn.Conv2d`` (same parameters, same ``state_dict`` keys, same initialisation); on CPU
tensors or for geometries the kernels do not cover it IS ``nn.Conv2d``."""
import ctypes

import torch
import torch.nn as nn

from ssds import _native as N


def _launch(fn, *args):
    N.check(fn(*args), fn.__name__)


class _DwConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, want_sums=False, coef=None, act=0):
        """``want_sums``: also return sums [C, 2] = per-channel (sum y, sum y^2) of the outputs -- the batch statistics of the
        BatchNorm that follows (ssdk_dwconv_fwd_stats: no pass over y); non-differentiable; None where the geometry runs on the
        tiled fallback kernels."""
        x = x.contiguous()
        ctx.wdt = w.dtype
        # an fp32 master weight next to a 16-bit tensor (autocast): cast here, outside autograd -- the fp32 weight gradient then
        # reaches the parameter as it is, without autocast's bf16 round trip and its two cast launches per layer and step
        w = w.detach().to(x.dtype).contiguous()
        n, c, h, wd = (int(v) for v in x.shape)
        ho, wo = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
        y = torch.empty((n, c, ho, wo), device=x.device, dtype=x.dtype)
        sums = None
        with torch.cuda.device(x.device):
            need = int(N.lib.ssdk_dwconv_fwd_stats_workspace_bytes(n, c, h, wd, stride, N.dtype_code(x))) if want_sums else 0
            if coef is not None:  # x is the INPUT of a deferred BatchNorm: the kernel stages act(a x + b) (batchnorm._BatchNormDeferred)
                ws = torch.empty(need + 16, dtype=torch.uint8, device=x.device) if need else None
                if need:
                    sums = torch.empty((c, 2), device=x.device, dtype=torch.float32)
                N.check(N.lib.ssdk_dwconv_fwd_affine(x.data_ptr(), coef.data_ptr(), int(act), w.data_ptr(), y.data_ptr(),
                                                     None if sums is None else sums.data_ptr(),
                                                     None if ws is None else (ws.data_ptr() + 15) & ~15, need, n, c, h, wd, stride,
                                                     N.dtype_code(x), N.stream_ptr(x.device)), "dwconv_fwd_affine")
            elif need:
                ws = torch.empty(need + 16, dtype=torch.uint8, device=x.device)
                sums = torch.empty((c, 2), device=x.device, dtype=torch.float32)
                N.check(N.lib.ssdk_dwconv_fwd_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), sums.data_ptr(), (ws.data_ptr() + 15) & ~15,
                                                    need, n, c, h, wd, stride, N.dtype_code(x), N.stream_ptr(x.device)), "dwconv_fwd_stats")
            else:
                _launch(N.lib.ssdk_dwconv_fwd, x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, h, wd, stride, N.dtype_code(x),
                        N.stream_ptr(x.device))
        if coef is not None:
            ctx.save_for_backward(x, w, coef)
        else:
            ctx.save_for_backward(x, w)
        ctx.stride = stride
        ctx.act = int(act)
        if want_sums:
            if sums is None:
                sums = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(sums)
            ctx.set_materialize_grads(False)  # (no zero-filled gradient tensor for sums: a fill launch per layer and step)
            return y, sums
        return y

    @staticmethod
    def backward(ctx, gy, _gsums=None):
        coef = None
        if len(ctx.saved_tensors) == 3:
            x, w, coef = ctx.saved_tensors
        else:
            x, w = ctx.saved_tensors
        stride = ctx.stride
        if gy is None:  # (only the non-differentiable statistics were used)
            return None, None, None, None, None, None
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        n, c, h, wd = (int(v) for v in x.shape)
        gx = gw = None
        dev = x.device
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _launch(N.lib.ssdk_dwconv_bwd_data, gy.data_ptr(), w.data_ptr(), gx.data_ptr(), n, c, h, wd, stride,
                        N.dtype_code(x), N.stream_ptr(dev))
            if ctx.needs_input_grad[1]:
                need = int(N.lib.ssdk_dwconv_bwd_weight_workspace_bytes(n, c, h, wd, stride))
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                gw32 = torch.empty((c, 1, 3, 3), device=dev, dtype=torch.float32)
                if coef is not None:
                    N.check(N.lib.ssdk_dwconv_bwd_weight_affine(x.data_ptr(), coef.data_ptr(), ctx.act, gy.data_ptr(), gw32.data_ptr(),
                                                                ws.data_ptr(), need, n, c, h, wd, stride, N.dtype_code(x),
                                                                N.stream_ptr(dev)), "dwconv_bwd_weight_affine")
                else:
                    _launch(N.lib.ssdk_dwconv_bwd_weight, x.data_ptr(), gy.data_ptr(), gw32.data_ptr(), ws.data_ptr(), need,
                            n, c, h, wd, stride, N.dtype_code(x), N.stream_ptr(dev))
                gw = gw32.to(ctx.wdt)
        return gx, gw, None, None, None, None


def dwconv3x3(x, weight, stride, want_sums=False, pending=None):
    """Depthwise 3x3, pad 1, no bias: x [N,C,H,W], weight [C,1,3,3] (same floating dtype), differentiable.  ``want_sums``: the
    output carries ``_ssdk_bn_sums`` = [C, 2] for the kernel-backed BatchNorm behind it (see pointwise.pointwise_conv)."""
    coef, act = pending if pending is not None else (None, 0)
    if want_sums:
        y, sums = _DwConv3x3.apply(x, weight, stride, True, coef, act)
        if sums.numel():
            y._ssdk_bn_sums = sums
        return y
    return _DwConv3x3.apply(x, weight, stride, False, coef, act)


class DepthwiseConv2d(nn.Conv2d):
    """``nn.Conv2d(C, C, 3, stride, 1, groups=C, bias=False)`` whose HIP-device forward/backward run on the
    ssdk kernels; anything else falls through to ``nn.Conv2d.forward``."""
    _ssdk_bn_follows = False  # set by pointwise.fuse_conv_bn_statistics: the next module is a kernel-backed BatchNorm
    _ssdk_expect_pending = False  # set by a deferred BatchNorm right before its (alias) output reaches this module

    def _native(self, x):
        return (x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.stride[0] == self.stride[1] and self.stride[0] in (1, 2)
                and self.groups == self.in_channels == self.out_channels and self.bias is None
                and self.padding_mode == "zeros")

    def forward(self, x):
        pending = x.__dict__.pop("_ssdk_pending_bn", None) if hasattr(x, "__dict__") else None
        expect, self._ssdk_expect_pending = self._ssdk_expect_pending, False
        if expect and pending is None:  # (would be silently wrong: the tensor in hand is the BatchNorm's INPUT)
            raise RuntimeError("DepthwiseConv2d: the deferred BatchNorm's coefficients did not arrive with its output")
        if pending is not None and not (self._native(x) and x.dtype in (torch.bfloat16, torch.float16)):
            raise RuntimeError("DepthwiseConv2d: a deferred BatchNorm output reached a path that cannot apply it")
        if not self._native(x):
            return super(DepthwiseConv2d, self).forward(x)
        w = self.weight
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
            if not (w.dtype == torch.float32 and x.dtype in (torch.bfloat16, torch.float16)):
                w = w.to(x.dtype)  # (fp32 master weights go in as they are: _DwConv3x3 casts outside autograd)
        elif w.dtype != x.dtype:
            w = w.to(x.dtype)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return super(DepthwiseConv2d, self).forward(x)
        from ssds.modeling.layers import pointwise as _pw

        ho, wo = (x.shape[2] - 1) // self.stride[0] + 1, (x.shape[3] - 1) // self.stride[0] + 1
        want = (self._ssdk_bn_follows and self.training and x.dtype != torch.float32
                and x.shape[0] * x.shape[1] * ho * wo * 2 >= _pw.BN_STATS_MIN_BYTES)
        if pending is not None and x.dtype != (torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype):
            raise RuntimeError("DepthwiseConv2d: dtype changed behind a deferred BatchNorm")
        with torch.autocast("cuda", enabled=False):
            return dwconv3x3(x, w, self.stride[0], want_sums=want, pending=pending)


def make_conv2d(in_planes, out_planes, kernel_size, stride=1, padding=0, groups=1, bias=True):
    """``nn.Conv2d`` factory used by the backbones: depthwise 3x3 convolutions get the kernel-backed subclass."""
    if groups == in_planes == out_planes and groups > 1 and kernel_size == 3 and padding == 1 and not bias:
        return DepthwiseConv2d(in_planes, out_planes, 3, stride, 1, groups=groups, bias=False)
    return nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, groups=groups, bias=bias)
