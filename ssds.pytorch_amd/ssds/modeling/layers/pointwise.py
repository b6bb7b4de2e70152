"""1x1 convolutions of the training step as plain batched GEMMs on the NCHW tensors themselves.

PyTorch-ROCm sends ``nn.Conv2d(k=1)`` in bf16 to MIOpen, which runs NHWC implicit-GEMM kernels between
``batched_transpose_*`` layout changes (10 of the 27 ms of kernel time of the reference's DDP step on SSD-MobileNetV2,
pipeline_anchor_apex.py:75-171, once the depthwise convolutions and BatchNorm are on the ssdk kernels).  In NCHW a
pointwise convolution needs no layout change at all:

    y[b]  = W      @ x[b]          [Cout,Cin] @ [Cin,HW]      forward
    dx[b] = W^T    @ dy[b]         [Cin,Cout] @ [Cout,HW]     input gradient
    dW    = sum_b dy[b] @ x[b]^T   [Cout,HW]  @ [HW,Cin]      weight gradient (fp32 partials, summed in index order)

i.e. three strided-batched library GEMMs (rocBLAS / hipBLASLt through ``torch.matmul`` / ``torch.bmm``) on views of
the tensors the neighbouring kernels already produce.  ``PointwiseConv2d`` is an ``nn.Conv2d`` (same parameters,
``state_dict`` keys and initialisation); CPU tensors, channels-last tensors and anything that is not a dense 1x1 /
stride 1 / pad 0 convolution take ``nn.Conv2d.forward``."""
import torch
import torch.nn as nn


class _Pointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        # detached views: torch.matmul folds the batch into one GEMM -- through a transposing COPY of the activation --
        # whenever an operand requires grad (its own autograd heuristic); here that copy cost more than the GEMM
        x3 = x.detach().view(b, cin, h * wd)
        w2 = w.detach().view(cout, cin)
        y = torch.matmul(w2, x3)
        if bias is not None:
            y += bias.view(1, cout, 1)
        ctx.save_for_backward(x3, w2)
        ctx.has_bias = bias is not None
        return y.view(b, cout, h, wd)

    @staticmethod
    def backward(ctx, gy):
        x3, w2 = ctx.saved_tensors
        b, cin, hw = x3.shape
        cout = w2.shape[0]
        gy3 = gy.contiguous().view(b, cout, hw)
        if gy3.dtype != x3.dtype:
            gy3 = gy3.to(x3.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(w2.t(), gy3).view(b, cin, *gy.shape[2:])
        if ctx.needs_input_grad[1]:
            if x3.dtype == torch.float32:
                part = torch.bmm(gy3, x3.transpose(1, 2))
            else:
                part = torch.bmm(gy3, x3.transpose(1, 2), out_dtype=torch.float32)
            gw = part.sum(0).to(w2.dtype).view(cout, cin, 1, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy3.sum((0, 2), dtype=torch.float32).to(w2.dtype)
        return gx, gw, gb


def pointwise_conv(x, weight, bias=None):
    """1x1 / stride 1 convolution of a contiguous NCHW tensor: weight [Cout,Cin,1,1], same floating dtype; differentiable."""
    return _Pointwise.apply(x, weight, bias)


class PointwiseConv2d(nn.Conv2d):
    def _native(self, x):
        return (x.is_cuda and x.dim() == 4 and self.kernel_size == (1, 1) and self.stride == (1, 1)
                and self.padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == "zeros" and x.is_contiguous())

    def forward(self, x):
        if not self._native(x):
            return super(PointwiseConv2d, self).forward(x)
        w, bias = self.weight, self.bias
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, w = x.to(dt), w.to(dt)
            bias = bias.to(dt) if bias is not None else None
        elif w.dtype != x.dtype:
            return super(PointwiseConv2d, self).forward(x)
        with torch.autocast("cuda", enabled=False):
            return pointwise_conv(x, w, bias)


def use_pointwise_gemm(model):
    """Switch every dense 1x1 / stride-1 ``nn.Conv2d`` of ``model`` to the GEMM-backed subclass (in place; no new
    parameters, same ``state_dict``)."""
    for m in model.modules():
        if (type(m) is nn.Conv2d and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
                and m.groups == 1 and m.dilation == (1, 1)):
            m.__class__ = PointwiseConv2d
    return model
