"""1x1 convolutions of the training step on the NCHW tensors themselves: hand-written matrix-core kernels for 16-bit tensors
(round 6: csrc/ssdk_pwtrain.hip -- forward, input gradient, weight gradient; no library GEMM in the bf16 / fp16 step), plain
batched library GEMMs for fp32 tensors.

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
import os

import torch
import torch.nn as nn

from ssds import _native as N


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


def _native_ok(x, cin):
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and cin % 8 == 0
            and os.environ.get("SSDK_PW_NATIVE", "1") != "0")


class _PointwiseNative(torch.autograd.Function):
    """16-bit NCHW x, weight [Cout, Cin, 1, 1] either fp32 (the master parameter under autocast: cast + transposed copy in ONE
    launch, the weight gradient comes back in fp32 without a cast) or in x's dtype; bias fp32 / 16-bit or None.
        forward          ssdk_pw_prepare (fp32 weight) + ssdk_pw_forward(a = W)
        input gradient   ssdk_pw_forward(a = W^T)
        weight gradient  ssdk_pw_wgrad (fp32, fixed-order reduction)"""

    @staticmethod
    def forward(ctx, x, w, bias, want_sums=False):
        """``want_sums``: also return sums [Cout, 2] = per-channel (sum y, sum y^2) of the stored outputs -- the batch
        statistics of the BatchNorm that follows (ssdk_pw_forward_stats: no pass over y); non-differentiable."""
        b, cin, h, wd = (int(v) for v in x.shape)
        cout, hw, dev, dt = int(w.shape[0]), h * wd, x.device, x.dtype
        code = N.dtype_code(x)
        x = x.detach()
        with torch.cuda.device(dev):
            sp = N.stream_ptr(dev)
            if w.dtype == torch.float32:
                w16 = torch.empty((cout, cin), device=dev, dtype=dt)
                wt16 = torch.empty((cin, cout), device=dev, dtype=dt)
                N.check(N.lib.ssdk_pw_prepare(w.detach().data_ptr(), w16.data_ptr(), wt16.data_ptr(), cout, cin, code, sp), "pw_prepare")
            else:
                w16 = w.detach().reshape(cout, cin).contiguous()
                wt16 = w16.t().contiguous()
            b32 = None if bias is None else bias.detach().float().contiguous()
            y = torch.empty((b, cout, h, wd), device=dev, dtype=dt)
            sums = None
            if want_sums:
                need = int(N.lib.ssdk_pw_stats_workspace_bytes(b, cin, cout, hw))
                ws = torch.empty(need + 16, dtype=torch.uint8, device=dev)
                sums = torch.empty((cout, 2), device=dev, dtype=torch.float32)
                N.check(N.lib.ssdk_pw_forward_stats(x.data_ptr(), w16.data_ptr(), None if b32 is None else b32.data_ptr(), y.data_ptr(),
                                                    sums.data_ptr(), (ws.data_ptr() + 15) & ~15, need, b, cin, cout, hw, code, sp),
                        "pw_forward_stats")
            else:
                N.check(N.lib.ssdk_pw_forward(x.data_ptr(), w16.data_ptr(), None if b32 is None else b32.data_ptr(), y.data_ptr(),
                                              b, cin, cout, hw, code, sp), "pw_forward")
        ctx.save_for_backward(x, wt16)
        ctx.meta = (w.dtype, None if bias is None else bias.dtype)
        if want_sums:
            ctx.mark_non_differentiable(sums)
            ctx.set_materialize_grads(False)  # (no zero-filled gradient tensor for sums: a fill launch per layer and step)
            return y, sums
        return y

    @staticmethod
    def backward(ctx, gy, _gsums=None):
        x, wt16 = ctx.saved_tensors
        wdt, bdt = ctx.meta
        b, cin, h, wd = (int(v) for v in x.shape)
        cout, hw, dev = int(wt16.shape[1]), h * wd, x.device
        if gy is None:  # (only the non-differentiable statistics were used)
            return None, None, None, None
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        code = N.dtype_code(x)
        gx = gw = gb = None
        with torch.cuda.device(dev):
            sp = N.stream_ptr(dev)
            if ctx.needs_input_grad[0]:
                if cout % 8 == 0:
                    gx = torch.empty_like(x)
                    N.check(N.lib.ssdk_pw_forward(gy.data_ptr(), wt16.data_ptr(), None, gx.data_ptr(), b, cout, cin, hw, code, sp),
                            "pw_forward (input gradient)")
                else:  # (the kernel's K must be a multiple of 8; no such layer in the reference's networks)
                    gx = torch.matmul(wt16, gy.view(b, cout, hw)).view(b, cin, h, wd)
            if ctx.needs_input_grad[1]:
                need = int(N.lib.ssdk_pw_wgrad_workspace_bytes(b, cout, cin, hw))
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                gw32 = torch.empty((cout, cin, 1, 1), device=dev, dtype=torch.float32)
                N.check(N.lib.ssdk_pw_wgrad(gy.data_ptr(), x.data_ptr(), gw32.data_ptr(), ws.data_ptr(), need, b, cout, cin, hw, code,
                                            sp), "pw_wgrad")
                gw = gw32 if wdt == torch.float32 else gw32.to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3), dtype=torch.float32).to(bdt)
        return gx, gw, gb, None


def pointwise_conv(x, weight, bias=None, want_sums=False):
    """1x1 / stride 1 convolution of a contiguous NCHW tensor: weight [Cout,Cin,1,1], same floating dtype; differentiable.
    ``want_sums`` (16-bit HIP tensors only): the output carries ``_ssdk_bn_sums`` = [Cout, 2] (sum y, sum y^2) for the
    kernel-backed BatchNorm that follows (batchnorm.FastBatchNorm2d reads and consumes the attribute)."""
    if _native_ok(x, int(x.shape[1])):
        if want_sums:
            y, sums = _PointwiseNative.apply(x, weight, bias, True)
            y._ssdk_bn_sums = sums
            return y
        return _PointwiseNative.apply(x, weight, bias)
    if weight.dtype != x.dtype:
        weight = weight.to(x.dtype)
        bias = bias.to(x.dtype) if bias is not None else None
    return _Pointwise.apply(x, weight, bias)


BN_STATS_MIN_BYTES = 64 << 20  # output tensors from 64 MiB on hand their BatchNorm the statistics


class PointwiseConv2d(nn.Conv2d):
    _ssdk_bn_follows = False  # set by fuse_conv_bn_statistics: the next module is a kernel-backed BatchNorm in training mode

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
            x = x.to(dt)
            if not (_native_ok(x, self.in_channels) and w.dtype == torch.float32):
                w = w.to(dt)  # (the native kernels take the fp32 master weights themselves: no cast launch, fp32 gradient)
                bias = bias.to(dt) if bias is not None else None
        elif w.dtype != x.dtype:
            return super(PointwiseConv2d, self).forward(x)
        # the statistics ride on the convolution only where the BatchNorm's own pass over y costs more than they do: measured
        # (tools/pw_probe.py, round 6) + 7 ... 39 us on the kernel + two small reduce launches against y bytes / ~5 TB/s
        want = (self._ssdk_bn_follows and self.training
                and x.shape[0] * self.out_channels * x.shape[2] * x.shape[3] * 2 >= BN_STATS_MIN_BYTES)
        with torch.autocast("cuda", enabled=False):
            return pointwise_conv(x, w, bias, want_sums=want)


def fuse_conv_bn_statistics(model):
    """For every ``nn.Sequential`` of ``model`` in which a PointwiseConv2d or a DepthwiseConv2d is directly followed by a kernel-backed BatchNorm
    (batchnorm.FastBatchNorm2d): the convolution's forward kernel also produces the per-channel (sum, sum of squares) of its
    output and the BatchNorm starts from them -- one full pass over the activation less per Conv-BN pair (call after
    use_pointwise_gemm and use_fast_batchnorm; SSDK_BN_STATS_FUSED=0 keeps the BatchNorm's own reduction).  -> pairs found."""
    from ssds.modeling.layers.batchnorm import FastBatchNorm2d
    from ssds.modeling.layers.dwconv import DepthwiseConv2d

    n = 0
    if os.environ.get("SSDK_BN_STATS_FUSED", "1") == "0":
        return n
    for seq in model.modules():
        if not isinstance(seq, nn.Sequential):
            continue
        mods = list(seq.children())
        for conv, bn in zip(mods, mods[1:]):
            if type(conv) in (PointwiseConv2d, DepthwiseConv2d) and type(bn) is FastBatchNorm2d:
                conv._ssdk_bn_follows = True
                n += 1
    return n


def use_pointwise_gemm(model):
    """Switch every dense 1x1 / stride-1 ``nn.Conv2d`` of ``model`` to the GEMM-backed subclass (in place; no new
    parameters, same ``state_dict``)."""
    for m in model.modules():
        if (type(m) is nn.Conv2d and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
                and m.groups == 1 and m.dilation == (1, 1)):
            m.__class__ = PointwiseConv2d
    return model


# ---- dense 3x3 convolutions of the training step on the same kernels (csrc/ssdk_pwtrain.hip, second half) -----------------------
_COL_CACHE = {"x": None, "stride": 0, "col": None, "ver": -1}  # the loc and conf heads of a level read the SAME feature map (ssd.py:100-103)


def _im2col(x, stride, fold=False):
    """-> col [B, Kp, Ho * Wo], or FOLDED [1, Kp, B * Ho * Wo] (pixel index b * Ho * Wo + p: see ``FOLD_BELOW``)."""
    c = _COL_CACHE
    if c["x"] is x and c["stride"] == (stride, fold) and c["ver"] == x._version:
        return c["col"]
    b, cin, h, w = (int(v) for v in x.shape)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    kp = (cin * 9 + 7) // 8 * 8
    col = torch.empty((1, kp, b * ho * wo) if fold else (b, kp, ho * wo), device=x.device, dtype=x.dtype)
    fn = N.lib.ssdk_im2col3x3_folded if fold else N.lib.ssdk_im2col3x3
    N.check(fn(x.data_ptr(), col.data_ptr(), b, cin, h, w, stride, N.dtype_code(x), N.stream_ptr(x.device)), "im2col3x3")
    c.update(x=x, stride=(stride, fold), col=col, ver=x._version)
    return col


def release_col_cache():
    """Drop the one cached im2col tensor (it is kept alive until the next 3x3 layer runs otherwise)."""
    _COL_CACHE.update(x=None, col=None, stride=0, ver=-1)


FOLD_BELOW = 128  # output pixels per image below which a 3x3 layer's batch is folded into the GEMM's pixel dimension


def _fold(t):
    """[B, C, P] -> [1, C, B * P] (pixel index b * P + p): the 1x1 kernels tile PIXELS of one image in groups of 128; an extras
    layer has 64 / 16 / 4 / 1 pixels per image, i.e. groups that are 50 ... 0.8 % full, image by image."""
    b, c, p_ = (int(v) for v in t.shape)
    return t.permute(1, 0, 2).reshape(1, c, b * p_).contiguous()


def _unfold(t, b):
    """[1, C, B * P] -> [B, C, P]"""
    c, bp = int(t.shape[1]), int(t.shape[2])
    return t.view(c, b, bp // b).permute(1, 0, 2).contiguous()


class _Conv3x3Native(torch.autograd.Function):
    """3x3 / pad 1 / stride 1 | 2 on a 16-bit NCHW tensor: im2col + the 1x1 kernels (see csrc/ssdk_pwtrain.hip).  weight
    [Cout, Cin, 3, 3] fp32 (master parameter under autocast) or 16-bit; bias or None.  Small maps (< FOLD_BELOW output pixels per
    image: the extras) run as ONE image of B * Ho * Wo pixels (``_fold``)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride):
        b, cin, h, wd = (int(v) for v in x.shape)
        cout, dev, dt = int(w.shape[0]), x.device, x.dtype
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        code = N.dtype_code(x)
        x = x.detach()
        fold = b > 1 and ho * wo < FOLD_BELOW and (b * ho * wo) % 8 == 0
        with torch.cuda.device(dev):
            sp = N.stream_ptr(dev)
            col = _im2col(x, stride, fold)
            kp = int(col.shape[1])
            gb, ghw = (1, b * ho * wo) if fold else (b, ho * wo)
            w2 = w.detach().reshape(cout, cin * 9)
            if kp != cin * 9:
                w2 = torch.nn.functional.pad(w2, (0, kp - cin * 9))
            if w2.dtype == torch.float32:
                w16 = torch.empty((cout, kp), device=dev, dtype=dt)
                wt16 = torch.empty((kp, cout), device=dev, dtype=dt)
                N.check(N.lib.ssdk_pw_prepare(w2.contiguous().data_ptr(), w16.data_ptr(), wt16.data_ptr(), cout, kp, code, sp), "pw_prepare")
            else:
                w16 = w2.contiguous()
                wt16 = w16.t().contiguous()
            b32 = None if bias is None else bias.detach().float().contiguous()
            y = torch.empty((gb, cout, ghw), device=dev, dtype=dt)
            N.check(N.lib.ssdk_pw_forward(col.data_ptr(), w16.data_ptr(), None if b32 is None else b32.data_ptr(), y.data_ptr(),
                                          gb, kp, cout, ghw, code, sp), "pw_forward (3x3)")
            y = (_unfold(y, b) if fold else y).view(b, cout, ho, wo)
        ctx.save_for_backward(col, wt16)
        ctx.meta = (w.dtype, None if bias is None else bias.dtype, (b, cin, h, wd), stride, fold)
        return y

    @staticmethod
    def backward(ctx, gy):
        col, wt16 = ctx.saved_tensors
        wdt, bdt, (b, cin, h, wd), stride, fold = ctx.meta
        kp, cout = (int(v) for v in wt16.shape)
        gb, ghw, dev, dt = int(col.shape[0]), int(col.shape[2]), col.device, col.dtype
        gy = gy.contiguous()
        if gy.dtype != dt:
            gy = gy.to(dt)
        g3 = gy.view(b, cout, -1)
        gf = _fold(g3) if fold else g3  # [gb, cout, ghw]
        code = N.dtype_code(col)
        gx = gw = gbias = None
        with torch.cuda.device(dev):
            sp = N.stream_ptr(dev)
            if ctx.needs_input_grad[0]:
                if cout % 8 == 0:
                    dcol = torch.empty_like(col)
                    N.check(N.lib.ssdk_pw_forward(gf.data_ptr(), wt16.data_ptr(), None, dcol.data_ptr(), gb, cout, kp, ghw, code, sp),
                            "pw_forward (3x3 input gradient)")
                else:
                    dcol = torch.matmul(wt16, gf).contiguous()
                gx = torch.empty((b, cin, h, wd), device=dev, dtype=dt)
                fn = N.lib.ssdk_col2im3x3_folded if fold else N.lib.ssdk_col2im3x3
                N.check(fn(dcol.data_ptr(), gx.data_ptr(), b, cin, h, wd, stride, code, sp), "col2im3x3")
            if ctx.needs_input_grad[1]:
                need = int(N.lib.ssdk_pw_wgrad_workspace_bytes(gb, cout, kp, ghw))
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                gw32 = torch.empty((cout, kp), device=dev, dtype=torch.float32)
                N.check(N.lib.ssdk_pw_wgrad(gf.data_ptr(), col.data_ptr(), gw32.data_ptr(), ws.data_ptr(), need, gb, cout, kp, ghw, code, sp),
                        "pw_wgrad (3x3)")
                gw = gw32[:, : cin * 9].reshape(cout, cin, 3, 3)
                gw = gw if wdt == torch.float32 else gw.to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            gbias = gy.sum((0, 2, 3), dtype=torch.float32).to(bdt)
        return gx, gw, gbias, None


class NativeConv3x3(nn.Conv2d):
    """``nn.Conv2d(k = 3, pad = 1, stride 1 | 2)`` whose 16-bit HIP-device forward / backward run on the ssdk kernels (same
    parameters, ``state_dict`` keys and initialisation); everything else is ``nn.Conv2d.forward``."""

    def _native(self, x):
        return (x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1)
                and self.stride in ((1, 1), (2, 2)) and self.groups == 1 and self.padding_mode == "zeros" and x.is_contiguous()
                and os.environ.get("SSDK_CONV3_NATIVE", "1") != "0")

    def forward(self, x):
        if not self._native(x):
            return super(NativeConv3x3, self).forward(x)
        w, bias = self.weight, self.bias
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
        if x.dtype not in (torch.bfloat16, torch.float16) or (w.dtype != torch.float32 and w.dtype != x.dtype):
            return super(NativeConv3x3, self).forward(x)
        with torch.autocast("cuda", enabled=False):
            return _Conv3x3Native.apply(x, w, bias, self.stride[0])


def use_native_conv3x3(model, min_in_channels=0):
    """Switch every dense 3x3 / pad 1 / stride 1 | 2 ``nn.Conv2d`` of ``model`` with at least ``min_in_channels`` input channels
    to the kernel-backed subclass (in place)."""
    for m in model.modules():
        if (type(m) is nn.Conv2d and m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride in ((1, 1), (2, 2))
                and m.groups == 1 and m.dilation == (1, 1) and m.padding_mode == "zeros" and m.in_channels >= min_in_channels):
            m.__class__ = NativeConv3x3
    return model


# ---- the network's first convolution (3x3 / stride 2 / pad 1 on the image) on its own kernels (csrc/ssdk_stemtrain.hip) ------------
class _StemConv3x3s2(torch.autograd.Function):
    """x [N, Cin <= 3, H, W] 16 bit (contiguous), w [Cout <= 32, Cin, 3, 3] fp32 master parameter -> y [N, Cout, Ho, Wo] 16 bit.
    Backward: the weight gradient (fp32, matrix cores over pixels, fixed-order partial sums); the image's gradient, if anybody
    asks for it, comes from the framework's convolution backward."""

    @staticmethod
    def forward(ctx, x, w):
        n, cin, h, wd = (int(v) for v in x.shape)
        cout, dev = int(w.shape[0]), x.device
        ho, wo = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
        x = x.detach()
        y = torch.empty((n, cout, ho, wo), device=dev, dtype=x.dtype)
        with torch.cuda.device(dev):
            N.check(N.lib.ssdk_stem3x3s2_fwd(x.data_ptr(), w.detach().contiguous().data_ptr(), y.data_ptr(), n, cin, h, wd, cout,
                                             N.dtype_code(x), N.stream_ptr(dev)), "stem3x3s2_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        n, cin, h, wd = (int(v) for v in x.shape)
        cout, dev = int(w.shape[0]), x.device
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = gw = None
        if ctx.needs_input_grad[1] and wd % 16 != 0:
            # rows that are not whole 16-byte groups (the 300 px configuration): the kernel's element-wise operand path is slower
            # than the library's weight gradient (tools/run/r06_s43.sh) -- only the forward is ours there
            gw = torch.ops.aten.convolution_backward(gy, x, w.to(x.dtype), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1].to(w.dtype)
        elif ctx.needs_input_grad[1]:
            need = int(N.lib.ssdk_stem3x3s2_wgrad_workspace_bytes(n, h))
            ws = torch.empty(need + 16, dtype=torch.uint8, device=dev)
            wp = (ws.data_ptr() + 15) & ~15
            gw = torch.empty((cout, cin, 3, 3), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                N.check(N.lib.ssdk_stem3x3s2_wgrad(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), wp, need, n, cin, h, wd, cout,
                                                   N.dtype_code(x), N.stream_ptr(dev)), "stem3x3s2_wgrad")
            gw = gw if w.dtype == torch.float32 else gw.to(w.dtype)
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w.to(x.dtype), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        return gx, gw


class StemConv3x3s2(nn.Conv2d):
    """``nn.Conv2d(Cin <= 3, Cout <= 32, k = 3, stride 2, pad 1, bias = False)`` -- the first layer of the MobileNet backbones --
    whose 16-bit HIP-device forward / weight gradient run on ``csrc/ssdk_stemtrain.hip`` (same parameter, ``state_dict`` key and
    initialisation); everything else is ``nn.Conv2d.forward``."""

    def _native(self, x):
        return (x.is_cuda and x.dim() == 4 and self.kernel_size == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1)
                and self.stride == (2, 2) and self.groups == 1 and self.padding_mode == "zeros" and self.bias is None
                and self.in_channels <= 3 and self.out_channels <= 32 and int(x.shape[1]) == self.in_channels)

    def forward(self, x):
        if not self._native(x):
            return super(StemConv3x3s2, self).forward(x)
        w = self.weight
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
        if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != torch.float32:
            return super(StemConv3x3s2, self).forward(x)
        with torch.autocast("cuda", enabled=False):
            return _StemConv3x3s2.apply(x.contiguous(), w)


def use_native_stem(model):
    """Switch the image-side 3x3 / stride-2 convolutions of ``model`` (<= 3 input channels, <= 32 filters, no bias) to the
    kernel-backed subclass (in place).  -> layers switched."""
    n = 0
    for m in model.modules():
        if (type(m) is nn.Conv2d and m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (2, 2) and m.groups == 1
                and m.dilation == (1, 1) and m.padding_mode == "zeros" and m.bias is None and m.in_channels <= 3
                and m.out_channels <= 32):
            m.__class__ = StemConv3x3s2
            n += 1
    return n
