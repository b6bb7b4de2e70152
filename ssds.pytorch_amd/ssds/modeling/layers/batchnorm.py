"""``nn.BatchNorm2d`` whose TRAINING forward / backward on a HIP device run on the ssdk kernels
(``csrc/ssdk_bntrain.hip``: two HBM-bound passes each, fixed-order reductions) instead of
``MIOpenBatchNorm{Fwd,Bwd}Spatial`` -- the largest item of the reference's DDP training step on SSD-MobileNetV2
once the depthwise convolutions are off MIOpen's naive kernels.  Same parameters, buffers, ``state_dict`` and
running-statistics semantics (momentum, unbiased variance, ``num_batches_tracked``); eval mode, CPU tensors and
non-affine / non-tracking variants are plain ``nn.BatchNorm2d``.

``fuse_bn_activations`` additionally folds the ReLU6 / ReLU that follows a BatchNorm inside an ``nn.Sequential``
(every Conv-BN-ReLU6 of the backbone, mobilenet.py:24-33) into those kernels: the clamp rides on the forward apply
pass and its gradient mask on the two backward passes, so the ``clamp`` / ``hardtanh_backward`` launches and one
read + write of each activation tensor per direction disappear.  The activation module stays in the ``Sequential``
(same ``state_dict``, same module list) and simply lets a tensor through that already carries its clamp."""
import torch
import torch.nn as nn

from ssds import _native as N


def _ws(dev, n, c):
    need = int(N.lib.ssdk_bn_workspace_bytes(n, c))
    return torch.empty(need + 16, dtype=torch.uint8, device=dev), need


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act=0, sums=None):
        """``sums``: [C, 2] fp32 (sum x, sum x^2) over N * H * W from the kernel that produced x (pointwise.pointwise_conv with
        want_sums): the forward pass then has no reduction of its own (ssdk_bn_act_train_fwd_sums)."""
        x = x.contiguous()
        n, c = int(x.shape[0]), int(x.shape[1])
        hw = int(x.shape[2]) * int(x.shape[3])
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(c, device=dev, dtype=torch.float32)
        invstd = torch.empty(c, device=dev, dtype=torch.float32)
        ws, need = _ws(dev, n, c)
        wp = (ws.data_ptr() + 15) & ~15
        with torch.cuda.device(dev):
            if sums is not None:
                N.check(N.lib.ssdk_bn_act_train_fwd_sums(x.data_ptr(), sums.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                         running_mean.data_ptr(), running_var.data_ptr(), y.data_ptr(),
                                                         mean.data_ptr(), invstd.data_ptr(), wp, need, n, c, hw, float(momentum),
                                                         float(eps), int(act), N.dtype_code(x), N.stream_ptr(dev)),
                        "bn_train_fwd_sums")
            else:
                N.check(N.lib.ssdk_bn_act_train_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(),
                                                    running_var.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), wp,
                                                    need, n, c, hw, float(momentum), float(eps), int(act), N.dtype_code(x),
                                                    N.stream_ptr(dev)), "bn_train_fwd")
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.act = int(act)
        ctx.mark_non_differentiable(running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        n, c = int(x.shape[0]), int(x.shape[1])
        hw = int(x.shape[2]) * int(x.shape[3])
        dev = x.device
        gx = torch.empty_like(x)
        gw = torch.empty(c, device=dev, dtype=torch.float32)
        gb = torch.empty(c, device=dev, dtype=torch.float32)
        ws, need = _ws(dev, n, c)
        wp = (ws.data_ptr() + 15) & ~15
        with torch.cuda.device(dev):
            N.check(N.lib.ssdk_bn_act_train_bwd(x.data_ptr(), gy.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                mean.data_ptr(), invstd.data_ptr(), gx.data_ptr(), gw.data_ptr(),
                                                gb.data_ptr(), wp, need, n, c, hw, ctx.act, N.dtype_code(x),
                                                N.stream_ptr(dev)), "bn_train_bwd")
        return gx, gw.to(weight.dtype), gb.to(weight.dtype), None, None, None, None, None, None


class _BatchNormDeferred(torch.autograd.Function):
    """Training BatchNorm whose OUTPUT is never written: the forward runs the statistics + coefficients only
    (ssdk_bn_act_train_stats) and returns an ALIAS of x together with coef [C, 4]; the depthwise convolution behind it applies
    act(a x + b) while it stages its input (dwconv._DwConv3x3 with ``coef``).  The backward pass is _BatchNormTrain's."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, sums):
        x = x.contiguous()
        n, c = int(x.shape[0]), int(x.shape[1])
        hw = int(x.shape[2]) * int(x.shape[3])
        dev = x.device
        mean = torch.empty(c, device=dev, dtype=torch.float32)
        invstd = torch.empty(c, device=dev, dtype=torch.float32)
        coef = torch.empty((c, 4), device=dev, dtype=torch.float32)
        ws, need = _ws(dev, n, c)
        wp = (ws.data_ptr() + 15) & ~15
        with torch.cuda.device(dev):
            N.check(N.lib.ssdk_bn_act_train_stats(x.data_ptr(), None if sums is None else sums.data_ptr(), weight.data_ptr(),
                                                  bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), mean.data_ptr(),
                                                  invstd.data_ptr(), coef.data_ptr(), wp, need, n, c, hw, float(momentum), float(eps),
                                                  N.dtype_code(x), N.stream_ptr(dev)), "bn_train_stats")
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.act = int(act)
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)  # (no zero-filled gradient tensor for coef: a fill launch per layer and step)
        return x.detach(), coef  # (an alias: what the consumer reads through it is x, NOT the BatchNorm's output)

    @staticmethod
    def backward(ctx, gy, _gcoef=None):
        if gy is None:
            return (None,) * 9
        return _BatchNormTrain.backward(ctx, gy)


class FastBatchNorm2d(nn.BatchNorm2d):
    _ssdk_act = 0  # 1 ReLU6 | 2 ReLU folded into the kernels (set per instance by fuse_bn_activations)
    _ssdk_counter_external = False  # True: somebody bumps num_batches_tracked for ALL layers in one launch (bump_counters)

    def forward(self, x):
        if not (self.training and x.is_cuda and x.dim() == 4 and self.affine and self.track_running_stats
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and self.weight.dtype == torch.float32):
            return super(FastBatchNorm2d, self).forward(x)
        if not self._ssdk_counter_external:
            self.num_batches_tracked.add_(1)  # nn.BatchNorm2d bookkeeping (batchnorm.py of torch)
        momentum = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
        sums = x.__dict__.pop("_ssdk_bn_sums", None) if hasattr(x, "__dict__") else None  # (from the producing 1x1 convolution)
        if sums is not None and (tuple(sums.shape) != (x.shape[1], 2) or not x.is_contiguous()):
            sums = None
        dw = self.__dict__.get("_ssdk_defer_to")  # the depthwise convolution that reads this BatchNorm's output (fuse_bn_into_depthwise)
        if (dw is not None and dw.training and self._ssdk_act and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
                and (not torch.is_autocast_enabled() or torch.get_autocast_dtype("cuda") == x.dtype)
                and N.lib.ssdk_dwconv_affine_supported(int(x.shape[0]), int(x.shape[1]), int(x.shape[2]), int(x.shape[3]),
                                                       int(dw.stride[0]), N.dtype_code(x))):
            with torch.autocast("cuda", enabled=False):
                y, coef = _BatchNormDeferred.apply(x, self.weight, self.bias, self.running_mean, self.running_var, momentum,
                                                   self.eps, self._ssdk_act, sums)
            y._ssdk_pending_bn = (coef, self._ssdk_act)  # consumed (and checked) by DepthwiseConv2d.forward
            y._ssdk_act_applied = self._ssdk_act
            dw._ssdk_expect_pending = True
            return y
        with torch.autocast("cuda", enabled=False):
            y = _BatchNormTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, momentum, self.eps,
                                      self._ssdk_act, sums)
        if self._ssdk_act:
            y._ssdk_act_applied = self._ssdk_act  # read by the activation module that follows (and by nothing else)
        return y


def bump_counters(model):
    """``num_batches_tracked += 1`` for every kernel-backed BatchNorm of ``model`` in ONE multi-tensor launch, instead of one
    4.5 us launch per layer inside each forward (59 layers in SSD-MobileNetV2: 0.27 ms of a 20 ms step).  The caller owns the
    bookkeeping for THIS forward: call it right before a training forward and ``release_counters`` behind it
    (pipeline_anchor_ddp.ModelWithLossBasic does).  -> the layers that were bumped."""
    cache = model.__dict__.get("_ssdk_bn_counters")
    if cache is None:
        cache = [m for m in model.modules() if type(m) is FastBatchNorm2d and m.track_running_stats]
        model.__dict__["_ssdk_bn_counters"] = cache
    live = [m for m in cache if m.training]
    for m in live:
        m._ssdk_counter_external = True
    if live:
        torch._foreach_add_([m.num_batches_tracked for m in live], 1)
    return live


def release_counters(live):
    """The layers bump their own counter again (a forward outside the training module: BatchNorm calibration, tests)."""
    for m in live:
        m._ssdk_counter_external = False


def use_fast_batchnorm(model):
    """Switch every plain ``nn.BatchNorm2d`` of ``model`` to the kernel-backed subclass (in place; no new parameters)."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.__class__ = FastBatchNorm2d
    return model


class _FusedAwayActivation(object):
    """Mixin of the activation that follows a fused BatchNorm: a tensor that already carries the clamp passes through."""

    def forward(self, x):
        if getattr(x, "_ssdk_act_applied", 0) == self._ssdk_act_code:
            return x
        return super(_FusedAwayActivation, self).forward(x)


class FusedAwayReLU6(_FusedAwayActivation, nn.ReLU6):
    _ssdk_act_code = 1


class FusedAwayReLU(_FusedAwayActivation, nn.ReLU):
    _ssdk_act_code = 2


def fuse_bn_into_depthwise(model):
    """For every pair of neighbouring ``nn.Sequential`` blocks [..., FastBatchNorm2d (+ folded ReLU6 / ReLU), activation] ->
    [DepthwiseConv2d, ...] inside one ``nn.Sequential`` (the expand and depthwise ConvBNReLU blocks of an inverted-residual
    block, mobilenet.py:56): the BatchNorm's apply pass disappears -- it computes statistics and coefficients only, and the depthwise
    forward / weight-gradient kernels normalise x while they stage it (the 6 x expanded tensor is written once, by the 1x1
    convolution, instead of twice).  Call after use_fast_batchnorm + fuse_bn_activations; SSDK_BN_DEFER=0 keeps the apply pass.
    -> pairs found."""
    import os

    from ssds.modeling.layers.dwconv import DepthwiseConv2d

    n = 0
    if os.environ.get("SSDK_BN_DEFER", "1") == "0":
        return n
    for seq in model.modules():
        if not isinstance(seq, nn.Sequential):
            continue
        kids = list(seq.children())
        for a, b in zip(kids, kids[1:]):
            if not (isinstance(a, nn.Sequential) and isinstance(b, nn.Sequential)):
                continue
            ak, bk = list(a.children()), list(b.children())
            if (len(ak) >= 2 and len(bk) >= 1 and type(ak[-2]) is FastBatchNorm2d and ak[-2]._ssdk_act
                    and isinstance(ak[-1], _FusedAwayActivation) and type(bk[0]) is DepthwiseConv2d):
                ak[-2].__dict__["_ssdk_defer_to"] = bk[0]  # (not a registered submodule: no second path to its parameters)
                n += 1
    return n


def fuse_bn_activations(model):
    """For every ``nn.Sequential`` of ``model`` in which a kernel-backed BatchNorm is directly followed by ``nn.ReLU6`` /
    ``nn.ReLU``: fold the activation into the BatchNorm kernels (in place; call after ``use_fast_batchnorm``).  Whenever
    the BatchNorm takes its plain path (eval mode, CPU, ...) the activation module runs as before."""
    n = 0
    for seq in model.modules():
        if not isinstance(seq, nn.Sequential):
            continue
        mods = list(seq.children())
        for bn, act in zip(mods, mods[1:]):
            if type(bn) is not FastBatchNorm2d:
                continue
            if type(act) is nn.ReLU6:
                bn._ssdk_act, act.__class__ = 1, FusedAwayReLU6
            elif type(act) is nn.ReLU:
                bn._ssdk_act, act.__class__ = 2, FusedAwayReLU
            else:
                continue
            n += 1
    return n
