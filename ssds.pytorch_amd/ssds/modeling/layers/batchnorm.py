"""``nn.BatchNorm2d`` whose TRAINING forward / backward on a HIP device run on the ssdk kernels
(``csrc/ssdk_bntrain.hip``: two HBM-bound passes each, fixed-order reductions) instead of
``MIOpenBatchNorm{Fwd,Bwd}Spatial`` -- the largest item of the reference's DDP training step on SSD-MobileNetV2
once the depthwise convolutions are off MIOpen's naive kernels.  Same parameters, buffers, ``state_dict`` and
running-statistics semantics (momentum, unbiased variance, ``num_batches_tracked``); eval mode, CPU tensors and
non-affine / non-tracking variants are plain ``nn.BatchNorm2d``."""
import torch
import torch.nn as nn

from ssds import _native as N


def _ws(dev, n, c):
    need = int(N.lib.ssdk_bn_workspace_bytes(n, c))
    return torch.empty(need + 16, dtype=torch.uint8, device=dev), need


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
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
            N.check(N.lib.ssdk_bn_train_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(),
                                            running_var.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), wp,
                                            need, n, c, hw, float(momentum), float(eps), N.dtype_code(x),
                                            N.stream_ptr(dev)), "bn_train_fwd")
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.mark_non_differentiable(running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, invstd = ctx.saved_tensors
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
            N.check(N.lib.ssdk_bn_train_bwd(x.data_ptr(), gy.data_ptr(), weight.data_ptr(), mean.data_ptr(),
                                            invstd.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(), wp, need, n, c,
                                            hw, N.dtype_code(x), N.stream_ptr(dev)), "bn_train_bwd")
        return gx, gw.to(weight.dtype), gb.to(weight.dtype), None, None, None, None


class FastBatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        if not (self.training and x.is_cuda and x.dim() == 4 and self.affine and self.track_running_stats
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and self.weight.dtype == torch.float32):
            return super(FastBatchNorm2d, self).forward(x)
        self.num_batches_tracked.add_(1)  # nn.BatchNorm2d bookkeeping (batchnorm.py of torch)
        momentum = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
        with torch.autocast("cuda", enabled=False):
            return _BatchNormTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, momentum, self.eps)


def use_fast_batchnorm(model):
    """Switch every plain ``nn.BatchNorm2d`` of ``model`` to the kernel-backed subclass (in place; no new parameters)."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.__class__ = FastBatchNorm2d
    return model
