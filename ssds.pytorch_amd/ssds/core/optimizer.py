"""Optimizer plumbing of the training entry points (reference ``ssds/core/optimizer.py``):
``trainable_param`` (:28-70) turns cfg.TRAIN.TRAINABLE_SCOPE ("a,b;c.d": ',' joins modules of one
parameter group, ';' separates groups with differential learning rates) into parameter lists and sets
``requires_grad``; ``configure_optimizer`` (:73-134) and ``configure_lr_scheduler`` (:137-166) map the cfg
names to torch.optim objects."""
import ctypes
import os

import torch
import torch.optim as optim
from torch.optim import lr_scheduler


class SsdkSGD(optim.Optimizer):
    """``torch.optim.SGD`` (momentum, weight decay, Nesterov; dampening 0) whose ``step()`` is csrc/ssdk_sgd.hip: every fp32
    parameter tensor of a group in a handful of launches (``ssdk_sgd_step``), no host synchronisation.  Same ``param_groups``
    keys and the same ``state[p]["momentum_buffer"]`` as torch.optim.SGD, so ``state_dict()`` / ``load_state_dict()`` and the
    lr schedulers are interchangeable with it (reference: core/optimizer.py:73-134 builds torch.optim.SGD).

    The NaN/Inf skip of the reference's loop (pipeline_anchor_apex.py:110-111, 126-127) is taken on the device: when the
    attribute ``found_inf`` (a 1-element float tensor, non-zero = skip -- the hook torch's GradScaler uses on fused optimizers)
    is set, the kernels read it and leave every tensor untouched.  A learning rate that is a device TENSOR is read by the kernel
    (``lr_dev``): a captured hipGraph keeps a live learning rate (pipeline_anchor_ddp.GraphedTrainStep).

    Parameters that are not fp32 HIP tensors take torch's own single-tensor update rule in Python (tests on the CPU)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=nesterov,
                        maximize=False, foreach=None, differentiable=False, fused=True)
        super(SsdkSGD, self).__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        from ssds import _native as N

        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, "found_inf", None)
        for group in self.param_groups:
            lr, mom, wd, nest = group["lr"], float(group["momentum"]), float(group["weight_decay"]), bool(group["nesterov"])
            native, other = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if mom != 0 and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ok = (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous()
                      and p.grad.is_contiguous() and not p.grad.is_sparse)
                (native if ok else other).append(p)
            if native:
                dev = native[0].device
                n = len(native)
                arr = ctypes.c_void_p * n
                ps = arr(*[p.data_ptr() for p in native])
                gs = arr(*[p.grad.data_ptr() for p in native])
                ms = arr(*[self.state[p]["momentum_buffer"].data_ptr() for p in native]) if mom != 0 else None
                ne = (ctypes.c_int64 * n)(*[p.numel() for p in native])
                lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) and lr.is_cuda else None
                lr_val = 0.0 if lr_dev is not None else float(lr)
                fi = None
                if found_inf is not None:
                    fi = found_inf if (found_inf.is_cuda and found_inf.dtype == torch.float32) else found_inf.to(dev, torch.float32)
                    self._found_inf_keepalive = fi
                with torch.cuda.device(dev):
                    N.check(N.lib.ssdk_sgd_step(n, ps, gs, ms, ne, lr_dev, lr_val, mom, wd, 1 if nest else 0,
                                                None if fi is None else fi.data_ptr(), N.stream_ptr(dev)), "sgd_step")
            if other:
                if found_inf is not None and bool(found_inf.item() > 0):
                    continue
                for p in other:
                    g = p.grad if wd == 0 else p.grad.add(p, alpha=wd)
                    if mom != 0:
                        buf = self.state[p]["momentum_buffer"]
                        buf.mul_(mom).add_(g)
                        g = g.add(buf, alpha=mom) if nest else buf
                    p.add_(g.to(p.dtype), alpha=-float(lr))
        return loss


def _resolve(model, dotted):
    m = model
    for part in dotted.split("."):
        if not hasattr(m, part):
            raise ValueError(dotted + " is not in the model")
        m = getattr(m, part)
    return m


def trainable_param(model, trainable_scope):
    if trainable_scope == "":
        for p in model.parameters():
            p.requires_grad = True
        return [list(model.parameters())]
    for p in model.parameters():
        p.requires_grad = False
    groups = []
    for scope in trainable_scope.split(";"):
        params = []
        for name in scope.split(","):
            sub = _resolve(model, name)
            for p in sub.parameters():
                p.requires_grad = True
            params.extend(sub.parameters())
        groups.append(params)
    return groups


def configure_optimizer(trainable_param, cfg):
    if len(cfg.DIFFERENTIAL_LEARNING_RATE) == 0 or len(trainable_param) == 1:
        params = trainable_param[0]
    else:
        assert len(cfg.DIFFERENTIAL_LEARNING_RATE) == len(trainable_param)
        params = [{"params": p, "lr": lr} for p, lr in zip(trainable_param, cfg.DIFFERENTIAL_LEARNING_RATE)]
    name = cfg.OPTIMIZER
    if name == "sgd":
        # On a HIP device the update runs as a few multi-tensor launches (SsdkSGD; SSDK_SGD_NATIVE=0: torch's fused SGD), and
        # the kernels take a device-side ``found_inf`` flag: pipeline_anchor_ddp.train_step skips a step on NaN/Inf without reading the flag back
        # (the reference syncs 4-6 times per step, pipeline_anchor_apex.py:114-126).
        flat = params if not isinstance(params[0], dict) else [q for g in params for q in g["params"]]
        fused = len(flat) > 0 and all(q.is_cuda and q.is_floating_point() for q in flat)
        if fused and os.environ.get("SSDK_SGD_NATIVE", "1") != "0":  # round 6: the update on csrc/ssdk_sgd.hip
            return SsdkSGD(params, lr=cfg.LEARNING_RATE, momentum=cfg.MOMENTUM, weight_decay=cfg.WEIGHT_DECAY)
        return optim.SGD(params, lr=cfg.LEARNING_RATE, momentum=cfg.MOMENTUM, weight_decay=cfg.WEIGHT_DECAY,
                         **({"fused": True} if fused else {}))
    if name == "rmsprop":
        return optim.RMSprop(params, lr=cfg.LEARNING_RATE, momentum=cfg.MOMENTUM, alpha=cfg.MOMENTUM_2,
                             eps=cfg.EPS, weight_decay=cfg.WEIGHT_DECAY)
    if name in ("adam", "amsgrad"):
        return optim.Adam(params, lr=cfg.LEARNING_RATE, betas=(cfg.MOMENTUM, cfg.MOMENTUM_2),
                          weight_decay=cfg.WEIGHT_DECAY, amsgrad=(name == "amsgrad"))
    raise AssertionError("optimizer can not be recognized")


def configure_lr_scheduler(optimizer, cfg):
    name = cfg.SCHEDULER
    if name == "step":
        return lr_scheduler.StepLR(optimizer, step_size=cfg.STEPS[0], gamma=cfg.GAMMA)
    if name == "multi_step":
        return lr_scheduler.MultiStepLR(optimizer, milestones=cfg.STEPS, gamma=cfg.GAMMA)
    if name == "exponential":
        return lr_scheduler.ExponentialLR(optimizer, gamma=cfg.GAMMA)
    if name == "sgdr":
        return lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=2, T_mult=2, eta_min=cfg.LR_MIN)
    raise AssertionError("scheduler can not be recognized.")
