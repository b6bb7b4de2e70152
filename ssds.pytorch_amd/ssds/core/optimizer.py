"""Optimizer plumbing of the training entry points (reference ``ssds/core/optimizer.py``):
``trainable_param`` (:28-70) turns cfg.TRAIN.TRAINABLE_SCOPE ("a,b;c.d": ',' joins modules of one
parameter group, ';' separates groups with differential learning rates) into parameter lists and sets
``requires_grad``; ``configure_optimizer`` (:73-134) and ``configure_lr_scheduler`` (:137-166) map the cfg
names to torch.optim objects."""
import torch.optim as optim
from torch.optim import lr_scheduler


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
        # On a HIP device the update runs as ONE fused multi-tensor launch, and the fused kernels take a device-side
        # ``found_inf`` flag: pipeline_anchor_ddp.train_step skips a step on NaN/Inf without reading the flag back
        # (the reference syncs 4-6 times per step, pipeline_anchor_apex.py:114-126).
        flat = params if not isinstance(params[0], dict) else [q for g in params for q in g["params"]]
        fused = len(flat) > 0 and all(q.is_cuda and q.is_floating_point() for q in flat)
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
