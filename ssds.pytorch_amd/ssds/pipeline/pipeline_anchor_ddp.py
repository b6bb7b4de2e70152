"""Training step of the anchor-based detectors under data parallelism -- the step body of the reference's
``ssds/pipeline/pipeline_anchor_apex.py`` (ModelWithLossBasic :16-72, train loop :75-171) without Apex:

    loc, conf = model(images)                      (bf16 autocast on the HIP device)
    per level: extract_targets (ONE HIP launch for the whole batch), focal + smooth-L1, masks, sums
    loss / sum_levels clamp(#foreground, 1)        (LOCAL normaliser, like the reference :69-71)
    backward  -> bucketed RCCL all-reduce (mean) overlapped with backward (torch DDP)
    optimizer.step()

The reference skips a step on NaN/Inf with a per-rank ``continue`` before backward (:110-111, 126-127),
which under DDP would leave the other ranks waiting in the all-reduce; here the decision is collective
(one all-reduced flag) and taken after backward so every rank always joins the gradient all-reduce."""
import os
import time

import torch
import torch.distributed as dist

from ssds.core import criterion as _crit
from ssds.modeling.layers import box


class ModelWithLossBasic(torch.nn.Module):
    r"""model + target assignment + losses in one module so that DDP wraps the whole step
    (reference pipeline_anchor_apex.py:16-72).  forward(images, targets, anchors) ->
    (cls_loss, loc_loss, [cls_loss per level], [loc_loss per level])."""

    def __init__(self, model, cls_criterion, loc_criterion, num_classes, match, center_sampling_radius):
        super(ModelWithLossBasic, self).__init__()
        self.model = model
        self.cls_criterion = cls_criterion
        self.loc_criterion = loc_criterion
        self.num_classes = num_classes
        self.match = match
        self.center_radius = center_sampling_radius

    def _fused(self, conf):
        """The fused target-assign + loss kernels apply when the criteria are exactly the reference's FocalLoss or
        MultiBoxLoss and SmoothL1Loss or IOULoss (iou / giou / diou / ciou) and the heads live on a HIP device
        (SSDK_FUSED_LOSS=0 keeps the unfused torch ops)."""
        return (conf[0].is_cuda and type(self.cls_criterion) in (_crit.FocalLoss, _crit.MultiBoxLoss)
                and type(self.loc_criterion) in (_crit.SmoothL1Loss, _crit.IOULoss)
                and os.environ.get("SSDK_FUSED_LOSS", "1") != "0")

    def forward(self, images, targets, anchors):
        if self.training and getattr(images, "is_cuda", False):
            from ssds.modeling.layers.batchnorm import bump_counters, release_counters

            live = bump_counters(self.model)  # num_batches_tracked of every kernel-backed BatchNorm: one launch, not one per layer
            try:
                loc, conf = self.model(images)
            finally:
                release_counters(live)
        else:
            loc, conf = self.model(images)
        cls_losses, loc_losses, fg_targets = [], [], []
        fused = self._fused(conf)
        if fused:
            from ssds.core.fused_loss import match_loss
        for j, (stride, anchor) in enumerate(anchors.items()):
            if fused:  # one launch: match + focal + smooth-L1 + masks + sums, gradients written in the same pass
                lc, cc = self.loc_criterion, self.cls_criterion
                cls_sum, loc_sum, fg = match_loss(
                    conf[j], loc[j], targets, anchors, self.num_classes, stride, self.match, self.center_radius,
                    getattr(cc, "alpha", 0.25), getattr(cc, "gamma", 2.0), getattr(lc, "beta", 0.11),
                    getattr(lc, "loss_type", "smoothl1"), getattr(cc, "negpos_ratio", None))
                fg_targets.append(fg.clamp(min=1))
                cls_losses.append(cls_sum)
                loc_losses.append(loc_sum)
                continue
            size = conf[j].shape[-2:]
            with torch.no_grad():
                conf_target, loc_target, depth = box.extract_targets(
                    targets, anchors, self.num_classes, stride, size, self.match, self.center_radius)
            fg_targets.append((depth > 0).sum().float().clamp(min=1))

            c = conf[j].view_as(conf_target).float()
            cls_mask = (depth >= 0).expand_as(conf_target).float()
            cls_loss = self.cls_criterion(c, conf_target, depth)
            cls_losses.append((cls_mask * cls_loss).sum())

            l = loc[j].view_as(loc_target).float()
            loc_loss = self.loc_criterion(l, loc_target)
            loc_mask = (depth > 0).expand_as(loc_loss).float()
            loc_losses.append((loc_mask * loc_loss).sum())

        fg_targets = torch.stack(fg_targets).sum()
        cls_loss = torch.stack(cls_losses).sum() / fg_targets
        loc_loss = torch.stack(loc_losses).sum() / fg_targets
        return cls_loss, loc_loss, cls_losses, loc_losses


def train_step(model_with_loss, images, targets, anchors, optimizer, autocast_dtype=torch.bfloat16):
    """One optimisation step.  Returns (cls_loss, loc_loss, skipped): the losses stay on the device; with the fused
    optimizer of a HIP device the skip decision does too (no host synchronisation; the reference syncs 4-6 times per step,
    pipeline_anchor_apex.py:114-126), otherwise the collective skip flag is the one value read back."""
    dev = images.device
    optimizer.zero_grad(set_to_none=True)
    use_autocast = dev.type == "cuda" and autocast_dtype is not None
    with torch.autocast(device_type=dev.type, dtype=autocast_dtype, enabled=use_autocast):
        cls_loss, loc_loss, _, _ = model_with_loss(images, targets, anchors)
        total = cls_loss + loc_loss
    bad = (~torch.isfinite(total.detach())).float()
    # every rank runs backward so that the bucketed gradient all-reduce is always matched; a non-finite
    # loss is replaced by zero gradients on that rank only if NOBODY may step
    torch.nan_to_num(total, nan=0.0, posinf=0.0, neginf=0.0).backward()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)  # collective decision
    skipped = _step_unless(optimizer, bad)
    return cls_loss.detach(), loc_loss.detach(), skipped


def _device_skip(optimizer):
    """True when every parameter group runs torch's fused multi-tensor kernels, which take the skip flag on the device."""
    return len(optimizer.param_groups) > 0 and all(g.get("fused") for g in optimizer.param_groups)


def _ensure_momentum_buffers(optimizer, every=False):
    """Zero momentum buffers for every parameter that has none yet.  torch's fused SGD allocates them with ``empty_like``
    inside its first step and returns early when ``found_inf`` is set: a skipped FIRST step would leave uninitialised
    memory behind as momentum.  With zero buffers in place the first real step computes ``0 * momentum + grad`` -- exactly
    the first-step rule (dampening is 0 in core/optimizer.configure_optimizer)."""
    from ssds.core.optimizer import SsdkSGD

    if isinstance(optimizer, SsdkSGD) and not every:
        return  # (csrc/ssdk_sgd.hip's host side creates missing buffers as zeros itself, skipped step or not)
    for group in optimizer.param_groups:
        if not group.get("momentum"):
            continue
        # per step: only parameters that take part in it (not every requires_grad one); ``every``: all trainable
        # parameters (GraphedTrainStep: the state must exist BEFORE the warm-up so that the snapshot can restore zeros)
        for p in group["params"]:
            if (p.requires_grad if every else p.grad is not None) and "momentum_buffer" not in optimizer.state[p]:
                optimizer.state[p]["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)


def _step_unless(optimizer, bad):
    """optimizer.step() unless ``bad`` (a 0/1 float tensor) is set.  Fused optimizers (core/optimizer.py on a HIP device) read
    the flag on the DEVICE (``found_inf``, the hook GradScaler uses): no host synchronisation in the step at all, and the step
    can be captured in a hipGraph.  Anything else reads the flag back -- the one host sync of the step.  Returns the flag (a
    tensor on the fused path: it is only converted where somebody prints it)."""
    if _device_skip(optimizer):
        if not torch.cuda.is_current_stream_capturing():
            _ensure_momentum_buffers(optimizer)  # (a captured step was warmed up: its buffers exist)
        optimizer.grad_scale = None
        optimizer.found_inf = bad.reshape(1)
        try:
            optimizer.step()
        finally:  # like GradScaler: a later plain optimizer.step() must not see this step's flag
            del optimizer.found_inf
            del optimizer.grad_scale
        return bad
    skipped = bool(bad.item() > 0)
    if not skipped:
        optimizer.step()
    return skipped


def float_lr_state_dict(optimizer):
    """``optimizer.state_dict()`` with every tensor-valued ``lr`` (GraphedTrainStep makes them device tensors) as a float: the
    form core/checkpoint.py and the reference's checkpoints store."""
    sd = optimizer.state_dict()
    for g in sd["param_groups"]:
        if isinstance(g.get("lr"), torch.Tensor):
            g["lr"] = float(g["lr"])
    return sd


class GraphedTrainStep(object):
    """The whole training step (forward under autocast, fused target assignment + losses, backward, device-side skip,
    fused optimizer update: ~600 launches) captured ONCE as a hipGraph and replayed with one launch per step
    (reference loop pipeline_anchor_apex.py:103-130).  Single-process steps only (under DDP the bucketed all-reduce hooks
    keep the eager path); needs the fused optimizer (no host read-back inside the step) and static shapes: ``images`` /
    ``targets`` are copied into the captured input tensors.

    What a captured step freezes, and what it does not:
      * the LEARNING RATE stays live: every parameter group's ``lr`` is turned into a device tensor before the capture
        (fused SGD reads it on the device), and torch's schedulers update a tensor ``lr`` in place (``fill_``), so
        ``lr_scheduler.step()`` / a warm-up that assigns through ``set_lr`` reach the replayed kernels.  SIDE EFFECT: the
        caller's ``param_group["lr"]`` stays a device tensor afterwards -- printing it synchronises, and
        ``optimizer.state_dict()`` would serialise tensors; ``float_lr_state_dict(optimizer)`` returns the state dict with
        plain floats for checkpoints;
      * momentum, weight decay, nesterov are kernel ARGUMENTS of the captured launch: changing them afterwards needs a new
        GraphedTrainStep;
      * the ``warmup`` eager steps (>= 1: allocator pools, MIOpen / rocBLAS plans, the optimizer's lazily created state --
        a capture of the very first step would bake ``is_first_step`` in and overwrite the momentum on every replay) run on
        the sample batch but leave NO trace: parameters, buffers (BatchNorm statistics) and the optimizer state are
        snapshotted before and restored after them (momentum buffers that did not exist are zeroed: see
        ``_ensure_momentum_buffers``)."""

    def __init__(self, model_with_loss, images, targets, anchors, optimizer, autocast_dtype=torch.bfloat16, warmup=3):
        if not (images.is_cuda and _device_skip(optimizer)):
            raise ValueError("GraphedTrainStep needs a HIP device and a fused optimizer")
        if warmup < 1:
            raise ValueError("GraphedTrainStep: warmup >= 1 (the optimizer's first step must not be the captured one)")
        self.optimizer = optimizer
        self.images, self.targets = images.clone(), targets.clone()
        dev = images.device
        for group in optimizer.param_groups:  # a live learning rate (see the class comment)
            if not isinstance(group["lr"], torch.Tensor):
                group["lr"] = torch.tensor(float(group["lr"]), device=dev, dtype=torch.float32)
        _ensure_momentum_buffers(optimizer, every=True)
        snap_model = [t.detach().clone() for t in list(model_with_loss.parameters()) + list(model_with_loss.buffers())]
        snap_opt = {p: {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                    for p, st in optimizer.state.items()}
        snap_lr = [g["lr"].clone() for g in optimizer.param_groups]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # eager warm-up on a side stream
            for _ in range(warmup):
                train_step(model_with_loss, self.images, self.targets, anchors, optimizer, autocast_dtype)
            with torch.no_grad():  # ... undone: the warm-up must not be part of the training trajectory
                for t, s in zip(list(model_with_loss.parameters()) + list(model_with_loss.buffers()), snap_model):
                    t.copy_(s)
                for p, st in snap_opt.items():
                    for k, v in st.items():
                        if isinstance(v, torch.Tensor):
                            optimizer.state[p][k].copy_(v)
                        else:
                            optimizer.state[p][k] = v
                for g, l in zip(optimizer.param_groups, snap_lr):
                    g["lr"].copy_(l)
        torch.cuda.current_stream(dev).wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            with torch.autocast(device_type="cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
                cls_loss, loc_loss, _, _ = model_with_loss(self.images, self.targets, anchors)
                total = cls_loss + loc_loss
            self.bad = (~torch.isfinite(total.detach())).float()
            torch.nan_to_num(total, nan=0.0, posinf=0.0, neginf=0.0).backward()
            _step_unless(optimizer, self.bad)
            self.cls_loss, self.loc_loss = cls_loss.detach(), loc_loss.detach()

    def set_lr(self, lr, group=None):
        """Assign a learning rate (warm-up loops that write ``param_group['lr'] = x`` would replace the live tensor)."""
        for i, g in enumerate(self.optimizer.param_groups):
            if group is None or group == i:
                g["lr"].fill_(float(lr))

    def __call__(self, images, targets):
        self.images.copy_(images, non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        self.graph.replay()
        return self.cls_loss, self.loc_loss, self.bad


def train_anchor_based_epoch(model, data_loader, optimizer, anchors, epoch, device, local_rank, log_every=20):
    """Epoch loop (reference pipeline_anchor_apex.py:75-171 without tqdm/TensorBoard): stdout lines on
    rank 0, losses fetched from the device only every ``log_every`` steps."""
    model.train()
    start = time.time()
    n = len(data_loader)
    for batch_idx, (images, targets) in enumerate(data_loader):
        if images.device != device:
            images, targets = images.to(device), targets.to(device)
        if targets.dtype != torch.float:
            targets = targets.float()
        cls_loss, loc_loss, skipped = train_step(model, images, targets, anchors, optimizer)
        if local_rank == 0 and ((batch_idx + 1) % log_every == 0 or batch_idx + 1 == n):
            print("Train: epoch {} | {}/{} | cls_loss {:.4f} | loc_loss {:.4f} | lr {:.5f} | skipped {} | "
                  "{:.1f}s".format(epoch, batch_idx + 1, n, float(cls_loss), float(loc_loss),
                                   optimizer.param_groups[0]["lr"], int(skipped), time.time() - start))


@torch.no_grad()
def eval_anchor_based_epoch(model, data_loader, decoder, anchors, num_classes, device):
    """Eval epoch (reference pipeline_anchor_basic.py:150-182 without tqdm/TensorBoard): forward, decode + NMS and the
    mAP bookkeeping all stay on the device; the host sees one number per class at the end.  Returns
    (mAP, (precision, recall, ap))."""
    from ssds.core.evaluation_metrics import MeanAveragePrecision

    model.eval()
    metric = MeanAveragePrecision(num_classes, decoder.conf_threshold, decoder.nms_threshold)
    for images, targets in data_loader:
        if images.device != device:
            images, targets = images.to(device), targets.to(device)
        targets = targets.float().clone()
        loc, conf = model(images)
        detections = decoder(loc, conf, anchors)
        targets[:, :, 2:4] = targets[:, :, :2] + targets[:, :, 2:4]  # xywh -> ltrb (:175)
        metric(detections, targets)
    return metric.get_results()
