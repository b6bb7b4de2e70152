"""Records the eval forward of a detector as a ``ConvPlan`` (fused_conv.py): every Conv-BN-activation
group becomes one descriptor of the fused HIP kernels, block-level residual adds ride in the epilogue of
the projecting 1x1 conv, and the loc | conf convs of each level become one split-output GEMM.

Covered: MobileNet v1/v2 backbones (nets/mobilenet.py), the SSD extras and heads (ssds/ssd.py) -- i.e. the
network of BASELINE configs 1, 2 and 4 -- and the FPN / BiFPN necks with their shared towers (ssds/fpn.py,
ssds/bifpn.py; configs 3 and 5) on top of backbone feature maps handed in as external inputs.  Anything else
raises ``PlanUnsupported`` and the caller runs the module-by-module path instead (and says so in
``fused_conv.STATS``)."""
import torch.nn as nn

import os

from .fused_conv import ConvPack, ConvPlan, MbPack, StemPack, conv_kind, pack_heads, sequential_groups, xpair_supported


class PlanUnsupported(Exception):
    pass


def flatten(module):
    """nn.Sequential tree -> flat list of leaf modules in execution order."""
    out = []
    for m in module.children():
        if isinstance(m, nn.Sequential):
            out.extend(flatten(m))
        else:
            out.append(m)
    return out


def groups_of(module):
    seq = nn.Sequential(*flatten(module))
    groups = sequential_groups(seq)
    if groups is None:
        raise PlanUnsupported("not a Conv-BN-act chain: {}".format(type(module).__name__))
    for conv, _, _ in groups:
        if conv_kind(conv) is None:
            raise PlanUnsupported("conv not covered by the HIP kernels: {}".format(conv))
    return groups


def record_chain(plan, val, module, residual=None, keep_input=False, lane=0):
    """Record the Conv-BN-act chain of ``module`` starting from value ``val``; the residual (if any, always
    the chain input) is added in the epilogue of the LAST conv.  Intermediate buffers are released as soon
    as they have been read; the chain input is released at the end unless ``keep_input``."""
    groups = groups_of(module)
    if residual is None and len(groups) == 2:  # an SSD extra layer on a small map: one launch (csrc/ssdk_xpair.hip)
        p1, p2 = (ConvPack(conv, bn, act, plan.dtype) for conv, bn, act in groups)
        if xpair_supported(p1, p2, val[3], val[4]):
            out = plan.xpair(val, p1, p2, lane=lane)
            if not keep_input:
                plan.release(val)
            return out
    cur = val
    for i, (conv, bn, act) in enumerate(groups):
        pack = ConvPack(conv, bn, act, plan.dtype)
        last = i == len(groups) - 1
        nxt = plan.conv(cur, pack, residual=residual if last else None, lane=lane)
        if cur is not val:
            plan.release(cur)
        cur = nxt
    if not keep_input:
        plan.release(val)
    return cur


def record_mobilenet(plan, val, net, on_output=None):
    from ssds.modeling.nets.mobilenet import InvertedResidual, MobileNetEx

    if not isinstance(net, MobileNetEx):
        raise PlanUnsupported("backbone {} has no planner".format(type(net).__name__))
    fused_block = os.environ.get("SSDK_FUSED_BLOCK", "1") != "0"
    first_blk = net.layer1[0]
    stem_fused = False
    if fused_block and isinstance(first_blk, InvertedResidual) and not first_blk.use_res_connect:
        sg, bg = groups_of(net.conv1), groups_of(first_blk.conv)
        if MbPack.stem_supported(sg, bg):  # stem conv + expand-free first block: ONE launch from the image
            pk = MbPack(bg, False, plan.dtype, stem_group=sg[0])
            if pk.fp16_safe():  # (folded weights outside the fp16 range of the kernel's internals: layer by layer)
                cur = plan.mbconv(val, pk)
                stem_fused = True
    if not stem_fused:
        cur = record_chain(plan, val, net.conv1, keep_input=True)  # the image is not an arena buffer
    outputs = []
    for j in range(len(net.settings)):
        level = j + 1
        if level > max(net.outputs):
            break
        for bi, blk in enumerate(getattr(net, "layer{}".format(level))):
            if stem_fused and level == 1 and bi == 0:
                continue
            is_out_input = any(cur is o for o in outputs)
            if isinstance(blk, InvertedResidual):
                res = cur if blk.use_res_connect else None
                groups = groups_of(blk.conv)
                pk = MbPack(groups, blk.use_res_connect, plan.dtype) if fused_block and MbPack.supported(groups, blk.use_res_connect) else None
                if pk is not None and pk.fp16_safe():
                    nxt = plan.mbconv(cur, pk)  # whole block, 1 launch
                    if not is_out_input:
                        plan.release(cur)
                    cur = nxt
                else:
                    cur = record_chain(plan, cur, blk.conv, residual=res, keep_input=is_out_input)
            else:
                cur = record_chain(plan, cur, blk, keep_input=is_out_input)
        if level in net.outputs:
            outputs.append(cur)
            if on_output is not None:
                on_output(len(outputs) - 1, cur)
    return outputs


def build_ssd_plan(model, x):
    """SSD (ssds/ssd.py) on a planned backbone (MobileNet, ResNet) -> finalized ConvPlan for inputs shaped like ``x``."""
    plan = ConvPlan(x.device, x.dtype, x.shape)

    def head(i, f, lane=None, position=None):
        # recorded right behind its feature map: the executor forks it onto the side stream, where it overlaps
        # the rest of the backbone / the extras chain (feature maps are never released, so that is safe)
        l, c = model.loc[i], model.conf[i]
        if conv_kind(l) != "dense" or conv_kind(c) != "dense":
            raise PlanUnsupported("head conv not covered")
        plan.head(f, pack_heads(l, c, plan.dtype), split=l.out_channels, act="none", act2="sigmoid", tag="both", lane=lane,
                  position=position)

    from ssds.modeling.nets.mobilenet import MobileNetEx

    # (Round 5 measured the mirror-image ordering -- extras chain + small heads as a side-stream run next to the two backbone-level
    #  heads -- at 60.9 / 62.4 k against 61.9 / 62.9 k img/s in line; the SSDK_SSD_TAIL_SIDE switch and its two recordings are
    #  gone in round 6: docs/HISTORY.md.)
    if isinstance(model.backbone, MobileNetEx):
        feats = record_mobilenet(plan, plan.input_value(), model.backbone, on_output=head)
    else:
        feats = record_backbone(plan, plan.input_value(), model.backbone)
        for i, f in enumerate(feats):
            head(i, f)
    # The heads of the extras' levels are recorded BEHIND the extras chain, next to each other (outputs keep the level order):
    # they depend only on their feature maps, and the executor launches neighbouring small-map layers that do not read each
    # other's outputs as ONE kernel (ssdk_run_ops -> conv_smallmap_group_kernel: the 4x4 / 2x2 / 1x1 heads are chains of
    # dependent latencies with 128 / 32 / 8 workgroups each).  SSDK_HEAD_BALANCE=0: every head right behind its feature map.
    balance = os.environ.get("SSDK_HEAD_BALANCE", "1") != "0" and len(model.extras) > 1
    deferred = []
    for j, extra in enumerate(model.extras):
        feats.append(record_chain(plan, feats[-1], extra, keep_input=True))
        if balance:
            deferred.append((len(feats) - 1, feats[-1]))
        else:
            head(len(feats) - 1, feats[-1])
    for i, f in deferred:
        head(i, f, lane=0, position=i)
    return plan.finalize()


# --------------------------------------------------------------------------------------------------------------
# ResNet backbones (nets/resnet.py)
# --------------------------------------------------------------------------------------------------------------
def _pack(conv, bn, act, dtype, kinds=("dense",)):
    if conv_kind(conv) not in kinds:
        raise PlanUnsupported("conv not covered by the HIP kernels: {}".format(conv))
    return ConvPack(conv, bn, act, dtype)


def record_resnet(plan, val, net):
    """conv1/bn1/relu (7x7 stem kernel) -> maxpool -> layer1..4; every block ends in ONE launch that adds the
    identity and applies the ReLU in the epilogue of its last conv (res_mode bit 1)."""
    from ssds.modeling.nets.resnet import BasicBlock, Bottleneck, ResNet

    if not isinstance(net, ResNet):
        raise PlanUnsupported("backbone {} has no planner".format(type(net).__name__))
    if not StemPack.supported(net.conv1, net.bn1):
        raise PlanUnsupported("stem not covered")
    dt = plan.dtype
    cur = plan.stem7(val, StemPack(net.conv1, net.bn1, "relu", dt))
    pooled = plan.pool(cur)
    plan.release(cur)
    cur = pooled
    outputs = []
    for li, layer in enumerate([net.layer1, net.layer2, net.layer3, net.layer4]):
        level = li + 2
        if level > max(net.outputs):
            break
        for blk in layer:
            keep_cur = any(cur is o for o in outputs)
            idt = cur
            if blk.downsample is not None:
                dconv, dbn = blk.downsample[0], blk.downsample[1]
                idt = plan.conv(cur, _pack(dconv, dbn, "none", dt))
            if isinstance(blk, Bottleneck):
                o1 = plan.conv(cur, _pack(blk.conv1, blk.bn1, "relu", dt))
                o2 = plan.conv(o1, _pack(blk.conv2, blk.bn2, "relu", dt))
                plan.release(o1)
                out = plan.conv(o2, _pack(blk.conv3, blk.bn3, "relu", dt), residual=idt, res_mode=2)
                plan.release(o2)
            elif isinstance(blk, BasicBlock):
                o1 = plan.conv(cur, _pack(blk.conv1, blk.bn1, "relu", dt))
                out = plan.conv(o1, _pack(blk.conv2, blk.bn2, "relu", dt), residual=idt, res_mode=2)
                plan.release(o1)
            else:
                raise PlanUnsupported("block {} has no planner".format(type(blk).__name__))
            if idt is not cur:
                plan.release(idt)
            if not keep_cur:
                plan.release(cur)
            cur = out
        if level in net.outputs:
            outputs.append(cur)
    return outputs


def record_regnet(plan, val, net):
    """RegNetX (nets/regnet.py): 3x3/s2 stem, then bottleneck blocks 1x1 -> grouped 3x3 (16 channels per group,
    stride s) -> 1x1 whose last conv adds the (projected) skip and applies the ReLU in its epilogue."""
    from ssds.modeling.nets.regnet import RegNet

    if not isinstance(net, RegNet):
        raise PlanUnsupported("backbone {} has no planner".format(type(net).__name__))
    dt = plan.dtype
    cur = plan.conv(val, _pack(net.stem.conv, net.stem.bn, "relu", dt, kinds=("stem",)))
    outputs = []
    for i, stage in enumerate([net.s1, net.s2, net.s3, net.s4]):
        level = i + 1
        if level > max(net.outputs):
            break
        for blk in stage.children():
            keep_cur = any(cur is o for o in outputs)
            skip = cur
            if blk.proj_block:
                skip = plan.conv(cur, _pack(blk.proj, blk.bn, "none", dt))
            f = blk.f
            a = plan.conv(cur, _pack(f.a, f.a_bn, "relu", dt))
            b = plan.conv(a, _pack(f.b, f.b_bn, "relu", dt, kinds=("g16", "dense")))
            plan.release(a)
            out = plan.conv(b, _pack(f.c, f.c_bn, "relu", dt), residual=skip, res_mode=2)
            plan.release(b)
            if skip is not cur:
                plan.release(skip)
            if not keep_cur:
                plan.release(cur)
            cur = out
        if level in net.outputs:
            outputs.append(cur)
    return outputs


def record_backbone(plan, val, net):
    """Backbone maps of a planned backbone (MobileNet, ResNet, RegNetX) from the image value; PlanUnsupported
    otherwise."""
    from ssds.modeling.nets.mobilenet import MobileNetEx
    from ssds.modeling.nets.regnet import RegNet
    from ssds.modeling.nets.resnet import ResNet

    if isinstance(net, MobileNetEx):
        return record_mobilenet(plan, val, net)
    if isinstance(net, ResNet):
        return record_resnet(plan, val, net)
    if isinstance(net, RegNet):
        return record_regnet(plan, val, net)
    raise PlanUnsupported("backbone {} has no planner".format(type(net).__name__))


# --------------------------------------------------------------------------------------------------------------
# FPN / BiFPN necks + shared towers on external backbone features
# --------------------------------------------------------------------------------------------------------------
def _tower_packs(tower, dtype):
    """SharedHead (4 x ConvBNReLU + final conv) -> ([ConvPack] of the ConvBNReLU layers, final ConvPack).  The
    towers are shared by every pyramid level, so they are packed once."""
    mods = list(tower.children())
    body = []
    for m in mods[:-1]:
        (conv, bn, act), = groups_of(m)
        body.append(ConvPack(conv, bn, act, dtype))
    last = mods[-1]
    if conv_kind(last) != "dense":
        raise PlanUnsupported("tower output conv not covered")
    return body, ConvPack(last, None, "none", dtype)


def _record_towers(plan, xx, loc_packs, conf_packs, lane=None, level=None):
    for (body, final), act, tag in ((loc_packs, "none", "loc"), (conf_packs, "sigmoid", "conf")):
        cur = xx
        for pk in body:
            nxt = plan.conv(cur, pk, role="tower", lane=lane or 0)
            if cur is not xx:
                plan.release(cur)
            cur = nxt
        plan.head(cur, final, act=act, tag=tag, lane=lane, level=level)
        if cur is not xx:
            plan.release(cur)


# pixels (batch x H x W) up to which a pyramid level goes to the side stream.  The 20x20, 10x10 and 5x5 levels of
# FPN-ResNet50@640 at batch 32 are 30 launches of a few dozen workgroups each, 0.85 ms one after the other, next to 3.3 ms of
# chip-filling launches on the two big levels (16384: +4 %); with the 40x40 level (51 200 pixels) on the side stream as well
# only the largest level stays in line and the two streams' launches fill each other's tails: another +1.9 % there, +2.3 % on
# BiFPN-RegNetX008@896 (measured, same box; SSDK_SMALL_LEVEL_PIXELS overrides).
SMALL_LEVEL_PIXELS = 65536


def _record_extras_and_towers(plan, model, pyramid, raw_last):
    """reference fpn.py:89-97 / bifpn.py:131-138: extras[i] on pyramid level i, extras[n] on the RAW last
    backbone map, later extras on the previous extra; both towers on every result.

    Order of the recording (round 4): all extras first, then the towers + heads of the SMALL levels as one contiguous
    block for the executor's side stream (one fork behind the extras, one join at the end of the plan), then the big
    levels on the main stream -- the small levels' launch chain runs next to the big levels instead of behind them.
    ``SSDK_LEVEL_LANES=0`` keeps everything in level order on one stream."""
    import os

    n = len(pyramid)
    loc_packs, conf_packs = _tower_packs(model.loc, plan.dtype), _tower_packs(model.conf, plan.dtype)
    levels = []
    xx = None
    for i, v in enumerate(model.extras):
        src = pyramid[i] if i < n else (raw_last if i == n else xx)
        xx = record_chain(plan, src, v, keep_input=True)
        levels.append(xx)
    limit = SMALL_LEVEL_PIXELS  # (a module attribute: tests monkeypatch it)
    small = [xx[1] * xx[3] * xx[4] <= limit for xx in levels]
    lanes = os.environ.get("SSDK_LEVEL_LANES", "1") != "0" and any(small) and not all(small)
    if lanes and sum(10 for sm in small if sm) > 320:
        lanes = False
    if not lanes:
        for i, xx in enumerate(levels):
            _record_towers(plan, xx, loc_packs, conf_packs, level=i)
        return
    for i, xx in enumerate(levels):
        if small[i]:
            _record_towers(plan, xx, loc_packs, conf_packs, lane=2, level=i)
    for i, xx in enumerate(levels):
        if not small[i]:
            _record_towers(plan, xx, loc_packs, conf_packs, lane=0, level=i)


def _neck_inputs(model, features, image):
    """(plan, backbone map values): from the image through a planned backbone, or the maps as external inputs."""
    if image is not None:
        plan = ConvPlan(image.device, image.dtype, image.shape)
        return plan, record_backbone(plan, plan.input_value(), model.backbone)
    f0 = features[0]
    plan = ConvPlan(f0.device, f0.dtype)
    return plan, [plan.add_input(f.shape) for f in features]


def build_fpn_plan(model, features=None, image=None):
    """SSDFPN (ssds/fpn.py) neck + towers -> finalized ConvPlan, on the backbone maps ``features`` (external
    inputs) or, with ``image``, including a planned backbone.  The 1x1 lateral of level i and the nearest-x2
    upsample-add of level i+1 (reference fpn.py:80-87) are ONE launch: the coarser map rides in as a
    half-resolution residual of the lateral GEMM's epilogue."""
    plan, vals = _neck_inputs(model, features, image)
    n = len(vals)
    pyramid = [None] * n
    for i in range(n - 1, -1, -1):
        conv = model.transforms[i]
        if conv_kind(conv) != "dense":
            raise PlanUnsupported("lateral conv not covered: {}".format(conv))
        pack = ConvPack(conv, None, "none", plan.dtype)
        if i == n - 1:
            pyramid[i] = plan.conv(vals[i], pack)
        else:
            if pyramid[i + 1][3] * 2 != vals[i][3] or pyramid[i + 1][4] * 2 != vals[i][4]:
                raise PlanUnsupported("pyramid levels are not exact halves")
            pyramid[i] = plan.conv(vals[i], pack, residual=pyramid[i + 1], res_mode=1)
    _record_extras_and_towers(plan, model, pyramid, vals[n - 1])
    return plan.finalize()


def _fusion_weights(w, dtype):
    """relu(w) / (sum relu(w) + 1e-6), rounded to the model dtype like the module path does (bifpn.py:35-38)."""
    import torch

    w = torch.relu(w.detach().float())
    w = w / (w.sum(0) + 1e-6)
    return w.to(dtype).float().cpu()


def _record_bifpn_layer(plan, m, xx):
    from ssds import _native as N

    n = m.levels
    assert len(xx) == n
    w1, w2 = _fusion_weights(m.w1, plan.dtype), _fusion_weights(m.w2, plan.dtype)
    xx = list(xx)
    skips = [None] + xx[1:-1] + [None]
    for i in range(n - 1, 0, -1):  # top-down (reference bifpn.py:41-46)
        if xx[i][3] * 2 != xx[i - 1][3] or xx[i][4] * 2 != xx[i - 1][4]:
            raise PlanUnsupported("BiFPN levels are not exact halves")
        fused = plan.fuse(xx[i - 1], xx[i], weights=(w1[0, i - 1], w1[1, i - 1], 0.0), mode_b=N.FUSE_UP2)
        xx[i - 1] = record_chain(plan, fused, getattr(m, "top-down-{}".format(i - 1)))
    for i in range(0, n - 2):  # bottom-up with skip (reference bifpn.py:49-55)
        fused = plan.fuse(xx[i + 1], xx[i], skips[i + 1], weights=(w2[0, i], w2[1, i], w2[2, i]),
                          mode_b=N.FUSE_POOL2, mode_c=N.FUSE_SAME)
        xx[i + 1] = record_chain(plan, fused, getattr(m, "bottom-up-{}".format(i + 1)))
    fused = plan.fuse(xx[n - 1], xx[n - 2], weights=(w1[0, n - 1], w1[1, n - 1], 0.0), mode_b=N.FUSE_POOL2)
    xx[n - 1] = record_chain(plan, fused, getattr(m, "bottom-up-{}".format(n - 1)))  # reference bifpn.py:57-62
    return xx


def build_bifpn_plan(model, features=None, image=None):
    """SSDBiFPN (ssds/bifpn.py): 1x1 transforms, stacked BiFPN layers (weighted fusions as one launch each),
    extras and shared towers -> finalized ConvPlan (inputs as in ``build_fpn_plan``)."""
    plan, vals = _neck_inputs(model, features, image)
    n = len(vals)
    xx = []
    for i in range(n):
        conv = model.transforms[i]
        if conv_kind(conv) != "dense":
            raise PlanUnsupported("transform conv not covered: {}".format(conv))
        xx.append(plan.conv(vals[i], ConvPack(conv, None, "none", plan.dtype)))
    for m in model.stack_bifpn:
        xx = _record_bifpn_layer(plan, m, xx)
    _record_extras_and_towers(plan, model, xx, vals[n - 1])
    return plan.finalize()
