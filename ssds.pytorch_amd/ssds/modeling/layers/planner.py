"""Records the eval forward of a detector as a ``ConvPlan`` (fused_conv.py): every Conv-BN-activation
group becomes one descriptor of the fused HIP kernels, block-level residual adds ride in the epilogue of
the projecting 1x1 conv, and the loc | conf convs of each level become one split-output GEMM.

Covered: MobileNet v1/v2 backbones (nets/mobilenet.py), the SSD extras and heads (ssds/ssd.py) -- i.e. the
network of BASELINE configs 1, 2 and 4.  Anything else raises ``PlanUnsupported`` and the caller runs the
module-by-module path instead (and says so in ``fused_conv.STATS``)."""
import torch.nn as nn

import os

from .fused_conv import ConvPack, ConvPlan, MbPack, conv_kind, pack_heads, sequential_groups


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


def record_chain(plan, val, module, residual=None, keep_input=False):
    """Record the Conv-BN-act chain of ``module`` starting from value ``val``; the residual (if any, always
    the chain input) is added in the epilogue of the LAST conv.  Intermediate buffers are released as soon
    as they have been read; the chain input is released at the end unless ``keep_input``."""
    groups = groups_of(module)
    cur = val
    for i, (conv, bn, act) in enumerate(groups):
        pack = ConvPack(conv, bn, act, plan.dtype)
        last = i == len(groups) - 1
        nxt = plan.conv(cur, pack, residual=residual if last else None)
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
            cur = plan.mbconv(val, MbPack(bg, False, plan.dtype, stem_group=sg[0]))
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
                if fused_block and MbPack.supported(groups, blk.use_res_connect):
                    nxt = plan.mbconv(cur, MbPack(groups, blk.use_res_connect, plan.dtype))  # whole block, 1 launch
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
    """SSD (ssds/ssd.py) on a MobileNet backbone -> finalized ConvPlan for inputs shaped like ``x``."""
    plan = ConvPlan(x.device, x.dtype, x.shape)

    def head(i, f):
        # recorded right behind its feature map: the executor forks it onto the side stream, where it overlaps
        # the rest of the backbone / the extras chain (feature maps are never released, so that is safe)
        l, c = model.loc[i], model.conf[i]
        if conv_kind(l) != "dense" or conv_kind(c) != "dense":
            raise PlanUnsupported("head conv not covered")
        plan.head(f, pack_heads(l, c, plan.dtype), split=l.out_channels, act2="sigmoid")

    feats = record_mobilenet(plan, plan.input_value(), model.backbone, on_output=head)
    for extra in model.extras:
        feats.append(record_chain(plan, feats[-1], extra, keep_input=True))
        head(len(feats) - 1, feats[-1])
    return plan.finalize()
