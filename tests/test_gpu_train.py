"""Training step on the HIP device: target assignment by the match kernel inside ModelWithLossBasic, losses
against the same step computed on CPU in fp32 with the numpy oracle's target assignment."""
from collections import OrderedDict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_model_with_loss_matches_cpu_oracle_step():
    import torch
    from oracle import box_oracle as O
    from ssds.core import criterion
    from ssds.dataset.synthetic import SyntheticDetectionLoader
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import box
    from ssds.pipeline.pipeline_anchor_ddp import ModelWithLossBasic, train_step

    torch.manual_seed(0)
    o, e, h = ssds.SSD.add_extras([[5, 7, "Conv:S"], [96, 320, 64]], [2, 2, 2], 5)
    model = ssds.SSD(nets.MobileNetV2(outputs=o), e, h, 5)
    mwl = ModelWithLossBasic(model, criterion.FocalLoss(), criterion.SmoothL1Loss(), 5, [0.5, 0.4], 0).cuda()
    anchors = OrderedDict((s, box.generate_anchors(s, [1], [2.0, 2.828])) for s in (16, 32, 64))
    loader = SyntheticDetectionLoader(4, (128, 128), 5, steps=1, device=torch.device("cuda"), max_gt=6)
    images, targets = loader.batch()
    targets[..., 2:4] = targets[..., 2:4].clamp(min=24)  # big enough to match the stride-16 anchors
    targets[targets[..., 4] < 0] = -1
    mwl.train()
    cls_loss, loc_loss, cls_l, loc_l = mwl(images, targets, anchors)  # fp32 on the device
    assert torch.isfinite(cls_loss) and torch.isfinite(loc_loss) and float(loc_loss) > 0

    # CPU fp32 replica with oracle targets
    import copy
    cpu = copy.deepcopy(mwl).cpu()
    loc, conf = cpu.model(images.cpu())
    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    c_sum, l_sum, fg = 0.0, 0.0, 0.0
    for j, (stride, _) in enumerate(anchors.items()):
        size = tuple(conf[j].shape[-2:])
        ct, bt, dp = (torch.from_numpy(x) for x in O.extract_targets(targets.cpu().numpy(), oanch, 5, stride, size, (0.5, 0.4)))
        fg += float((dp > 0).sum().clamp(min=1))
        c = conf[j].view_as(ct).float()
        c_sum += float(((dp >= 0).expand_as(ct).float() * cpu.cls_criterion(c, ct, dp)).sum())
        l = loc[j].view_as(bt).float()
        ll = cpu.loc_criterion(l, bt)
        l_sum += float(((dp > 0).expand_as(ll).float() * ll).sum())
    np.testing.assert_allclose(float(cls_loss), c_sum / fg, rtol=2e-3)
    np.testing.assert_allclose(float(loc_loss), l_sum / fg, rtol=2e-3)

    # one optimiser step under bf16 autocast changes the weights and keeps them finite
    opt = torch.optim.SGD(mwl.parameters(), lr=0.01, momentum=0.9)
    before = torch.cat([p.detach().flatten() for p in mwl.parameters()]).clone()
    c, l, skipped = train_step(mwl, images, targets, anchors, opt)
    after = torch.cat([p.detach().flatten() for p in mwl.parameters()])
    assert not skipped and torch.isfinite(after).all() and not torch.equal(before, after)
