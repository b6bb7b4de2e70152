"""Plugin factories -- signatures and return types of the reference's
``ssds/modeling/model_builder.py`` (create_model :9-27, create_anchors :30-56, create_decoder :59-74)."""
from collections import OrderedDict

import torch

from . import nets, ssds
from .layers.box import configure_ratio_scale, generate_anchors
from .layers.decoder import Decoder


def create_model(cfg):
    """cfg.MODEL -> detector nn.Module: ``getattr(ssds, cfg.SSDS)`` head on ``getattr(nets, cfg.NETS)``
    backbone (reference model_builder.py:9-27)."""
    ratios, scales = configure_ratio_scale(len(cfg.SIZES), cfg.ASPECT_RATIOS, cfg.SIZES)
    number_box = [len(r) * len(s) for r, s in zip(ratios, scales)]
    nets_outputs, extras, head = getattr(ssds, cfg.SSDS).add_extras(
        feature_layer=cfg.FEATURE_LAYER, mbox=number_box, num_classes=cfg.NUM_CLASSES
    )
    model = getattr(ssds, cfg.SSDS)(
        backbone=getattr(nets, cfg.NETS)(outputs=nets_outputs, num_images=cfg.NUM_IMAGES),
        extras=extras,
        head=head,
        num_classes=cfg.NUM_CLASSES,
    )
    return model


def create_anchors(cfg, model, image_size, visualize=False):
    """OrderedDict{stride: anchors[A,4]} (CPU fp32) in level order.  Like the reference
    (model_builder.py:30-56) the strides come from one probe forward: stride_l = W_in // W_conf_l."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        p = next(model.parameters())
        x = torch.rand((1, 3, image_size[0], image_size[1]), device=p.device, dtype=p.dtype)
        conf = model(x)[-1]
        strides = [x.shape[-1] // c.shape[-1] for c in conf]
    model.train(was_training)
    ratios, scales = configure_ratio_scale(len(strides), cfg.ASPECT_RATIOS, cfg.SIZES)
    anchors = OrderedDict(
        [(strides[i], generate_anchors(strides[i], ratios[i], scales[i])) for i in range(len(strides))]
    )
    if visualize:
        print("Anchor Boxs (width, height)")
        for k, v in anchors.items():
            print("Stride {}: {}".format(k, (v[:, 2:] - v[:, :2] + 1).int().tolist()))
    return anchors


def create_decoder(cfg):
    """cfg.POST_PROCESS -> Decoder (reference model_builder.py:59-74)."""
    return Decoder(
        cfg.SCORE_THRESHOLD,
        cfg.IOU_THRESHOLD,
        cfg.MAX_DETECTIONS,
        cfg.MAX_DETECTIONS_PER_LEVEL,
        cfg.RESCORE_CENTER,
        cfg.USE_DIOU,
    )
