"""Deployment sidecar of the reference's exporter (``ssds/utils/export.py:94-106``): the JSON that travels next to
an exported network -- image size, post-processing thresholds and the per-level flattened anchors -- plus the
``state_dict`` checkpoint.  The ONNX / TensorRT graph export of the reference is out of scope; the
parameters file keeps the exact keys and value layout, so tooling that reads the reference's ``<model>.onnx.json``
reads this one."""
import json

import torch


def export_params(cfg, anchors, nhwc=False):
    """dict with the reference's keys (export.py:94-104); ``anchors`` = OrderedDict{stride: [A,4]} in level order."""
    return {
        "image_size": list(cfg.MODEL.IMAGE_SIZE),
        "score": cfg.POST_PROCESS.SCORE_THRESHOLD,
        "iou": cfg.POST_PROCESS.IOU_THRESHOLD,
        "max_detects": cfg.POST_PROCESS.MAX_DETECTIONS,
        "max_detects_per_level": cfg.POST_PROCESS.MAX_DETECTIONS_PER_LEVEL,
        "rescore": cfg.POST_PROCESS.RESCORE_CENTER,
        "use_diou": cfg.POST_PROCESS.USE_DIOU,
        "NHWC": bool(nhwc),
        "anchors": [torch.as_tensor(v).reshape(-1).tolist() for _, v in anchors.items()],
    }


def save_export(model, cfg, anchors, export_path, nhwc=False):
    """``<export_path>.json`` (parameters, indent 2 like the reference) + ``<export_path>.pth`` (state_dict)."""
    with open(export_path + ".json", "w") as f:
        json.dump(export_params(cfg, anchors, nhwc), f, indent=2)
    torch.save(model.state_dict(), export_path + ".pth")
    return export_path + ".json", export_path + ".pth"


def decoder_from_params(params):
    """Rebuild (Decoder, anchors OrderedDict) from an exported parameters dict.  The strides are recovered from
    the anchors themselves: an anchor set is centred on its stride cell (box.py:46-58: x1 + x2 + 1 == stride)."""
    from collections import OrderedDict

    from ssds.modeling.layers.decoder import Decoder

    anchors = OrderedDict()
    for flat in params["anchors"]:
        a = torch.tensor(flat, dtype=torch.float32).view(-1, 4)
        stride = int(round(float(a[0, 0] + a[0, 2] + 1)))
        anchors[stride] = a
    dec = Decoder(params["score"], params["iou"], params["max_detects"], params["max_detects_per_level"],
                  params["rescore"], params["use_diou"])
    return dec, anchors
