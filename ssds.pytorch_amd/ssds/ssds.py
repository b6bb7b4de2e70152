"""``SSDDetector`` -- the inference entry point (reference ``ssds/ssds.py:7-68``): cfg file ->
model + anchors + decoder; numpy image(s) in, numpy detections out."""
import ctypes

import numpy as np
import torch

from . import _native as N
from .core import checkpoint, config
from .modeling import model_builder


def preprocess(imgs, mean, std, dtype=torch.bfloat16):
    """Raw image batch on a HIP device -> normalised [N,C,H,W] ``dtype`` tensor in ONE launch
    (``ssdk_preprocess``): transpose (if NHWC), ``(x - mean) / std`` in fp32 in the reference's order
    (ssds.py:55), one rounding.  imgs: uint8 / float32 / bf16 / f16, [N,H,W,C] or [N,C,H,W] with C <= 4 (NHWC is
    recognised like the reference does, ``shape[3] == 3``, ssds.py:53); mean / std: scalars or per-channel lists."""
    N.require_device(imgs, "preprocess")
    nhwc = imgs.shape[3] == 3
    imgs = imgs.contiguous()
    n = int(imgs.shape[0])
    h, w, c = (int(v) for v in (imgs.shape[1:] if nhwc else (imgs.shape[2], imgs.shape[3], imgs.shape[1])))
    src = N.U8 if imgs.dtype == torch.uint8 else N.dtype_code(imgs)

    def vec(v):
        v = [float(v)] * c if not isinstance(v, (list, tuple)) else [float(t) for t in v]
        return (ctypes.c_float * c)(*v)

    y = torch.empty((n, c, h, w), device=imgs.device, dtype=dtype)
    with torch.cuda.device(imgs.device):
        rc = N.lib.ssdk_preprocess(imgs.data_ptr(), src, N.NHWC if nhwc else N.NCHW, n, h, w, c, vec(mean), vec(std),
                                   y.data_ptr(), N._DTYPES[dtype], N.stream_ptr(imgs.device))
    N.check(rc, "preprocess")
    return y


class SSDDetector(object):
    r"""Args:
        cfg_file (str):  path to the yaml config
        is_print (bool): print the model and the anchor shapes
        dtype: compute dtype of the network on the HIP device (bf16 by default; the reference is fp32)
    """

    def __init__(self, cfg_file, is_print=False, dtype=torch.bfloat16):
        cfg = config.cfg_from_file(cfg_file)
        print("===> Building model")
        self.model = model_builder.create_model(cfg.MODEL)
        if is_print:
            print("Model architectures:\n{}\n".format(self.model))
        if not torch.cuda.is_available():
            raise RuntimeError("SSDDetector needs a HIP device (MI355X); there is no CPU path")
        self.device = torch.device("cuda:0")
        if cfg.RESUME_CHECKPOINT:
            print("Loading initial model weights from {:s}".format(cfg.RESUME_CHECKPOINT))
            checkpoint.resume_checkpoint(self.model, cfg.RESUME_CHECKPOINT, "")
        self.dtype = dtype
        self.model.eval().to(self.device, dtype)
        self.anchors = model_builder.create_anchors(cfg.MODEL, self.model, cfg.MODEL.IMAGE_SIZE, is_print)
        self.decoder = model_builder.create_decoder(cfg.POST_PROCESS)
        self.image_size = tuple(cfg.MODEL.IMAGE_SIZE)
        self.num_classes = cfg.MODEL.NUM_CLASSES
        self.mean = cfg.DATASET.PREPROC.MEAN
        self.std = cfg.DATASET.PREPROC.STD

    @torch.no_grad()
    def __call__(self, imgs):
        r"""imgs: np.ndarray [H,W,3], [3,H,W], [N,H,W,3] or [N,3,H,W] (reference ssds.py:41-68).
        Returns (scores [N,100] f32, boxes [N,100,4] int, classes [N,100] int); 3-d input drops N."""
        pick1st = False
        if len(imgs.shape) == 3:
            imgs = imgs[None, ...]
            pick1st = True
        if len(imgs.shape) != 4:
            raise AssertionError("image dims has to be 3 or 4")
        if imgs.dtype not in (np.uint8, np.float32, np.float16):
            imgs = imgs.astype(np.float32)  # what the reference's torch.Tensor(imgs) does (ssds.py:54)
        raw = torch.from_numpy(np.ascontiguousarray(imgs)).to(self.device)  # raw upload (uint8: 1/4 of the fp32 bytes)
        x = preprocess(raw, self.mean, self.std, self.dtype)  # transpose + normalise + cast: one launch
        loc, conf = self.model(x)
        detections = self.decoder(loc, conf, self.anchors)
        out_scores, out_boxes, out_classes = (d.cpu().numpy() for d in detections)  # the one D2H copy
        if pick1st:
            return out_scores[0], out_boxes[0].astype(int), out_classes[0].astype(int)
        return out_scores, out_boxes.astype(int), out_classes.astype(int)
