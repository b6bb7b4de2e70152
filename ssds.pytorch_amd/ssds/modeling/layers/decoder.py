"""``Decoder`` -- same constructor/attributes/call contract as the reference's
``ssds/modeling/layers/decoder.py:15-49``; the body is one fused C-ABI call (``ssdk_decode_nms``)."""
from .box import decode_nms


class Decoder(object):
    r"""Decode (per level) + NMS.

    Attributes read by callers (reference pipeline_anchor_basic.py:161-163): ``conf_threshold``,
    ``nms_threshold``, ``top_n``, ``top_n_per_level``, ``rescore``, ``use_diou``.
    """

    def __init__(self, conf_threshold, nms_threshold, top_n, top_n_per_level, rescore, use_diou):
        self.conf_threshold = conf_threshold
        self.nms_threshold = nms_threshold
        self.top_n = top_n
        self.top_n_per_level = top_n_per_level
        self.rescore = rescore
        self.use_diou = use_diou

    def __call__(self, loc, conf, anchors):
        r"""
        Returns:
            out_scores (batch, top_n), out_boxes (batch, top_n, 4) ltrb, out_classes (batch, top_n);
            fp32, zero padded (reference decoder.py:25-49).
        """
        return decode_nms(
            loc, conf, anchors, self.conf_threshold, self.top_n_per_level, self.rescore,
            self.nms_threshold, self.top_n, self.use_diou,
        )
