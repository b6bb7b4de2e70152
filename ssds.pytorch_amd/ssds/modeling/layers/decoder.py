"""``Decoder`` -- same constructor/attributes/call contract as the reference's
``ssds/modeling/layers/decoder.py:15-49``; the body is one fused C-ABI call (``ssdk_decode_nms``)."""
from ssds import _native as N

from .box import _TailPipe, decode_nms


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
        self._tail = None
        self._ctx = {}  # device index -> N.Context: this decoder's HIP objects (never shared with another decoder)
        self._prof = False

    def context(self, device):
        """The ``ssdk_ctx`` this decoder uses on ``device`` (created on first use)."""
        c = self._ctx.get(device.index)
        if c is None:
            c = self._ctx[device.index] = N.Context(device)
            if self._prof:
                c.set_profiling(self._prof)
        return c

    def set_profiling(self, on):
        """Record hipEvents around the stage's launches (read back with ``timings_ms``; bench.py)."""
        self._prof = 2 if on == 2 else bool(on)
        for c in self._ctx.values():
            c.set_profiling(self._prof)

    def timings_ms(self, back=0, device=None):
        c = next(iter(self._ctx.values())) if device is None else self._ctx[device.index]
        return c.timings_ms(back)

    def enable_tail_stream(self, stream=None):
        """Serving-loop mode: the latency-bound end of the stage (per-level merge/sort/decode and NMS, 64-384
        workgroups) runs on its own HIP stream and overlaps the NEXT batch's forward pass; the HBM-bound scan stays on
        the caller's stream.  The returned tensors are then complete on that stream: call ``wait()`` (or synchronise
        the device) before reading them from another stream.  Off by default."""
        self._tail = _TailPipe(stream)
        return self

    def disable_tail_stream(self):
        if self._tail is not None:
            self._tail.wait()
        self._tail = None

    def wait(self):
        if self._tail is not None:
            self._tail.wait()

    def __call__(self, loc, conf, anchors):
        r"""
        Returns:
            out_scores (batch, top_n), out_boxes (batch, top_n, 4) ltrb, out_classes (batch, top_n);
            fp32, zero padded (reference decoder.py:25-49).
        """
        return decode_nms(
            loc, conf, anchors, self.conf_threshold, self.top_n_per_level, self.rescore,
            self.nms_threshold, self.top_n, self.use_diou, tail=self._tail, ctx=self.context(conf[0].device),
        )
