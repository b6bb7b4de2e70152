"""VOC-style mean average precision of an eval epoch on the HIP device -- the reference's ``MeanAveragePrecision``
(``ssds/core/evaluation_metrics.py:5-142``, driven by ``pipeline_anchor_basic.py:161-182``) with the same
constructor, ``__call__(detections, targets)`` and ``get_results()`` contract.

The reference loops over images x classes in Python and appends to per-class lists; here a batch is ONE launch
(``ssdk_map_match``) that appends a (sort key, true-positive flag) record per detection slot, and ``get_results`` is
one radix sort of the epoch's records plus one launch (``ssdk_map_average_precision``).  Nothing returns to the host
before ``get_results``.  Tie contract: among same-class boxes of equal IoU the first wins (``argmax`` on CPU);
detections of equal score keep their arrival order (the reference's ``np.argsort(...)[::-1]`` leaves it undefined).
The numpy-2 breakage of the reference (``np.float`` / ``np.NAN``, :90, :126) does not apply."""
import numpy as np
import torch

from ssds import _native as N


class MeanAveragePrecision(object):
    def __init__(self, num_classes, conf_threshold, iou_threshold):
        self.num_classes = int(num_classes)
        self.conf_threshold = float(conf_threshold)
        self.iou_threshold = float(iou_threshold)
        self._keys, self._tp = [], []
        self._npos = None

    def __call__(self, detections, targets):
        """detections = (scores [B,D], boxes [B,D,4] ltrb, classes [B,D]) as returned by ``Decoder``; targets
        [B,G,5] = ltrb + label (-1 padding).  Appends this batch's records; no host synchronisation."""
        scores, boxes, classes = detections
        N.require_device(scores, "MeanAveragePrecision")
        dev = scores.device
        scores, boxes, classes = (x.contiguous().float() for x in (scores, boxes, classes))
        t = targets.to(dev).contiguous().float()
        B, D = int(scores.shape[0]), int(scores.shape[1])
        G = int(t.shape[1])
        if self._npos is None:
            self._npos = torch.zeros(self.num_classes, device=dev, dtype=torch.int32)
        keys = torch.empty((B, D), device=dev, dtype=torch.int64)
        tp = torch.empty((B, D), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            rc = N.lib.ssdk_map_match(scores.data_ptr(), boxes.data_ptr(), classes.data_ptr(), B, D, t.data_ptr(), G,
                                      self.num_classes, self.conf_threshold, self.iou_threshold, keys.data_ptr(),
                                      tp.data_ptr(), self._npos.data_ptr(), N.stream_ptr(dev))
        N.check(rc, "map_match")
        self._keys.append(keys.view(-1))
        self._tp.append(tp.view(-1))

    def _sorted(self):
        keys, tp = torch.cat(self._keys), torch.cat(self._tp)
        skeys, order = torch.sort(keys, stable=True)  # rocPRIM radix sort
        bounds = torch.arange(self.num_classes + 1, device=keys.device, dtype=torch.int64) << 32
        seg = torch.searchsorted(skeys, bounds).contiguous()
        return skeys, tp[order].contiguous(), seg

    def get_results(self):
        """(mAP, (precision, recall, ap)) like the reference: ap[c] is NaN for a class without ground truth, mAP
        the mean over the others; precision / recall are per-class numpy arrays for PR curves."""
        C = self.num_classes
        if self._npos is None:
            return float("nan"), ([[0], [0]] * C, [[0], [1]] * C, [float("nan")] * C)
        dev = self._npos.device
        skeys, tp, seg = self._sorted()
        ap = torch.empty(C, device=dev, dtype=torch.float64)
        with torch.cuda.device(dev):
            rc = N.lib.ssdk_map_average_precision(tp.data_ptr(), seg.data_ptr(), self._npos.data_ptr(), C,
                                                  ap.data_ptr(), N.stream_ptr(dev))
        N.check(rc, "map_average_precision")
        mAP = float(torch.nanmean(ap)) if bool((self._npos > 0).any()) else float("nan")
        ap_h, npos_h, seg_h, tp_h = ap.cpu().numpy(), self._npos.cpu().numpy(), seg.cpu().numpy(), tp.cpu().numpy()
        recall, precision = [], []
        for c in range(C):  # PR-curve arrays for the caller's plots (evaluation_metrics.py:124-140)
            if npos_h[c] == 0:
                recall += [[0], [1]]
                precision += [[0], [0]]
                continue
            cum = np.cumsum(tp_h[seg_h[c]:seg_h[c + 1]].astype(np.int64))
            recall.append(cum.astype(float) / float(npos_h[c]))
            precision.append(cum.astype(float) / np.arange(1, len(cum) + 1, dtype=float))
        return mAP, (precision, recall, ap_h.tolist())

    def records(self):
        """Per class, in arrival order: (scores, detect_ismatched) and the ground-truth counts -- the reference's
        ``self.score`` / ``self.detect_ismatched`` / ``self.npos`` (:10-13), for tests."""
        keys, tp = torch.cat(self._keys).cpu().numpy(), torch.cat(self._tp).cpu().numpy().astype(bool)
        cls = (keys >> 32).astype(np.int64)
        u = (~keys & 0xFFFFFFFF).astype(np.uint32)
        bits = np.where(u & 0x80000000, u & 0x7FFFFFFF, ~u).astype(np.uint32)
        score = bits.view(np.float32)
        return ([score[cls == c] for c in range(self.num_classes)], [tp[cls == c] for c in range(self.num_classes)],
                self._npos.cpu().numpy())
