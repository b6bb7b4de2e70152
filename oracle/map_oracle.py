"""CPU oracle for the eval-epoch mAP bookkeeping.

TEST INFRASTRUCTURE ONLY: only ``tests/`` may import it; the product path never does.

Plain-numpy restatement of ``MeanAveragePrecision`` (reference ``ssds/core/evaluation_metrics.py:5-142``,
ShuangXieIrene/ssds.pytorch v1.5).  Parity is PINNED: ``tests/golden/map.npz`` holds the reference's own
``detect_ismatched`` / ``score`` / ``npos`` lists and ``get_results()`` output for the seeded cases of
``tests/golden/cases.py`` (``make_golden.py:gen_map``; the reference's ``np.float`` / ``np.NAN`` (:90, :126) do not exist
in numpy 2, the generating script aliases them to ``float`` / ``np.nan`` and says so), and
``tests/test_oracle_golden.py`` checks this restatement against it.

Contract: IoU in fp32 in the reference's operation order; ``argmax`` ties resolve to the first index (CPU torch);
detections of equal score keep arrival order when ranked (the reference's ``np.argsort(...)[::-1]`` is unspecified
there; the golden cases have distinct scores per class)."""
import numpy as np

F32 = np.float32


def matrix_iou(a, b):
    """evaluation_metrics.py:16-26 (no +1 on widths, unlike box.py)."""
    a, b = a.astype(F32), b.astype(F32)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    area_i = (np.prod(rb - lt, axis=2, dtype=F32) * (lt < rb).all(axis=2)).astype(F32)
    area_a = np.prod(a[:, 2:] - a[:, :2], axis=1, dtype=F32)
    area_b = np.prod(b[:, 2:] - b[:, :2], axis=1, dtype=F32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return area_i / (area_a[:, None] + area_b[None, :] - area_i)


class MeanAveragePrecision(object):
    def __init__(self, num_classes, conf_threshold, iou_threshold):
        self.num_classes, self.conf_threshold, self.iou_threshold = num_classes, conf_threshold, iou_threshold
        self.score = [[] for _ in range(num_classes)]
        self.detect_ismatched = [[] for _ in range(num_classes)]
        self.npos = [0] * num_classes

    def __call__(self, detections, targets):
        """evaluation_metrics.py:15-61.  detections = (scores [B,D], boxes [B,D,4], classes [B,D]); targets [B,G,5]
        ltrb + label."""
        for s, bx, cl, tg in zip(*detections, targets):
            keep = s > F32(self.conf_threshold)  # :29-31
            s, bx, cl = s[keep], bx[keep], cl[keep]
            for c in range(self.num_classes):
                tc = tg[tg[:, 4] == c]  # :33
                sel = cl == c
                sc, bc = s[sel], bx[sel]
                self.npos[c] += len(tc)  # :36, :56
                if len(sc) == 0:  # :36-41
                    continue
                self.score[c] += sc.tolist()
                if len(tc) == 0:  # :42-47
                    self.detect_ismatched[c] += [False] * len(sc)
                    continue
                iou = matrix_iou(bc, tc[:, :4])
                tid = np.argmax(iou, axis=1)  # :49 first maximum
                taken = np.zeros(len(tc), bool)
                lab = np.zeros(len(sc), bool)
                for i, t in enumerate(tid):  # :52-56
                    if iou[i, t] >= F32(self.iou_threshold) and not taken[t]:
                        taken[t] = True
                        lab[i] = True
                self.detect_ismatched[c] += lab.tolist()

    def get_results(self):
        """evaluation_metrics.py:63-142 -> (mAP, ap list)."""
        ap = []
        for labels, scores, npos in zip(self.detect_ismatched, self.score, self.npos):
            if npos == 0:  # :124-128
                ap.append(np.nan)
                continue
            order = np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")  # :130-131 (tie contract above)
            tpl = np.asarray(labels, dtype=int)[order]
            tp, fp = np.cumsum(tpl), np.cumsum(1 - tpl)
            rec = tp.astype(float) / float(npos)
            prec = tp.astype(float) / np.maximum(tp + fp, np.finfo(np.float64).eps)
            if not prec.size:  # :97-98
                ap.append(0.0)
                continue
            r = np.concatenate([[0], rec, [1]])  # :104-112
            p = np.concatenate([[0], prec, [0]])
            for i in range(len(p) - 2, -1, -1):
                p[i] = max(p[i], p[i + 1])
            idx = np.where(r[1:] != r[:-1])[0] + 1
            ap.append(float(np.sum((r[idx] - r[idx - 1]) * p[idx])))
        return float(np.nanmean(ap)) if not np.all(np.isnan(ap)) else float("nan"), ap
