"""Pins the numpy oracle (oracle/box_oracle.py) to the REFERENCE: every fixture in
tests/golden/*.npz was produced by the reference's own functions (make_golden.py)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import cases
from oracle import box_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def test_anchor_kat_survey():
    # SURVEY.md section 4 known answers (computed from the reference)
    a = O.generate_anchors(8, [1, 2, 0.5], [2.0, 2.828])
    want = np.array(
        [[-4, -4, 11, 11], [-2, -8, 9, 15], [-7, -2, 14, 9],
         [-7.312, -7.312, 14.312, 14.312], [-4.484, -12.968, 11.484, 19.968],
         [-11.554, -4.484, 18.554, 11.484]], F32)
    np.testing.assert_allclose(a, want, atol=1e-3)
    b = O.generate_anchors(32, [1, 2, 0.5], [2.0, 4.0, 8.0])
    np.testing.assert_array_equal(b[0], [-16, -16, 47, 47])
    np.testing.assert_array_equal(b[1], [-7, -30, 38, 61])
    np.testing.assert_array_equal(b[8], [-164, -72, 195, 103])


def test_anchors_bit_exact():
    g = load("anchors")
    for k in g.files:
        if k.endswith("_spec"):
            continue
        spec = g[k + "_spec"]
        s, nr, ns = int(spec[0]), int(spec[1]), int(spec[2])
        r = list(spec[3:3 + nr])
        sc = list(spec[3 + nr:3 + nr + ns])
        np.testing.assert_array_equal(O.generate_anchors(s, r, sc), g[k], err_msg=k)


def test_codec():
    g = load("codec")
    d = O.box2delta(g["boxes"], g["anchors"])
    np.testing.assert_allclose(d, g["deltas"], rtol=2e-6, atol=2e-6)
    b = O.delta2box(g["d2"], g["anchors"], [32, 20], 16)
    np.testing.assert_allclose(b, g["back"], rtol=1e-5, atol=1e-3)


def test_decode_kat():
    g = load("decode")
    conf = np.zeros((1, 2, 2, 2), F32)
    conf[0, 1, 0, 1] = 0.9
    conf[0, 0, 1, 0] = 0.6
    loc = np.zeros((1, 4, 2, 2), F32)
    anc = np.array([[-4, -4, 11, 11]], F32)
    s, b, c = O.decode(conf, loc, 8, 0.05, 10, anc, False)
    np.testing.assert_array_equal(s[0, :3], F32([0.9, 0.6, 0]))
    np.testing.assert_array_equal(b[0, :2], [[4, 0, 15, 11], [0, 4, 11, 15]])
    np.testing.assert_array_equal(c[0, :2], [1, 0])
    for r in (0, 1):
        s, b, c = O.decode(conf, loc, 8, 0.05, 10, anc, bool(r))
        np.testing.assert_allclose(s, g["kat_r%d_scores" % r], atol=1e-6)
        np.testing.assert_array_equal(b, g["kat_r%d_boxes" % r])
        np.testing.assert_array_equal(c, g["kat_r%d_classes" % r])
    s, _, _ = O.decode(conf, loc, 8, 0.05, 10, anc, True)
    np.testing.assert_allclose(s[0, :2], [0.42, 0.28], atol=1e-6)


@pytest.mark.parametrize("name", list(cases.DECODE_CASES))
def test_decode_vs_reference(name):
    g = load("decode")
    d = cases.decode_inputs(name)
    anchors = cases.anchors_for(d["A"], d["stride"], O.generate_anchors)
    assert cases.checksum(d["cls"], d["box"], anchors) == g[name + "_crc"]
    s, b, c = O.decode(d["cls"], d["box"], d["stride"], d["thr"], d["top_n"], anchors, d["rescore"])
    # indices/classes bit exact; boxes within 1e-3 (north_star); scores tight
    np.testing.assert_array_equal(c, g[name + "_classes"])
    np.testing.assert_allclose(b, g[name + "_boxes"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(s, g[name + "_scores"], atol=1e-6, rtol=1e-5, equal_nan=True)


def test_nms_kat():
    s, b, c = O.nms(
        F32([[0.9, 0.8, 0.5]]), F32([[[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]]]),
        np.zeros((1, 3), F32), 0.5, 3, True)
    np.testing.assert_array_equal(s, F32([[0.9, 0.5, 0]]))
    g = load("nms")
    np.testing.assert_array_equal(s, g["kat_scores"])
    np.testing.assert_array_equal(b, g["kat_boxes"])


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_nms_vs_reference(name):
    g = load("nms")
    d = cases.nms_inputs(name)
    assert cases.checksum(d["scores"], d["boxes"], d["classes"]) == g[name + "_crc"]
    s, b, c = O.nms(d["scores"], d["boxes"], d["classes"], d["thr"], d["ndet"], d["diou"])
    np.testing.assert_array_equal(s, g[name + "_scores"])  # keep set + order bit exact
    np.testing.assert_array_equal(b, g[name + "_boxes"])
    np.testing.assert_array_equal(c, g[name + "_classes"])


@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_decoder_vs_reference(name):
    g = load("decoder")
    d = cases.decoder_inputs(name, O.generate_anchors)
    assert cases.checksum(*d["loc"], *d["conf"]) == g[name + "_crc"]
    dec = O.Decoder(d["thr"], d["nms"], d["top_n"], d["per_level"], d["rescore"], d["diou"])
    ms, mb, mc = dec.decode_levels(d["loc"], d["conf"], d["anchors"])
    np.testing.assert_array_equal(mc, g[name + "_mid_classes"])
    np.testing.assert_allclose(mb, g[name + "_mid_boxes"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(ms, g[name + "_mid_scores"], atol=1e-6, rtol=1e-5)
    # final stage on the reference's own intermediate => keep set must be bit exact
    s, b, c = O.nms(g[name + "_mid_scores"], g[name + "_mid_boxes"], g[name + "_mid_classes"],
                    d["nms"], d["top_n"], d["diou"])
    np.testing.assert_array_equal(s, g[name + "_scores"])
    np.testing.assert_array_equal(b, g[name + "_boxes"])
    np.testing.assert_array_equal(c, g[name + "_classes"])
    # end to end through the oracle only
    s, b, c = dec(d["loc"], d["conf"], d["anchors"])
    np.testing.assert_array_equal(c, g[name + "_classes"])
    np.testing.assert_allclose(b, g[name + "_boxes"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(s, g[name + "_scores"], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", list(cases.MATCH_CASES))
def test_extract_targets_vs_reference(name):
    g = load("match")
    d = cases.match_inputs(name, O.generate_anchors)
    assert cases.checksum(d["targets"], d["anchors"]) == g[name + "_crc"]
    anchors = OrderedDict([(d["stride"], d["anchors"])])
    ct, bt, dp = O.extract_targets(d["targets"], anchors, d["C"], d["stride"], d["size"],
                                   tuple(map(float, d["match"])), d["radius"])
    np.testing.assert_array_equal(dp, g[name + "_depth"])  # matching decisions bit exact
    np.testing.assert_array_equal(ct.astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt, g[name + "_box"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(cases.SCALE_MATCH_CASES))
def test_extract_targets_by_scale_vs_reference(name):
    g = load("match_scale")
    d = cases.match_inputs(name, O.generate_anchors)
    assert cases.checksum(d["targets"], d["anchors"]) == g[name + "_crc"]
    anchors = OrderedDict([(d["stride"], d["anchors"])])
    ct, bt, dp = O.extract_targets(d["targets"], anchors, d["C"], d["stride"], d["size"],
                                   [list(map(float, d["match"]))], d["radius"])
    np.testing.assert_array_equal(dp, g[name + "_depth"])  # assignment decisions bit exact
    np.testing.assert_array_equal(ct.astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt, g[name + "_box"], rtol=1e-5, atol=1e-5)


def test_match_by_scale_picks_level_range_by_stride_position():
    # box.py:389: the [lower, upper] pair of a level is match[list(anchors).index(stride)]
    d = cases.match_inputs("s_ssd300_l2", O.generate_anchors)
    other = O.generate_anchors(30, [1, 2, 0.5], [2.0, 2.828])
    anchors = OrderedDict([(30, other), (d["stride"], d["anchors"])])
    rng = list(map(float, d["match"]))
    a = O.extract_targets(d["targets"], anchors, d["C"], d["stride"], d["size"], [[9.0, 9.5], rng], 0)
    b = O.extract_targets(d["targets"], OrderedDict([(d["stride"], d["anchors"])]), d["C"], d["stride"],
                          d["size"], [rng], 0)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_match_kat_layout():
    # output index order [B, A, ., y(H), x(W)]
    d = cases.match_inputs("kat_layout", O.generate_anchors)
    np.testing.assert_array_equal(d["anchors"], [[-4, -4, 11, 11], [0, 0, 7, 7]])
    ct, bt, dp = O.extract_targets(d["targets"], OrderedDict([(8, d["anchors"])]), 7, 8, (2, 3), (0.5, 0.4))
    assert ct.shape == (1, 2, 7, 2, 3) and bt.shape == (1, 2, 4, 2, 3) and dp.shape == (1, 2, 1, 2, 3)
    np.testing.assert_array_equal(dp[0, 0, 0], [[0, 0, 0], [0, 0, 6]])
    np.testing.assert_array_equal(dp[0, 1, 0], 0)
    assert ct[0, 0, :, 1, 2].argmax() == 5 and ct[0, 0, :, 1, 2].sum() == 1
    np.testing.assert_array_equal(bt[0, 0, :, 1, 2], 0)


def test_preprocess_oracle_matches_reference_expression():
    """The oracle's front-door restatement against the literal reference expression in torch (ssds.py:53-55)."""
    import torch

    rs = np.random.RandomState(2)
    imgs = rs.randint(0, 256, (2, 9, 7, 3)).astype(np.uint8)
    t = torch.Tensor(imgs.transpose(0, 3, 1, 2))
    np.testing.assert_array_equal(O.preprocess(imgs, 0, 255), ((t - 0) / 255).numpy())
    np.testing.assert_array_equal(O.preprocess(imgs.transpose(0, 3, 1, 2), 3.0, 2.0), ((t - 3.0) / 2.0).numpy())


@pytest.mark.parametrize("name", list(cases.MAP_CASES))
def test_map_oracle_vs_reference(name):
    """oracle/map_oracle.py against the reference's MeanAveragePrecision lists and get_results()."""
    from oracle import map_oracle as MO

    g = load("map")
    d = cases.map_inputs(name)
    m = MO.MeanAveragePrecision(d["C"], d["conf_thr"], d["iou_thr"])
    for bt in d["batches"]:
        m((bt["scores"], bt["boxes"], bt["classes"]), bt["targets"])
    np.testing.assert_array_equal(np.array([len(x) for x in m.score]), g[name + "/lens"])
    np.testing.assert_array_equal(np.array([v for x in m.score for v in x], F32), g[name + "/score"])
    np.testing.assert_array_equal(np.array([v for x in m.detect_ismatched for v in x], bool), g[name + "/matched"])
    np.testing.assert_array_equal(np.array(m.npos), g[name + "/npos"])
    assert g[name + "/matched"].sum() > 0
    mAP, ap = m.get_results()
    np.testing.assert_allclose(np.array(ap), g[name + "/ap"], rtol=1e-12, atol=0, equal_nan=True)
    np.testing.assert_allclose(mAP, float(g[name + "/mAP"]), rtol=1e-12)


def test_map_oracle_properties():
    """Size-independent properties of the mAP bookkeeping (oracle/map_oracle.py): image order does not matter when
    scores are distinct, detections at or below the score threshold do not count, perfect detections give AP 1,
    splitting an epoch into batches changes nothing."""
    from oracle import map_oracle as MO

    d = cases.map_inputs("m_coco_like")
    C, ct, it = d["C"], d["conf_thr"], d["iou_thr"]

    def run(batches):
        m = MO.MeanAveragePrecision(C, ct, it)
        for bt in batches:
            m((bt["scores"], bt["boxes"], bt["classes"]), bt["targets"])
        return m.get_results()

    base_map, base_ap = run(d["batches"])
    # one image per call, in reverse order
    singles = []
    for bt in d["batches"]:
        for b in range(bt["scores"].shape[0]):
            singles.append({k: v[b:b + 1] for k, v in bt.items()})
    rev_map, rev_ap = run(singles[::-1])
    np.testing.assert_allclose(np.array(rev_ap), np.array(base_ap), rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rev_map, base_map, rtol=1e-12)
    # extra detections at the threshold are ignored (score > threshold, strictly)
    padded = []
    for bt in d["batches"]:
        B, D = bt["scores"].shape
        s = np.concatenate([bt["scores"], np.full((B, 5), ct, F32)], 1)
        bx = np.concatenate([bt["boxes"], np.tile(np.array([0, 0, 50, 50], F32), (B, 5, 1))], 1)
        cl = np.concatenate([bt["classes"], np.zeros((B, 5), F32)], 1)
        padded.append(dict(scores=s, boxes=bx, classes=cl, targets=bt["targets"]))
    pad_map, _ = run(padded)
    assert pad_map == base_map
    # perfect detections
    perfect = []
    for bt in d["batches"]:
        t = bt["targets"]
        valid = t[..., 4] >= 0
        s = np.where(valid, np.linspace(0.9, 0.5, t.shape[1], dtype=F32)[None, :], 0).astype(F32)
        perfect.append(dict(scores=s, boxes=t[..., :4].copy(), classes=np.where(valid, t[..., 4], 0).astype(F32), targets=t))
    p_map, p_ap = run(perfect)
    assert abs(p_map - 1.0) < 1e-12 and all(np.isnan(a) or abs(a - 1.0) < 1e-12 for a in p_ap)
    assert 0.0 <= base_map <= 1.0
