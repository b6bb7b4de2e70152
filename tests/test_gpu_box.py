"""Parity of the HIP kernels (through the C-ABI) with the reference fixtures and the pinned oracle.

Bars (north_star): anchor indices / classes / NMS keep sets bit-exact, decoded box coordinates within
1e-3, scores within 1e-5 relative."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import cases
from oracle import box_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32
BOX_ATOL = 1e-3


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def dev_heads(d):
    import torch

    if d["dtype"] == "f32":
        return torch.from_numpy(d["cls"]).cuda(), torch.from_numpy(d["box"]).cuda()
    tdt = torch.bfloat16 if d["dtype"] == "bf16" else torch.float16
    c = torch.from_numpy(d["raw_cls"].view(np.int16)).cuda().view(tdt)
    b = torch.from_numpy(d["raw_box"].view(np.int16)).cuda().view(tdt)
    return c, b


def cmp_decode(got, want, what, rescore=True):
    """classes (== flat indices of the winners) bit exact, boxes within 1e-3.  Raw scores are copies and
    must be bit exact; centre-rescored scores are score*sqrt(min/max ratios) of the decoded box, which
    amplifies the <=1-ulp expf difference between device and host when a box edge nearly touches the
    anchor centre, hence 1e-4 on those."""
    s, b, c = (t.cpu().numpy() for t in got)
    ws, wb, wc = want
    np.testing.assert_array_equal(c, wc, err_msg=what + " classes")
    np.testing.assert_allclose(b, wb, atol=BOX_ATOL, rtol=0, err_msg=what + " boxes")
    if rescore:
        np.testing.assert_allclose(s, ws, atol=1e-4, rtol=1e-4, equal_nan=True, err_msg=what + " scores")
    else:
        np.testing.assert_array_equal(s, ws, err_msg=what + " scores")


@pytest.mark.parametrize("name", list(cases.DECODE_CASES))
def test_decode_vs_reference_and_oracle(name):
    import torch
    from ssds.modeling.layers import box

    g = load("decode")
    d = cases.decode_inputs(name)
    anchors = cases.anchors_for(d["A"], d["stride"], O.generate_anchors)
    cls, loc = dev_heads(d)
    got = box.decode(cls, loc, d["stride"], d["thr"], d["top_n"], torch.from_numpy(anchors), d["rescore"])
    assert all(t.dtype == torch.float32 for t in got)
    cmp_decode(got, (g[name + "_scores"], g[name + "_boxes"], g[name + "_classes"]), name + " vs reference",
               d["rescore"])
    want = O.decode(d["cls"], d["box"], d["stride"], d["thr"], d["top_n"], anchors, d["rescore"])
    cmp_decode(got, want, name + " vs oracle", d["rescore"])


def test_decode_kat():
    import torch
    from ssds.modeling.layers import box

    conf = torch.zeros(1, 2, 2, 2)
    conf[0, 1, 0, 1] = 0.9
    conf[0, 0, 1, 0] = 0.6
    loc = torch.zeros(1, 4, 2, 2)
    anc = torch.tensor([[-4.0, -4, 11, 11]])
    s, b, c = box.decode(conf.cuda(), loc.cuda(), 8, 0.05, 10, anc, False)
    np.testing.assert_array_equal(s.cpu().numpy()[0, :3], F32([0.9, 0.6, 0]))
    np.testing.assert_array_equal(b.cpu().numpy()[0, :2], [[4, 0, 15, 11], [0, 4, 11, 15]])
    np.testing.assert_array_equal(c.cpu().numpy()[0, :2], [1, 0])
    s, _, _ = box.decode(conf.cuda(), loc.cuda(), 8, 0.05, 10, anc, True)
    np.testing.assert_allclose(s.cpu().numpy()[0, :2], [0.42, 0.28], atol=1e-6)


def _tie_case(kind, n_img, A, C, H, W, seed):
    rs = np.random.RandomState(seed)
    n = A * C * H * W
    if kind == "constant":
        cls = np.full((n_img, n), 0.5, F32)
    elif kind == "ramp_up":  # every new element beats everything seen so far: worst case for pruning
        cls = np.tile(np.linspace(0.02, 0.98, n, dtype=F32), (n_img, 1))
    elif kind == "ramp_down":
        cls = np.tile(np.linspace(0.98, 0.02, n, dtype=F32), (n_img, 1))
    elif kind == "quantized":  # few distinct values -> massive ties at the cut
        cls = (rs.randint(0, 16, size=(n_img, n)) / 16.0).astype(F32)
    elif kind == "sparse":  # almost nothing passes
        cls = np.zeros((n_img, n), F32)
        for b in range(n_img):
            hot = rs.choice(n, size=37, replace=False)
            cls[b, hot] = (0.1 + 0.8 * rs.random_sample(37)).astype(F32)
    else:
        raise ValueError(kind)
    box = (rs.standard_normal((n_img, A * 4 * H * W)) * 0.5).astype(F32)
    return cls.reshape(n_img, A * C, H, W), box.reshape(n_img, A * 4, H, W)


@pytest.mark.parametrize("kind", ["constant", "ramp_up", "ramp_down", "quantized", "sparse"])
@pytest.mark.parametrize("tpu", ["1", "4", "0"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_decode_tie_contract_and_multi_unit(kind, tpu, dtype, monkeypatch):
    """Ties, monotone ramps and multi-unit merges against the oracle's (score desc, index asc) contract.
    SSDK_TILES_PER_UNIT=1/4 forces many units per (image, level) so that the merge path is exercised."""
    import torch
    from ssds.modeling.layers import box

    monkeypatch.setenv("SSDK_TILES_PER_UNIT", tpu)
    A, C, H, W, stride = 3, 20, 24, 28, 8
    cls, loc = _tie_case(kind, 2, A, C, H, W, 7)
    if dtype == "bf16":
        tc = torch.from_numpy(cls).to(torch.bfloat16)
        tl = torch.from_numpy(loc).to(torch.bfloat16)
        cls, loc = tc.float().numpy(), tl.float().numpy()
    else:
        tc, tl = torch.from_numpy(cls), torch.from_numpy(loc)
    anchors = O.generate_anchors(stride, [1, 2, 0.5], [2.0])
    for top_n, thr, rescore in ((300, 0.05, True), (64, 0.3, False), (1000, 0.01, False)):
        got = box.decode(tc.cuda(), tl.cuda(), stride, thr, top_n, torch.from_numpy(anchors), rescore)
        want = O.decode(cls, loc, stride, thr, top_n, anchors, rescore)
        cmp_decode(got, want, "%s tpu=%s %s top_n=%d" % (kind, tpu, dtype, top_n), rescore)


def test_decode_unaligned_images():
    """n = A*C*H*W not a multiple of the 16-byte vector: images start misaligned, tail is partial."""
    import torch
    from ssds.modeling.layers import box

    rs = np.random.RandomState(3)
    for (A, C, H, W) in ((3, 5, 7, 9), (1, 3, 5, 7), (2, 7, 3, 3), (1, 1, 1, 1), (3, 1, 1, 3)):
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            n = A * C * H * W
            cls = torch.from_numpy(rs.random_sample((5, A * C, H, W)).astype(F32)).to(dtype)
            loc = torch.from_numpy((rs.standard_normal((5, A * 4, H, W)) * 0.3).astype(F32)).to(dtype)
            anchors = cases.anchors_for(A, 16, O.generate_anchors)
            cf = cases.make_unique_above(cls.float().numpy(), 0.0) if dtype == torch.float32 else cls.float().numpy()
            if dtype == torch.float32:
                cls = torch.from_numpy(cf)
            got = box.decode(cls.cuda(), loc.cuda(), 16, 0.2, 40, torch.from_numpy(anchors), True)
            want = O.decode(cf, loc.float().numpy(), 16, 0.2, 40, anchors, True)
            cmp_decode(got, want, "unaligned n=%d %s" % (n, dtype))


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_nms_vs_reference(name):
    import torch
    from ssds.modeling.layers import box

    g = load("nms")
    d = cases.nms_inputs(name)
    s, b, c = box.nms(torch.from_numpy(d["scores"]).cuda(), torch.from_numpy(d["boxes"]).cuda(),
                      torch.from_numpy(d["classes"]).cuda(), d["thr"], d["ndet"], d["diou"])
    np.testing.assert_array_equal(s.cpu().numpy(), g[name + "_scores"])  # keep set + order bit exact
    np.testing.assert_array_equal(b.cpu().numpy(), g[name + "_boxes"])
    np.testing.assert_array_equal(c.cpu().numpy(), g[name + "_classes"])


def test_nms_kat_and_ties():
    import torch
    from ssds.modeling.layers import box

    s, b, c = box.nms(torch.tensor([[0.9, 0.8, 0.5]]).cuda(),
                      torch.tensor([[[0.0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]]]).cuda(),
                      torch.zeros(1, 3).cuda(), 0.5, 3, True)
    np.testing.assert_array_equal(s.cpu().numpy(), F32([[0.9, 0.5, 0]]))
    # equal scores: stable order (position ascending), oracle contract
    rs = np.random.RandomState(9)
    d = cases.nms_inputs("clustered_diou")
    sc = (np.round(d["scores"] * 8) / 8).astype(F32)
    got = box.nms(torch.from_numpy(sc).cuda(), torch.from_numpy(d["boxes"]).cuda(),
                  torch.from_numpy(d["classes"]).cuda(), 0.5, 100, True)
    want = O.nms(sc, d["boxes"], d["classes"], 0.5, 100, True)
    for g_, w_ in zip(got, want):
        np.testing.assert_array_equal(g_.cpu().numpy(), w_)


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_every_decoder_golden_through_the_16_bit_scan(name, dtype_name):
    """The reference fixtures are fp32 and therefore exercise scan_kernel; every BASELINE configuration runs 16-bit heads
    through scan16_kernel (packed 16-bit compares, sampled cut, vector queues).  Each Decoder fixture's inputs, rounded to
    bf16 / fp16, through the device stage (scan16 -> levelsel -> nmswalk) against the oracle on the SAME rounded heads:
    per-level output and final detections, classes / order bit-exact (the rounding creates ties: the (score desc, flat
    index asc) contract decides them on both sides)."""
    import torch
    from ssds.modeling.layers.box import decode_nms

    d = cases.decoder_inputs(name, O.generate_anchors)
    if not (d["thr"] > 0 and d["per_level"] <= 512):
        pytest.skip("the 16-bit scan needs a positive threshold and K <= 512")
    tdt = getattr(torch, dtype_name)
    loc = [torch.from_numpy(x).to(tdt) for x in d["loc"]]
    conf = [torch.from_numpy(x).to(tdt) for x in d["conf"]]
    anchors = OrderedDict((k, torch.from_numpy(v)) for k, v in d["anchors"].items())
    (s, b, c), mid = decode_nms([t.cuda() for t in loc], [t.cuda() for t in conf], anchors, d["thr"], d["per_level"], d["rescore"],
                                d["nms"], d["top_n"], d["diou"], return_mid=True)
    odec = O.Decoder(d["thr"], d["nms"], d["top_n"], d["per_level"], d["rescore"], d["diou"])
    ol, oc = [t.float().numpy() for t in loc], [t.float().numpy() for t in conf]
    wm = odec.decode_levels(ol, oc, d["anchors"])
    np.testing.assert_array_equal(mid[2].cpu().numpy(), wm[2])
    np.testing.assert_allclose(mid[1].cpu().numpy(), wm[1], atol=BOX_ATOL, rtol=0)
    np.testing.assert_allclose(mid[0].cpu().numpy(), wm[0], atol=1e-4, rtol=1e-4, equal_nan=True)
    # NMS on the device's own per-level output (bit-identical input on both sides: ulp-level differences of the rescored
    # scores must not decide the walk order of tied candidates)
    wn = O.nms(mid[0].cpu().numpy(), mid[1].cpu().numpy(), mid[2].cpu().numpy(), d["nms"], d["top_n"], d["diou"])
    for g_, w_ in zip((s, b, c), wn):
        np.testing.assert_array_equal(g_.cpu().numpy(), w_)


@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_decoder_vs_reference(name):
    import torch
    from ssds.modeling.layers.box import decode_nms
    from ssds.modeling.layers.decoder import Decoder

    g = load("decoder")
    d = cases.decoder_inputs(name, O.generate_anchors)
    loc = [torch.from_numpy(x).cuda() for x in d["loc"]]
    conf = [torch.from_numpy(x).cuda() for x in d["conf"]]
    anchors = OrderedDict((k, torch.from_numpy(v)) for k, v in d["anchors"].items())
    dec = Decoder(d["thr"], d["nms"], d["top_n"], d["per_level"], d["rescore"], d["diou"])
    s, b, c = dec(loc, conf, anchors)
    np.testing.assert_array_equal(c.cpu().numpy(), g[name + "_classes"])
    np.testing.assert_allclose(b.cpu().numpy(), g[name + "_boxes"], atol=BOX_ATOL, rtol=0)
    np.testing.assert_allclose(s.cpu().numpy(), g[name + "_scores"], atol=1e-4, rtol=1e-4)
    # the concatenated per-level decode the reference materialises (decoder.py:48)
    (_, _, _), mid = decode_nms(loc, conf, anchors, d["thr"], d["per_level"], d["rescore"], d["nms"],
                                d["top_n"], d["diou"], return_mid=True)
    np.testing.assert_array_equal(mid[2].cpu().numpy(), g[name + "_mid_classes"])
    np.testing.assert_allclose(mid[1].cpu().numpy(), g[name + "_mid_boxes"], atol=BOX_ATOL, rtol=0)
    np.testing.assert_allclose(mid[0].cpu().numpy(), g[name + "_mid_scores"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("name", list(cases.MATCH_CASES))
def test_extract_targets_vs_reference(name):
    import torch
    from ssds.modeling.layers import box

    g = load("match")
    d = cases.match_inputs(name, O.generate_anchors)
    anchors = OrderedDict([(d["stride"], torch.from_numpy(d["anchors"]))])
    ct, bt, dp = box.extract_targets(torch.from_numpy(d["targets"]).cuda(), anchors, d["C"], d["stride"],
                                     d["size"], list(map(float, d["match"])), d["radius"])
    np.testing.assert_array_equal(dp.cpu().numpy(), g[name + "_depth"])  # matching decisions bit exact
    np.testing.assert_array_equal(ct.cpu().numpy().astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt.cpu().numpy(), g[name + "_box"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(cases.SCALE_MATCH_CASES))
def test_extract_targets_by_scale_vs_reference(name):
    import torch
    from ssds.modeling.layers import box

    g = load("match_scale")
    d = cases.match_inputs(name, O.generate_anchors)
    anchors = OrderedDict([(d["stride"], torch.from_numpy(d["anchors"]))])
    ct, bt, dp = box.extract_targets(torch.from_numpy(d["targets"]).cuda(), anchors, d["C"], d["stride"],
                                     d["size"], [list(map(float, d["match"]))], d["radius"])
    np.testing.assert_array_equal(dp.cpu().numpy(), g[name + "_depth"])  # assignment decisions bit exact
    np.testing.assert_array_equal(ct.cpu().numpy().astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt.cpu().numpy(), g[name + "_box"], rtol=1e-5, atol=1e-5)


def test_extract_targets_by_scale_full_size_vs_oracle():
    """SSD-MobileNetV2@512 level 0 (32x32, A=6, C=80), B=64, G=32: whole batch in one launch vs the oracle
    on a sample of images; every image obeys the structural properties (one-hot rows, depth <-> class)."""
    import torch
    from ssds.modeling.layers import box

    rs = np.random.RandomState(77)
    B, G, C, H, W, stride = 64, 32, 80, 32, 32, 16
    anc = O.generate_anchors(stride, [1, 2, 0.5], [2.0, 2.828])
    targets = np.full((B, G, 5), -1, np.float32)
    for b in range(B):
        n = rs.randint(0, G + 1)
        wh = np.ceil(rs.uniform(0.02 * 512, 0.4 * 512, (n, 2)))
        xy = np.floor(rs.uniform(0, 0.8 * 512, (n, 2)))
        wh = np.minimum(wh, 512 - xy)
        targets[b, :n] = np.concatenate([xy, wh, rs.randint(0, C, (n, 1))], 1)
    anchors = OrderedDict([(stride, torch.from_numpy(anc))])
    for rng, radius in (([0.5, 4.0], 0), ([0.25, 3.0], 1.5)):
        ct, bt, dp = box.extract_targets(torch.from_numpy(targets).cuda(), anchors, C, stride, (H, W), [rng], radius)
        ctn, btn, dpn = ct.cpu().numpy(), bt.cpu().numpy(), dp.cpu().numpy()
        assert ((ctn == 0) | (ctn == 1)).all()
        onehot = ctn.sum(2, keepdims=True)
        np.testing.assert_array_equal(onehot, (dpn > 0).astype(np.float32))  # fg <=> exactly one class bit
        lab = ctn.argmax(2)[:, :, None]
        np.testing.assert_array_equal(np.where(dpn > 0, lab + 1, 0), dpn)
        for b in (0, 7, 63):
            oc, ob, od = O.extract_targets(targets[b:b + 1], OrderedDict([(stride, anc)]), C, stride, (H, W), [rng], radius)
            np.testing.assert_array_equal(dpn[b:b + 1], od)
            np.testing.assert_array_equal(ctn[b:b + 1], oc)
            np.testing.assert_allclose(btn[b:b + 1], ob, rtol=1e-5, atol=1e-5)


def test_full_size_ssd512_bf16_properties_and_sampled_oracle():
    """BASELINE config 2 shape (SSD-MobileNetV2@512, B=64, bf16 heads): size-independent properties over
    the whole batch + the oracle on a sample of images."""
    import torch
    from ssds.modeling.layers.decoder import Decoder

    torch.manual_seed(1234)
    B, A, C = 64, 6, 80
    maps, strides = [32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512]
    conf = [torch.sigmoid(torch.randn(B, A * C, m, m, device="cuda") * 1.5 - 4.6).to(torch.bfloat16) for m in maps]
    loc = [(torch.randn(B, A * 4, m, m, device="cuda") * 0.5).to(torch.bfloat16) for m in maps]
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828]))) for s in strides)
    dec = Decoder(0.01, 0.6, 100, 300, True, True)
    s, b, c = dec(loc, conf, anchors)
    s2, b2, c2 = dec(loc, conf, anchors)  # deterministic (no atomics-order dependence in the result)
    assert torch.equal(s, s2) and torch.equal(b, b2) and torch.equal(c, c2)
    sn, bn, cn = s.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    assert sn.shape == (B, 100) and bn.shape == (B, 100, 4)
    assert np.all(np.diff(sn, axis=1) <= 0), "scores must be sorted descending"
    assert np.all((bn >= 0) & (bn <= 511)) and np.all(bn[..., 2:] >= bn[..., :2] - 1)
    assert np.all((cn >= 0) & (cn < C) & (cn == np.round(cn)))
    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    odec = O.Decoder(0.01, 0.6, 100, 300, True, True)
    for img in (0, 17, 63):
        w = odec([l[img:img + 1].float().cpu().numpy() for l in loc],
                 [x[img:img + 1].float().cpu().numpy() for x in conf], oanch)
        # bf16 heads tie heavily: the contract (score desc, index asc) makes this exact
        np.testing.assert_array_equal(cn[img:img + 1], w[2])
        np.testing.assert_allclose(bn[img:img + 1], w[1], atol=BOX_ATOL, rtol=0)
        np.testing.assert_allclose(sn[img:img + 1], w[0], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("src", ["u8_nhwc", "u8_nchw", "f32_nhwc", "f32_nchw"])
@pytest.mark.parametrize("dst", ["float32", "bfloat16", "float16"])
def test_preprocess_matches_reference_expression(src, dst):
    """ssdk_preprocess == torch.Tensor(imgs) -> (x - mean) / std -> .to(dtype) (ssds.py:53-55), bit for bit."""
    import torch
    from ssds.ssds import preprocess

    rs = np.random.RandomState(5)
    n, h, w = 3, 37, 53  # odd width: partial 8-pixel groups
    shape = (n, h, w, 3) if src.endswith("nhwc") else (n, 3, h, w)
    raw = rs.randint(0, 256, shape).astype(np.uint8) if src.startswith("u8") else (rs.rand(*shape) * 255).astype(np.float32)
    for mean, std in ((0, 255), ([123.675, 116.28, 103.53], [58.395, 57.12, 57.375])):
        want = torch.from_numpy(O.preprocess(raw, mean, std)).to(getattr(torch, dst))
        got = preprocess(torch.from_numpy(raw).cuda(), mean, std, getattr(torch, dst))
        assert got.shape == want.shape and got.is_contiguous()
        assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("name", list(cases.MAP_CASES))
def test_mean_average_precision_vs_reference(name):
    """ssdk_map_match + ssdk_map_average_precision (ssds.core.evaluation_metrics.MeanAveragePrecision) against the
    reference's per-class lists (bit-exact flags, scores and counts) and AP / mAP (fp64, 1e-12)."""
    import torch
    from ssds.core.evaluation_metrics import MeanAveragePrecision

    g = load("map")
    d = cases.map_inputs(name)
    m = MeanAveragePrecision(d["C"], d["conf_thr"], d["iou_thr"])
    for bt in d["batches"]:
        m(tuple(torch.from_numpy(bt[k]).cuda() for k in ("scores", "boxes", "classes")),
          torch.from_numpy(bt["targets"]).cuda())
    score, matched, npos = m.records()
    np.testing.assert_array_equal(np.array([len(x) for x in score]), g[name + "/lens"])
    np.testing.assert_array_equal(np.concatenate(score).astype(F32), g[name + "/score"])
    np.testing.assert_array_equal(np.concatenate(matched), g[name + "/matched"])
    np.testing.assert_array_equal(npos, g[name + "/npos"])
    mAP, (prec, rec, ap) = m.get_results()
    np.testing.assert_allclose(np.array(ap), g[name + "/ap"], rtol=1e-12, atol=1e-15, equal_nan=True)
    np.testing.assert_allclose(mAP, float(g[name + "/mAP"]), rtol=1e-12)
    assert len(prec) == len(rec)


def test_mean_average_precision_full_size_vs_oracle():
    """COCO-sized eval batches (B=64, 100 detections, 80 classes, 10 batches) against the numpy oracle, plus the
    size-independent properties: perfect detections give mAP 1, no detections give 0."""
    import torch
    from oracle import map_oracle as MO
    from ssds.core.evaluation_metrics import MeanAveragePrecision

    rs = np.random.RandomState(9)
    B, D, Gm, C = 64, 100, 30, 80
    m, o = MeanAveragePrecision(C, 0.01, 0.5), MO.MeanAveragePrecision(C, 0.01, 0.5)
    pool = rs.permutation(4000000)[: 10 * B * D]
    for it in range(10):
        tg = np.full((B, Gm, 5), -1, F32)
        sc = np.zeros((B, D), F32)
        bx = np.zeros((B, D, 4), F32)
        cl = np.zeros((B, D), F32)
        for b in range(B):
            g = rs.randint(0, Gm + 1)
            xy = rs.random_sample((g, 2)) * 400
            wh = 16 + rs.random_sample((g, 2)) * 150
            tg[b, :g] = np.concatenate([xy, xy + wh, rs.randint(0, C, (g, 1))], 1)
            n = rs.randint(0, D + 1)
            s = np.sort((0.011 + 0.98 * pool[(it * B + b) * D:(it * B + b) * D + n] / 4e6).astype(F32))[::-1]
            src = rs.randint(0, max(g, 1), n)
            hit = (rs.random_sample(n) < 0.6) & (g > 0)
            jit = (rs.random_sample((n, 4)) - 0.5) * 40
            rnd_xy = rs.random_sample((n, 2)) * 400
            rnd = np.concatenate([rnd_xy, rnd_xy + 16 + rs.random_sample((n, 2)) * 150], 1)
            sc[b, :n] = s
            bx[b, :n] = np.where(hit[:, None], tg[b, src, :4] + jit, rnd)
            cl[b, :n] = np.where(hit, tg[b, src, 4], rs.randint(0, C, n))
        m(tuple(torch.from_numpy(x).cuda() for x in (sc, bx, cl)), torch.from_numpy(tg).cuda())
        o((sc, bx, cl), tg)
    score, matched, npos = m.records()
    np.testing.assert_array_equal(npos, np.array(o.npos))
    for c in range(C):
        np.testing.assert_array_equal(score[c], np.array(o.score[c], F32))
        np.testing.assert_array_equal(matched[c], np.array(o.detect_ismatched[c], bool))
    mAP, (_, _, ap) = m.get_results()
    omAP, oap = o.get_results()
    np.testing.assert_allclose(np.array(ap), np.array(oap), rtol=1e-12, atol=1e-15, equal_nan=True)
    np.testing.assert_allclose(mAP, omAP, rtol=1e-12)
    assert 0.05 < mAP < 0.95

    tg_t = torch.from_numpy(tg).cuda()
    perfect = MeanAveragePrecision(C, 0.01, 0.5)
    ps = torch.linspace(0.9, 0.5, Gm).repeat(B, 1).cuda() * (tg_t[..., 4] >= 0)
    perfect((ps, tg_t[..., :4].contiguous(), tg_t[..., 4].clamp(min=0).contiguous()), tg_t)
    assert abs(perfect.get_results()[0] - 1.0) < 1e-12
    nothing = MeanAveragePrecision(C, 0.01, 0.5)
    nothing((torch.zeros(B, D).cuda(), torch.zeros(B, D, 4).cuda(), torch.zeros(B, D).cuda()), tg_t)
    assert nothing.get_results()[0] == 0.0


def test_decoder_tail_stream_matches_inline_decode():
    """Decoder.enable_tail_stream(): level + NMS kernels on their own stream, two alternating workspaces; a loop of
    different batches with other work enqueued in between must give exactly the inline results."""
    import torch
    from ssds.modeling.layers import box
    from ssds.modeling.layers.decoder import Decoder

    torch.manual_seed(4)
    B, A, C = 8, 6, 20
    sizes, strides = [16, 8, 4, 2], [16, 32, 64, 128]
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
    batches = []
    for it in range(7):
        conf = [torch.rand(B, A * C, h, h, device="cuda").pow(4).to(torch.bfloat16) for h in sizes]
        loc = [torch.randn(B, A * 4, h, h, device="cuda").mul(0.2).to(torch.bfloat16) for h in sizes]
        batches.append((loc, conf))
    inline = Decoder(0.05, 0.5, 50, 100, True, True)
    want = [tuple(t.clone() for t in inline(l, c, anchors)) for l, c in batches]
    piped = Decoder(0.05, 0.5, 50, 100, True, True).enable_tail_stream()
    filler = torch.randn(2048, 2048, device="cuda")
    got = []
    for l, c in batches:
        # fresh copies that die right after the call: their memory may only be recycled after the tail work
        l2, c2 = [t.clone() for t in l], [t.clone() for t in c]
        got.append(piped(l2, c2, anchors))
        del l2, c2
        filler = filler @ filler.t() * 1e-3  # main-stream work the tail overlaps with (and allocator churn)
        junk = [torch.full((B, A * C, h, h), 0.9, device="cuda", dtype=torch.bfloat16) for h in sizes]
        del junk
    piped.wait()
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a, b)
    piped.disable_tail_stream()
    again = piped(*batches[0], anchors)
    for a, b in zip(again, want[0]):
        assert torch.equal(a, b)


def test_extract_targets_by_iou_full_size_vs_oracle():
    """BASELINE config 4 geometry (SSD-MobileNetV2@512 training: B=64, A=6, C=80, G=32 COCO-shaped targets of SURVEY
    8d), IoU matching [0.5, 0.4] -- the branch every BASELINE config uses -- on every level: whole batch in one launch
    vs the oracle on sampled images; all images obey the structural properties (exactly one class bit wherever the
    best overlap reaches the unmatch threshold -- the ignore band keeps its class bit, box.py:201-203 only clears the
    background -- the bit of a foreground anchor is its depth - 1, box targets finite)."""
    import torch
    from ssds.modeling.layers import box

    rs = np.random.RandomState(78)
    B, G, C, S = 64, 32, 80, 512
    targets = np.full((B, G, 5), -1, np.float32)
    for b in range(B):
        n = rs.randint(1, G + 1) if b != 5 else 0  # one image without ground truth
        wh = np.ceil(rs.uniform(0.02 * S, 0.4 * S, (n, 2)))
        xy = np.floor(rs.uniform(0, 0.8 * S, (n, 2)))
        wh = np.minimum(wh, S - xy)
        targets[b, :n] = np.concatenate([xy, wh, rs.randint(0, C, (n, 1))], 1)
    tdev = torch.from_numpy(targets).cuda()
    for (m, stride) in zip([32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512]):
        anc = O.generate_anchors(stride, [1, 2, 0.5], [2.0, 2.828])
        anchors = OrderedDict([(stride, torch.from_numpy(anc))])
        ct, bt, dp = box.extract_targets(tdev, anchors, C, stride, (m, m), [0.5, 0.4], 0)
        ctn, btn, dpn = ct.cpu().numpy(), bt.cpu().numpy(), dp.cpu().numpy()
        assert ctn.shape == (B, 6, C, m, m) and btn.shape == (B, 6, 4, m, m) and dpn.shape == (B, 6, 1, m, m)
        assert ((ctn == 0) | (ctn == 1)).all() and np.isfinite(btn).all()
        np.testing.assert_array_equal(ctn.sum(2, keepdims=True), (dpn != 0).astype(np.float32))
        lab = ctn.argmax(2)[:, :, None]
        np.testing.assert_array_equal(np.where(dpn > 0, lab + 1, dpn), dpn)
        assert set(np.unique(dpn[dpn <= 0])) <= {-1.0, 0.0}
        assert (dpn[5] == 0).all() and (ctn[5] == 0).all()  # no ground truth: everything is background
        if m >= 8:
            assert (dpn > 0).any() and (dpn < 0).any(), "the case must exercise foreground and the ignore band"
        for b in (0, 5, 31, 63):
            oc, ob, od = O.extract_targets(targets[b:b + 1], OrderedDict([(stride, anc)]), C, stride, (m, m),
                                           [0.5, 0.4], 0)
            np.testing.assert_array_equal(dpn[b:b + 1], od)
            np.testing.assert_array_equal(ctn[b:b + 1], oc)
            np.testing.assert_allclose(btn[b:b + 1], ob, rtol=1e-5, atol=1e-5)


FULL_SHAPES = {
    # BASELINE config 3: FPN + ResNet50 @640, batch 32, A = 9 (6.138 M scores per image)
    "fpn640": dict(B=32, A=9, C=80, size=640, maps=[80, 40, 20, 10, 5], strides=[8, 16, 32, 64, 128]),
    # BASELINE config 5: BiFPN + RegNetX-800MF @896, 16 images per GPU, A = 9 (12.03 M scores per image)
    "bifpn896": dict(B=16, A=9, C=80, size=896, maps=[112, 56, 28, 14, 7], strides=[8, 16, 32, 64, 128]),
}


@pytest.mark.parametrize("shape,dtype", [("fpn640", "bfloat16"), ("fpn640", "float16"), ("bifpn896", "float16"),
                                         ("bifpn896", "bfloat16")])
def test_full_size_fpn_bifpn_shapes_properties_and_sampled_oracle(shape, dtype):
    """BASELINE config 3 / 5 head shapes (A = 9, five levels, up to 9 M scores per (image, level): many scan units per
    level and multi-unit merges) in the dtypes the configs name: size-independent properties over the whole batch +
    the oracle on sampled images, through the mid (per-level decode) outputs as well."""
    import torch
    from ssds.modeling.layers.box import decode_nms

    cfg = FULL_SHAPES[shape]
    tdt = getattr(torch, dtype)
    B, A, C, S = cfg["B"], cfg["A"], cfg["C"], cfg["size"]
    torch.manual_seed(4321)
    conf = [torch.sigmoid(torch.randn(B, A * C, m, m, device="cuda") * 1.5 - 4.6).to(tdt) for m in cfg["maps"]]
    loc = [(torch.randn(B, A * 4, m, m, device="cuda") * 0.5).to(tdt) for m in cfg["maps"]]
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.52, 3.175])))
                          for s in cfg["strides"])
    args = (0.01, 300, True, 0.6, 100, True)
    (s, b, c), mid = decode_nms(loc, conf, anchors, *args, return_mid=True)
    (s2, b2, c2), mid2 = decode_nms(loc, conf, anchors, *args, return_mid=True)
    for x, y in zip((s, b, c) + tuple(mid), (s2, b2, c2) + tuple(mid2)):
        assert torch.equal(x, y), "decode + NMS must be deterministic"
    sn, bn, cn = s.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    ms, mb, mc = (t.cpu().numpy() for t in mid)
    L = len(cfg["maps"])
    assert sn.shape == (B, 100) and ms.shape == (B, L * 300) and mb.shape == (B, L * 300, 4)
    assert np.all(np.diff(sn, axis=1) <= 0) and np.all(sn[:, 0] > 0)
    assert np.all((bn >= 0) & (bn <= S - 1)) and np.all((mb >= 0) & (mb <= S - 1))
    assert np.all((cn >= 0) & (cn < C) & (cn == np.round(cn)))
    # every final detection is one of the image's per-level candidates (score, box and class together)
    for img in range(B):
        cand = {(float(a), float(k)) + tuple(map(float, bx)) for a, k, bx in zip(ms[img], mc[img], mb[img])}
        n = int((sn[img] > 0).sum())
        assert all(((float(sn[img, i]), float(cn[img, i])) + tuple(map(float, bn[img, i]))) in cand for i in range(n))
    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    odec = O.Decoder(0.01, 0.6, 100, 300, True, True)
    for img in (0, B - 1):
        ol = [l[img:img + 1].float().cpu().numpy() for l in loc]
        oc = [x[img:img + 1].float().cpu().numpy() for x in conf]
        wm = odec.decode_levels(ol, oc, oanch)
        np.testing.assert_array_equal(mc[img:img + 1], wm[2])
        np.testing.assert_allclose(mb[img:img + 1], wm[1], atol=BOX_ATOL, rtol=0)
        np.testing.assert_allclose(ms[img:img + 1], wm[0], atol=1e-4, rtol=1e-4)
        w = odec(ol, oc, oanch)
        np.testing.assert_array_equal(cn[img:img + 1], w[2])
        np.testing.assert_allclose(bn[img:img + 1], w[1], atol=BOX_ATOL, rtol=0)
        np.testing.assert_allclose(sn[img:img + 1], w[0], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("kind", ["constant", "ramp_up", "quantized", "sparse", "logit"])
@pytest.mark.parametrize("tpu", ["1", "4", "0"])
def test_decoder_every_path_agrees_with_the_oracle(kind, tpu, monkeypatch):
    """Decoder.__call__ through each way the stage can run -- levelsel_kernel + nmswalk_kernel (default) vs level_kernel +
    nms_kernel (SSDK_DECODE_FUSED=0), seeded barrier-free scan (default) vs the TopK stream only (SSDK_SCAN_FAST=0) -- on inputs
    full of ties, with many scan units per level (SSDK_TILES_PER_UNIT=1/4: the tail merges up to 32 sorted lists):
    every path equals the oracle (final and per-level outputs), hence each other, bit for bit in classes / order."""
    import torch
    from ssds.modeling.layers.box import decode_nms

    A, C = 3, 20
    maps, strides = [(24, 28), (12, 14), (6, 7), (3, 4)], [8, 16, 32, 64]
    rs = np.random.RandomState(11)
    conf, loc = [], []
    for li, (h, w) in enumerate(maps):
        if kind == "logit":
            c = cases.sigmoid(rs.standard_normal((2, A * C, h, w)).astype(F32) * F32(1.5) - F32(4.6))
            l = (rs.standard_normal((2, A * 4, h, w)) * 0.5).astype(F32)
        else:
            c, l = _tie_case(kind, 2, A, C, h, w, 20 + li)
        conf.append(torch.from_numpy(c).to(torch.bfloat16))
        loc.append(torch.from_numpy(l).to(torch.bfloat16))
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0]))) for s in strides)
    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    args = (0.05, 100, True, 0.5, 60, True)
    odec = O.Decoder(args[0], args[3], args[4], args[1], args[2], args[5])
    ol, oc = [t.float().numpy() for t in loc], [t.float().numpy() for t in conf]
    wmid = odec.decode_levels(ol, oc, oanch)
    want = odec(ol, oc, oanch)
    monkeypatch.setenv("SSDK_TILES_PER_UNIT", tpu)
    dl, dc = [t.cuda() for t in loc], [t.cuda() for t in conf]
    for env in ({}, {"SSDK_DECODE_FUSED": "0"}, {"SSDK_SCAN_FAST": "0"}, {"SSDK_DECODE_FUSED": "0", "SSDK_SCAN_FAST": "0"}):
        for k in ("SSDK_DECODE_FUSED", "SSDK_SCAN_FAST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        (s, b, c), mid = decode_nms(dl, dc, anchors, *args, return_mid=True)
        what = "%s tpu=%s %s" % (kind, tpu, env)
        np.testing.assert_array_equal(mid[2].cpu().numpy(), wmid[2], err_msg=what + " mid classes")
        np.testing.assert_allclose(mid[1].cpu().numpy(), wmid[1], atol=BOX_ATOL, rtol=0, err_msg=what + " mid boxes")
        np.testing.assert_allclose(mid[0].cpu().numpy(), wmid[0], atol=1e-4, rtol=1e-4, equal_nan=True,
                                   err_msg=what + " mid scores")
        np.testing.assert_array_equal(c.cpu().numpy(), want[2], err_msg=what + " classes")
        np.testing.assert_allclose(b.cpu().numpy(), want[1], atol=BOX_ATOL, rtol=0, err_msg=what + " boxes")
        np.testing.assert_allclose(s.cpu().numpy(), want[0], atol=1e-4, rtol=1e-4, err_msg=what + " scores")


def test_two_threads_two_decoders_share_nothing():
    """SURVEY 8b "re-entrant": two host threads, each with its own Decoder (= its own ssdk_ctx: events, profiling ring,
    tail-stream state) on its own stream, decode different batches concurrently; both equal their serial results."""
    import threading

    import torch
    from ssds.modeling.layers.decoder import Decoder

    A, C = 6, 20
    maps, strides = [16, 8, 4], [16, 32, 64]
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828]))) for s in strides)
    data = []
    for seed in (1, 2):
        g = torch.Generator(device="cuda").manual_seed(seed)
        conf = [torch.sigmoid(torch.randn(8, A * C, m, m, device="cuda", generator=g) * 1.5 - 3.0).to(torch.bfloat16) for m in maps]
        loc = [(torch.randn(8, A * 4, m, m, device="cuda", generator=g) * 0.5).to(torch.bfloat16) for m in maps]
        data.append((loc, conf))
    want = [tuple(t.clone() for t in Decoder(0.05, 0.5, 50, 200, True, True)(l, c, anchors)) for l, c in data]
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            dec = Decoder(0.05, 0.5, 50, 200, True, True)
            dec.set_profiling(True)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(200):
                    got = dec(data[i][0], data[i][1], anchors)
            st.synchronize()
            for a, b in zip(got, want[i]):
                assert torch.equal(a, b)
            assert dec.timings_ms(0)[0] > 0
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def _decoder_vs_oracle(loc, conf, anchors, args, what, check_mid=True):
    """decode_nms on device tensors against the numpy oracle on the same (upcast) values: per-level and final outputs."""
    import torch
    from ssds.modeling.layers.box import decode_nms

    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    odec = O.Decoder(args[0], args[3], args[4], args[1], args[2], args[5])
    ol, oc = [t.float().cpu().numpy() for t in loc], [t.float().cpu().numpy() for t in conf]
    (s, b, c), mid = decode_nms(loc, conf, anchors, *args, return_mid=True)
    if check_mid:
        wm = odec.decode_levels(ol, oc, oanch)
        np.testing.assert_array_equal(mid[2].cpu().numpy(), wm[2], err_msg=what + " mid classes")
        np.testing.assert_allclose(mid[1].cpu().numpy(), wm[1], atol=BOX_ATOL, rtol=0, err_msg=what + " mid boxes")
        np.testing.assert_allclose(mid[0].cpu().numpy(), wm[0], atol=1e-4, rtol=1e-4, equal_nan=True,
                                   err_msg=what + " mid scores")
    want = odec(ol, oc, oanch)
    np.testing.assert_array_equal(c.cpu().numpy(), want[2], err_msg=what + " classes")
    np.testing.assert_allclose(b.cpu().numpy(), want[1], atol=BOX_ATOL, rtol=0, err_msg=what + " boxes")
    np.testing.assert_allclose(s.cpu().numpy(), want[0], atol=1e-4, rtol=1e-4, err_msg=what + " scores")


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("kind", ["nan_inf_negative", "misleading_sample", "all_equal", "two_values", "dense_above", "near_ties",
                                  "few_values", "stairs"])
@pytest.mark.parametrize("tpu", ["0", "3", "40"])
def test_scan16_special_values_ties_and_fallback(kind, dtype, tpu, monkeypatch):
    """The 16-bit scan (ssdk_scan16.hip) compares bit patterns as signed 16-bit integers and trusts a sample of the unit
    for its cut; everything that could break those two assumptions, against the oracle (box.py:435-446 semantics):
      nan_inf_negative   NaNs (never pass `>= thr`, box.py:440), +inf (passes), negative scores and -0.0;
      misleading_sample  the sample tiles hold nothing, every other tile is dense: the wave buffers overflow and the unit
                         falls back to the exact TopK stream;
      all_equal          one value everywhere (f16: in the middle of a histogram bin -> the in-bin refinement);
      two_values         60 % of the scores at one value, the rest at a second, larger one in the LAST part of the map
                         (ties at the cut in some units, nothing but ties above the cut in others);
      dense_above        every score above the threshold and distinct per position (more candidates than any buffer);
      near_ties          sigmoid(N(-4.595, 0.02)): every score on a handful of neighbouring 16-bit values -- the value at the
                         cut occurs thousands of times per unit without filling the sample's maxima (round 5: the unit
                         counts its sample registers, raises the cut or settles the cut value as a tie);
      few_values         uniform over [0.01, 0.02): ~128 bf16 values of equal frequency (several raises of the cut);
      stairs             five values, the larger the rarer (1, 4, 16, 64 per mille): the cut lands on each of them for
                         some K."""
    import torch

    tdt = getattr(torch, dtype)
    monkeypatch.setenv("SSDK_TILES_PER_UNIT", tpu) if tpu != "0" else monkeypatch.delenv("SSDK_TILES_PER_UNIT", raising=False)
    rs = np.random.RandomState(5)
    A, C, B = 3, 40, 3
    maps, strides = [(40, 48), (20, 24), (5, 6)], [8, 16, 64]
    conf, loc = [], []
    for h, w in maps:
        n = A * C * h * w
        base = cases.sigmoid(rs.standard_normal((B, n)).astype(F32) * F32(1.5) - F32(4.6))
        if kind == "nan_inf_negative":
            c = base.copy()
            for bi in range(B):
                idx = rs.choice(n, size=max(8, n // 50), replace=False)
                vals = np.array([np.nan, np.inf, -np.inf, -0.0, -0.5, -3.0, 2.5, np.nan], F32)
                c[bi, idx] = vals[np.arange(idx.size) % vals.size]
            c[0, :4] = np.nan  # inside the first sample vector of the first unit
        elif kind == "misleading_sample":
            c = np.full((B, n), 0.001, F32)
            tile = 256 * 8
            t = np.arange(n) // tile
            ntiles = (n + tile - 1) // tile
            stride = max(ntiles // 8, 1)
            dense = (t % stride != 0) | (t >= 8 * stride)  # (exact for a single unit; harmless otherwise)
            c[:, dense] = (0.02 + 0.9 * rs.random_sample((B, int(dense.sum())))).astype(F32)
        elif kind == "near_ties":
            c = cases.sigmoid(rs.standard_normal((B, n)).astype(F32) * F32(0.02) - F32(4.595))
        elif kind == "few_values":
            c = (0.01 + 0.01 * rs.random_sample((B, n))).astype(F32)
        elif kind == "stairs":
            u = rs.random_sample((B, n))
            c = np.full((B, n), 0.125, F32)
            for v, f in ((0.25, 0.064), (0.375, 0.016), (0.5, 0.004), (0.75, 0.001)):
                c[u < f] = v
        elif kind == "all_equal":
            c = np.full((B, n), 0.3337, F32)
        elif kind == "two_values":
            c = np.full((B, n), 0.25, F32)
            c[:, int(n * 0.6):] = 0.75
        else:
            c = (0.02 + 0.97 * rs.random_sample((B, n))).astype(F32)
        conf.append(torch.from_numpy(c.reshape(B, A * C, h, w)).to(tdt).cuda())
        loc.append(torch.from_numpy((rs.standard_normal((B, A * 4, h, w)) * 0.5).astype(F32)).to(tdt).cuda())
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0]))) for s in strides)
    for args in ((0.01, 300, True, 0.6, 100, True), (0.05, 37, False, 0.5, 20, False), (0.2, 512, True, 0.6, 100, True)):
        _decoder_vs_oracle(loc, conf, anchors, args, "%s %s tpu=%s K=%d" % (kind, dtype, tpu, args[1]))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_scan16_failed_prediction_falls_back_to_the_proven_cut(dtype, monkeypatch):
    """The near-tie rule of ssdk_scan16.hip may raise its cut on a PREDICTION (the sample tiles hold 240 scores above the cut
    value, x 5 tiles per sample tile = 1 200 expected in the unit, >= 2 K): here the larger value lives in the sample tiles
    ONLY (every fifth tile of an 80-tile unit), so fewer than K keys come back and the unit must rerun exactly from the last
    proven cut.  Layout: 0.125 everywhere (above the threshold, below the cut), ~16 scores of 0.25 in every tile (the cut
    value: frequent in the unit, 230 of the 1 024 sample maxima), 15 scores of 0.5 in the sample tiles."""
    import torch

    tdt = getattr(torch, dtype)
    monkeypatch.setenv("SSDK_TILES_PER_UNIT", "80")
    rs = np.random.RandomState(11)
    A, C, B = 3, 40, 2
    maps, strides = [(40, 48), (20, 24)], [8, 16]
    conf, loc = [], []
    tile = 256 * 8
    for h, w in maps:
        n = A * C * h * w
        c = np.full((B, n), 0.125, F32)
        for bi in range(B):
            for t in range((n + tile - 1) // tile):
                lo, hi = t * tile, min((t + 1) * tile, n)
                pos = rs.choice(hi - lo, size=min(31, hi - lo), replace=False) + lo
                c[bi, pos[:16]] = 0.25
                if t % 5 == 0 and t < 80:
                    c[bi, pos[16:31]] = 0.5
        conf.append(torch.from_numpy(c.reshape(B, A * C, h, w)).to(tdt).cuda())
        loc.append(torch.from_numpy((rs.standard_normal((B, A * 4, h, w)) * 0.5).astype(F32)).to(tdt).cuda())
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0]))) for s in strides)
    for args in ((0.01, 300, True, 0.6, 100, True), (0.01, 260, False, 0.5, 50, True)):
        _decoder_vs_oracle(loc, conf, anchors, args, "failed prediction %s K=%d" % (dtype, args[1]))


@pytest.mark.parametrize("tpu", ["1", "0"])
def test_tail_generic_select_when_one_bin_holds_more_than_k_keys(tpu, monkeypatch):
    """fp32 heads whose scores sit in a band narrower than one bin of the tail's per-level histogram (and differ from one
    another): the boundary bin holds every candidate of every unit, far more than K -- the tail's adaptive radix select
    over the units' lists decides (ssdk_tail.hip, `generic`), with many units per level (SSDK_TILES_PER_UNIT=1) and with
    the planner's own unit size."""
    import torch

    monkeypatch.setenv("SSDK_TILES_PER_UNIT", tpu) if tpu != "0" else monkeypatch.delenv("SSDK_TILES_PER_UNIT", raising=False)
    rs = np.random.RandomState(3)
    A, C, B = 3, 20, 2
    maps, strides = [(24, 28), (12, 14), (3, 4)], [8, 16, 64]
    conf, loc = [], []
    for h, w in maps:
        n = A * C * h * w
        c = (F32(0.5) + rs.permutation(n).astype(F32) * F32(2.0 ** -22))[None].repeat(B, 0)  # distinct, all inside one bin
        c[1] = c[1][::-1]
        conf.append(torch.from_numpy(np.ascontiguousarray(c).reshape(B, A * C, h, w)).cuda())
        loc.append(torch.from_numpy((rs.standard_normal((B, A * 4, h, w)) * 0.5).astype(F32)).cuda())
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0]))) for s in strides)
    for args in ((0.05, 100, True, 0.5, 60, True), (0.01, 300, False, 0.6, 100, True)):
        _decoder_vs_oracle(loc, conf, anchors, args, "narrow band tpu=%s K=%d" % (tpu, args[1]))


def test_sixteen_bit_heads_with_a_non_positive_threshold_take_the_generic_scan():
    """thr <= 0 lets negative scores through, which the 16-bit integer compare cannot order: scan_kernel runs instead."""
    import torch

    rs = np.random.RandomState(8)
    A, C, B, h, w = 3, 10, 2, 16, 20
    c = (rs.standard_normal((B, A * C, h, w)) * 0.3).astype(F32)
    conf = [torch.from_numpy(c).to(torch.bfloat16).cuda()]
    loc = [torch.from_numpy((rs.standard_normal((B, A * 4, h, w)) * 0.5).astype(F32)).to(torch.bfloat16).cuda()]
    anchors = OrderedDict([(8, torch.from_numpy(O.generate_anchors(8, [1, 2, 0.5], [2.0])))])
    _decoder_vs_oracle(loc, conf, anchors, (-0.1, 100, False, 0.5, 50, True), "thr < 0")
    _decoder_vs_oracle(loc, conf, anchors, (0.0, 100, False, 0.5, 50, True), "thr = 0")
