"""a13-a18 on the device against the REFERENCE's own classes: the recorded plans (backbone + neck + towers + heads
on the HIP kernels) of this repo's SSD / SSDFPN / SSDBiFPN, loaded with the seeded weights of
tests/golden/net_*.npz, against the fp32 outputs the reference's modules produced on the same weights and inputs
(ssd.py:42-74, fpn.py:58-101, bifpn.py:30-63,104-142); and SSDDetector.__call__ end to end (ssds.py:41-68).

Tolerances are per element, in units of the RMS of the reference tensor of that level (not of its maximum): a
16-bit network through 20-50 layers carries ~sqrt(depth) * 2^-9 (bf16) / 2^-12 (fp16) of relative rounding noise; a
wiring or folding error is O(1)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import cases
import nethelp
from oracle import box_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (median, 99.9th percentile, max) of |got - want| / rms(want), per level
BARS = {"bfloat16": (0.012, 0.08, 0.2), "float16": (0.003, 0.02, 0.05)}


def _nerr(got, want):
    import torch

    g = got.float().cpu()
    assert g.shape == want.shape, (g.shape, want.shape)
    rms = float(want.pow(2).mean().sqrt())
    return ((g - want).abs() / max(rms, 1e-6)).flatten()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("name", list(cases.NET_CASES))
def test_plan_matches_reference_module(name, dtype):
    import torch
    from ssds.modeling.layers import fused_conv as FC

    tdt = getattr(torch, dtype)
    model, x, fx = nethelp.build(name)
    wl, wc = nethelp.want(fx)
    model = model.cuda().to(tdt)
    before, plans = FC.STATS["native_layers"], FC.STATS["plan_runs"]
    with torch.no_grad():
        loc, conf = model(x.cuda().to(tdt))
        loc2, conf2 = model(x.cuda().to(tdt))
    assert FC.STATS["native_layers"] > before, "nothing ran on the HIP kernels"
    if name != "ssd_stub":  # (a stub backbone under SSD leaves only extras + heads: per-layer launches, no plan)
        assert FC.STATS["plan_runs"] >= plans + 2, "the forward did not run as a recorded plan"
    med_bar, p999_bar, max_bar = BARS[dtype]
    report = []
    for i, (l, a, c, b) in enumerate(zip(loc, wl, conf, wc)):
        assert l.is_contiguous() and c.is_contiguous() and l.dtype == tdt
        assert torch.equal(l, loc2[i]) and torch.equal(c, conf2[i]), "replay is not deterministic"
        for tag, e in (("loc", _nerr(l, a)), ("conf", _nerr(c, b))):
            k = max(int(e.numel() * 0.999) - 1, 0)
            med, p999, mx = float(e.median()), float(e.kthvalue(k + 1).values), float(e.max())
            report.append("%s%d med %.4f p99.9 %.4f max %.4f" % (tag, i, med, p999, mx))
            assert med <= med_bar and p999 <= p999_bar and mx <= max_bar, (name, dtype, report)
    print(name, dtype, "; ".join(report))


def _seeded_detector(cfg_name, dtype):
    """SSDDetector on the cfg with seeded, BatchNorm-calibrated weights (the reference's init puts every score at the
    threshold, which decides nothing)."""
    import torch
    from ssds.ssds import SSDDetector

    det = SSDDetector(os.path.join(ROOT, "experiments", "cfgs", cfg_name), dtype=dtype)
    model = det.model.float().cpu()
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = cases.seeded_state(spec, 123)
    for k in state:  # an untrained-looking score distribution: logits ~ N(-4, ~1), a few confident peaks
        if k.startswith("conf.") and k.endswith("weight"):
            state[k] = state[k] * np.float32(0.6)
        if k.startswith("conf.") and k.endswith("bias"):
            state[k] = (state[k] * 3 - 4.0).astype(np.float32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = None
    model.train()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for _ in range(2):
            model(torch.rand((2, 3) + det.image_size, generator=g))
    det.model = model.eval()
    ref = {k: v.clone() for k, v in model.state_dict().items()}
    det.model.to(det.device, dtype)
    return det, ref


@pytest.mark.parametrize("layout", ["nhwc_u8", "nchw_f32", "hwc_u8"])
def test_ssd_detector_call_end_to_end(layout):
    """SSDDetector.__call__ (ssds.py:41-68) on raw images: (1) the call equals preprocess -> model -> Decoder ->
    astype(int) composed by hand from its parts, with the numpy oracle decoding the device's own head outputs
    (bit-exact classes / keep order, boxes within 1e-3 before the int cast); (2) against the fp32 CPU pipeline
    (oracle preprocess -> this module in fp32 -> oracle Decoder) every confident detection is found again."""
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.ssds import preprocess

    det, ref_state = _seeded_detector("ssd_mobilenetv2_300.yml", torch.bfloat16)
    rs = np.random.RandomState(11)
    n = 1 if layout == "hwc_u8" else 3
    img = rs.randint(0, 256, (n, 300, 300, 3)).astype(np.uint8)
    if layout == "nchw_f32":
        img = np.ascontiguousarray(img.transpose(0, 3, 1, 2)).astype(np.float32)
    arg = img[0] if layout == "hwc_u8" else img
    scores, boxes, classes = det(arg)
    if layout == "hwc_u8":
        assert scores.shape == (100,) and boxes.shape == (100, 4) and classes.shape == (100,)
        scores, boxes, classes = scores[None], boxes[None], classes[None]
    assert scores.dtype == np.float32 and boxes.dtype.kind == "i" and classes.dtype.kind == "i"
    assert scores.shape == (n, 100) and boxes.shape == (n, 100, 4)

    # (1) glue: the same chain by hand; the oracle decodes the device's head outputs
    want_x = torch.from_numpy(O.preprocess(img, det.mean, det.std)).to(torch.bfloat16)
    x = preprocess(torch.from_numpy(img).cuda(), det.mean, det.std, torch.bfloat16)
    assert torch.equal(x.cpu(), want_x)
    with torch.no_grad():
        loc, conf = det.model(x)
    oanch = OrderedDict((k, v.numpy()) for k, v in det.anchors.items())
    d = det.decoder
    odec = O.Decoder(d.conf_threshold, d.nms_threshold, d.top_n, d.top_n_per_level, d.rescore, d.use_diou)
    ws, wb, wc = odec([t.float().cpu().numpy() for t in loc], [t.float().cpu().numpy() for t in conf], oanch)
    np.testing.assert_array_equal(classes, wc.astype(int))
    np.testing.assert_allclose(scores, ws, atol=1e-4, rtol=1e-4)
    assert np.abs(boxes - wb).max() <= 1.0 + 1e-3  # int() of a value within 1e-3
    assert (scores[:, 0] > 0.05).all(), "the seeded model must produce confident detections"

    # (2) the whole pipeline in fp32 on the CPU (oracle preprocess -> this module in fp32 -> oracle Decoder).  With
    # untrained weights the top-100 of 240 000 near-iid scores is reshuffled by any rounding noise, so identities of
    # detections cannot be compared across precisions; what can: the head tensors (per element, RMS units) and the
    # order statistics of the detection scores.
    config.reset_cfg()
    cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_300.yml"))
    cpu = model_builder.create_model(cfg.MODEL)
    cpu.load_state_dict(ref_state)
    cpu.eval()
    with torch.no_grad():
        cl, cc = cpu(torch.from_numpy(O.preprocess(img, det.mean, det.std)))
    med_bar, p999_bar, max_bar = BARS["bfloat16"]
    for l, a, c, b in zip(loc, cl, conf, cc):
        for e in (_nerr(l, a), _nerr(c, b)):
            k = max(int(e.numel() * 0.999) - 1, 0)
            assert float(e.median()) <= med_bar and float(e.kthvalue(k + 1).values) <= p999_bar and float(e.max()) <= max_bar
    fs, fb, fc = odec([t.numpy() for t in cl], [t.numpy() for t in cc], oanch)
    assert np.abs(np.sort(scores, 1) - np.sort(fs, 1)).max() < 0.06, "score order statistics drifted"
