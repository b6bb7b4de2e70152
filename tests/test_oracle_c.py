"""The C restatement of the box math behind the C-ABI symbols (oracle/ssdk_cpu.c -> oracle/_build/libssdk_cpu.so,
SURVEY.md 8b: "same symbols compiled for CPU for tests"), pinned to the REFERENCE by the same fixtures that pin the numpy
oracle (tests/golden/*.npz, written by the reference's own functions) and cross-checked against the numpy oracle on
seeded inputs, 16-bit heads included.  Test infrastructure: the product never loads this library."""
import ctypes
import os
import re
import subprocess
from collections import OrderedDict

import numpy as np
import pytest

import cases
from oracle import box_oracle as O
from ssds._native import Level  # the ctypes mirror of `ssdk_level` (struct layout only: libssdk.so is not called here)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
F32 = np.float32
DT = {"f32": 0, "bf16": 1, "f16": 2}
fp, vp, i32, f32, sz = ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "oracle", "_build", "libssdk_cpu.so")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    l = ctypes.CDLL(so)
    l.ssdk_last_error.restype = ctypes.c_char_p
    l.ssdk_generate_anchors.argtypes = [i32, fp, i32, fp, i32, fp]
    l.ssdk_decode.argtypes = [ctypes.POINTER(Level), i32, i32, f32, i32, i32, fp, fp, fp, vp, sz, vp]
    l.ssdk_nms.argtypes = [fp, fp, fp, i32, i32, f32, i32, i32, fp, fp, fp, vp, sz, vp]
    l.ssdk_decode_nms.argtypes = [ctypes.POINTER(Level), i32, i32, i32, f32, i32, i32, f32, i32, i32] + [fp] * 6 + [vp, sz, vp]
    l.ssdk_match_targets.argtypes = [fp, i32, i32, fp, i32, i32, i32, i32, i32, f32, f32, f32, fp, fp, fp, vp]
    l.ssdk_match_targets_by_scale.argtypes = [fp, i32, i32, fp, i32, i32, i32, i32, i32, f32, f32, i32, fp, fp, fp, vp]
    for n in ("ssdk_decode_workspace_bytes", "ssdk_nms_workspace_bytes", "ssdk_decode_nms_workspace_bytes"):
        getattr(l, n).restype = sz
    return l


def ptr(a):
    return a.ctypes.data_as(fp)


def ok(lib, rc):
    assert rc == 0, lib.ssdk_last_error().decode()


def c_anchors(lib, stride, ratios, scales):
    r, s = np.asarray(ratios, F32), np.asarray(scales, F32)
    out = np.empty((len(r) * len(s), 4), F32)
    ok(lib, lib.ssdk_generate_anchors(stride, ptr(r), len(r), ptr(s), len(s), ptr(out)))
    return out


def make_level(cls, box, anchors, stride, keep):
    """cls / box: [B, A*C, H, W] / [B, A*4, H, W] arrays of fp32 or uint16 (bf16 / f16 bits)."""
    lv = Level()
    cls, box = np.ascontiguousarray(cls), np.ascontiguousarray(box)
    keep += [cls, box]  # the struct holds raw addresses
    A = anchors.shape[0]
    lv.cls, lv.box = cls.ctypes.data, box.ctypes.data
    lv.A, lv.C, lv.H, lv.W, lv.stride = A, cls.shape[1] // A, cls.shape[2], cls.shape[3], int(stride)
    for i, v in enumerate(np.asarray(anchors, F32).reshape(-1)):
        lv.anchors[i] = v
    return lv


def c_decode(lib, cls, box, stride, thr, top_n, anchors, rescore, dtype="f32"):
    keep = []
    lv = make_level(cls, box, anchors, stride, keep)
    B = cls.shape[0]
    s, b, c = np.full((B, top_n), 7, F32), np.full((B, top_n, 4), 7, F32), np.full((B, top_n), 7, F32)  # must be overwritten
    ok(lib, lib.ssdk_decode(ctypes.byref(lv), B, DT[dtype], thr, top_n, int(rescore), ptr(s), ptr(b), ptr(c), None, 0, None))
    return s, b, c


def c_nms(lib, scores, boxes, classes, thr, ndet, diou):
    scores, boxes, classes = (np.ascontiguousarray(x, F32) for x in (scores, boxes, classes))
    B, N = scores.shape
    s, b, c = np.full((B, ndet), 7, F32), np.full((B, ndet, 4), 7, F32), np.full((B, ndet), 7, F32)
    ok(lib, lib.ssdk_nms(ptr(scores), ptr(boxes), ptr(classes), B, N, thr, ndet, int(diou), ptr(s), ptr(b), ptr(c), None, 0, None))
    return s, b, c


def c_decoder(lib, loc, conf, anchors, thr, nms, top_n, per_level, rescore, diou, dtype="f32", mid=True):
    keep = []
    L, B = len(loc), loc[0].shape[0]
    arr = (Level * L)(*[make_level(c, l, a, s, keep) for l, c, (s, a) in zip(loc, conf, anchors.items())])
    N = L * per_level
    out = [np.full((B, top_n), 7, F32), np.full((B, top_n, 4), 7, F32), np.full((B, top_n), 7, F32)]
    mids = [np.full((B, N), 7, F32), np.full((B, N, 4), 7, F32), np.full((B, N), 7, F32)] if mid else [None] * 3
    ok(lib, lib.ssdk_decode_nms(arr, L, B, DT[dtype], thr, per_level, int(rescore), nms, top_n, int(diou), *[ptr(o) for o in out],
                                *[ptr(m) if m is not None else None for m in mids], None, 0, None))
    return out, mids


def c_match(lib, targets, anchors, C, stride, size, match, radius=0.0, by_scale=False):
    targets, anchors = np.ascontiguousarray(targets, F32), np.ascontiguousarray(anchors, F32)
    B, Gt = targets.shape[:2]
    A, (H, W) = anchors.shape[0], size
    ct, bt, dp = np.full((B, A, C, H, W), 7, F32), np.full((B, A, 4, H, W), 7, F32), np.full((B, A, 1, H, W), 7, F32)
    if by_scale:
        rc = lib.ssdk_match_targets_by_scale(ptr(targets), B, Gt, ptr(anchors), A, C, H, W, stride, match[0], match[1], int(radius > 0),
                                             ptr(ct), ptr(bt), ptr(dp), None)
    else:
        rc = lib.ssdk_match_targets(ptr(targets), B, Gt, ptr(anchors), A, C, H, W, stride, match[0], match[1], radius, ptr(ct), ptr(bt),
                                    ptr(dp), None)
    ok(lib, rc)
    return ct, bt, dp


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


# ---- the boundary: same names, same prototypes ------------------------------------------------------------------------
def test_exports_are_the_header_s_symbols(lib):
    """Every function the CPU library defines is declared in include/ssdk.h (it is compiled against that header, so the
    prototypes are the header's), and the box-math entry points of SURVEY 8b are all there."""
    so = os.path.join(ROOT, "oracle", "_build", "libssdk_cpu.so")
    nm = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    header = open(os.path.join(ROOT, "include", "ssdk.h")).read()
    declared = set(re.findall(r"\b(ssdk_[a-z0-9_]+)\s*\(", header))
    assert defined <= declared, defined - declared
    assert {"ssdk_generate_anchors", "ssdk_decode", "ssdk_nms", "ssdk_decode_nms", "ssdk_match_targets", "ssdk_match_targets_by_scale",
            "ssdk_decode_workspace_bytes", "ssdk_nms_workspace_bytes", "ssdk_decode_nms_workspace_bytes", "ssdk_version",
            "ssdk_last_error"} <= defined
    assert lib.ssdk_version() == int(re.search(r"#define SSDK_VERSION (\d+)", header).group(1))


def test_bad_arguments_are_status_codes(lib):
    z = np.zeros(8, F32)
    assert lib.ssdk_generate_anchors(0, ptr(z), 1, ptr(z), 1, ptr(z)) == -1 and b"generate_anchors" in lib.ssdk_last_error()
    assert lib.ssdk_nms(ptr(z), ptr(z), ptr(z), 1, 0, 0.5, 1, 1, ptr(z), ptr(z), ptr(z), None, 0, None) == -1


# ---- against the reference's fixtures ----------------------------------------------------------------------------------
def test_anchors_bit_exact_vs_reference(lib):
    g = load("anchors")
    for k in g.files:
        if k.endswith("_spec"):
            continue
        spec = g[k + "_spec"]
        s, nr, ns = int(spec[0]), int(spec[1]), int(spec[2])
        np.testing.assert_array_equal(c_anchors(lib, s, spec[3:3 + nr], spec[3 + nr:3 + nr + ns]), g[k], err_msg=k)


@pytest.mark.parametrize("name", list(cases.DECODE_CASES))
def test_decode_vs_reference(lib, name):
    g = load("decode")
    d = cases.decode_inputs(name)
    anchors = cases.anchors_for(d["A"], d["stride"], O.generate_anchors)
    s, b, c = c_decode(lib, d["cls"], d["box"], d["stride"], d["thr"], d["top_n"], anchors, d["rescore"])
    np.testing.assert_array_equal(c, g[name + "_classes"])  # indices / classes bit exact
    np.testing.assert_allclose(b, g[name + "_boxes"], atol=1e-3, rtol=0)  # north_star: 1e-3 on decoded coordinates
    np.testing.assert_allclose(s, g[name + "_scores"], atol=1e-6, rtol=1e-5, equal_nan=True)


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_nms_vs_reference(lib, name):
    g = load("nms")
    d = cases.nms_inputs(name)
    s, b, c = c_nms(lib, d["scores"], d["boxes"], d["classes"], d["thr"], d["ndet"], d["diou"])
    np.testing.assert_array_equal(s, g[name + "_scores"])  # keep set + order bit exact
    np.testing.assert_array_equal(b, g[name + "_boxes"])
    np.testing.assert_array_equal(c, g[name + "_classes"])


@pytest.mark.parametrize("name", list(cases.DECODER_CASES))
def test_decoder_vs_reference(lib, name):
    g = load("decoder")
    d = cases.decoder_inputs(name, O.generate_anchors)
    (s, b, c), (ms, mb, mc) = c_decoder(lib, d["loc"], d["conf"], d["anchors"], d["thr"], d["nms"], d["top_n"], d["per_level"],
                                        d["rescore"], d["diou"])
    np.testing.assert_array_equal(mc, g[name + "_mid_classes"])
    np.testing.assert_allclose(mb, g[name + "_mid_boxes"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(ms, g[name + "_mid_scores"], atol=1e-6, rtol=1e-5)
    np.testing.assert_array_equal(c, g[name + "_classes"])
    np.testing.assert_allclose(b, g[name + "_boxes"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(s, g[name + "_scores"], atol=1e-6, rtol=1e-5)
    # the final stage on the reference's own intermediate: keep set bit exact
    s, b, c = c_nms(lib, g[name + "_mid_scores"], g[name + "_mid_boxes"], g[name + "_mid_classes"], d["nms"], d["top_n"], d["diou"])
    np.testing.assert_array_equal(s, g[name + "_scores"])
    np.testing.assert_array_equal(b, g[name + "_boxes"])
    np.testing.assert_array_equal(c, g[name + "_classes"])


@pytest.mark.parametrize("name", list(cases.MATCH_CASES))
def test_match_targets_vs_reference(lib, name):
    g = load("match")
    d = cases.match_inputs(name, O.generate_anchors)
    ct, bt, dp = c_match(lib, d["targets"], d["anchors"], d["C"], d["stride"], d["size"], tuple(map(float, d["match"])), float(d["radius"]))
    np.testing.assert_array_equal(dp, g[name + "_depth"])  # matching decisions bit exact
    np.testing.assert_array_equal(ct.astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt, g[name + "_box"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(cases.SCALE_MATCH_CASES))
def test_match_targets_by_scale_vs_reference(lib, name):
    g = load("match_scale")
    d = cases.match_inputs(name, O.generate_anchors)
    ct, bt, dp = c_match(lib, d["targets"], d["anchors"], d["C"], d["stride"], d["size"], tuple(map(float, d["match"])), float(d["radius"]),
                         by_scale=True)
    np.testing.assert_array_equal(dp, g[name + "_depth"])
    np.testing.assert_array_equal(ct.astype(np.uint8), g[name + "_cls"])
    np.testing.assert_allclose(bt, g[name + "_box"], rtol=1e-5, atol=1e-5)


# ---- against the numpy oracle on seeded inputs -----------------------------------------------------------------------------
def _heads(seed, B, A, C, sizes, dtype):
    rs = np.random.RandomState(seed)
    loc, conf, bits_l, bits_c = [], [], [], []
    for h, w in sizes:
        c = cases.sigmoid(rs.normal(-3.0, 1.5, (B, A * C, h, w)).astype(F32))
        l = rs.normal(0, 0.5, (B, A * 4, h, w)).astype(F32)
        if dtype != "f32":
            rnd = cases.bf16_round if dtype == "bf16" else cases.f16_round
            (cb, c), (lb, l) = rnd(c), rnd(l)
            bits_c.append(cb.reshape(c.shape))
            bits_l.append(lb.reshape(l.shape))
        c = cases.make_unique_above(c, 0.0) if dtype == "f32" else c
        conf.append(c)
        loc.append(l)
    return loc, conf, (bits_l or loc), (bits_c or conf)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_decoder_matches_the_numpy_oracle(lib, dtype):
    """Whole Decoder (decode every level -> concat -> nms) on random heads; 16-bit heads tie a lot, so the stable
    (score descending, index ascending) order is exercised."""
    B, A, C = 3, 6, 11
    sizes, strides = [(10, 12), (5, 6), (3, 3), (1, 1)], [8, 16, 32, 64]
    loc, conf, raw_l, raw_c = _heads(11, B, A, C, sizes, dtype)
    anchors = OrderedDict((s, O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
    for rescore, diou in ((True, True), (False, False)):
        dec = O.Decoder(0.05, 0.5, 40, 50, rescore, diou)
        ms, mb, mc = dec.decode_levels(loc, conf, anchors)
        want = dec(loc, conf, anchors)
        (s, b, c), (gs, gb, gc) = c_decoder(lib, raw_l, raw_c, anchors, 0.05, 0.5, 40, 50, rescore, diou, dtype)
        np.testing.assert_array_equal(gc, mc)
        np.testing.assert_allclose(gb, mb, atol=1e-3, rtol=0)
        np.testing.assert_allclose(gs, ms, atol=1e-6, rtol=1e-5, equal_nan=True)
        np.testing.assert_array_equal(c, want[2])
        np.testing.assert_allclose(b, want[1], atol=1e-3, rtol=0)
        np.testing.assert_allclose(s, want[0], atol=1e-6, rtol=1e-5)
        (s2, b2, c2), _ = c_decoder(lib, raw_l, raw_c, anchors, 0.05, 0.5, 40, 50, rescore, diou, dtype, mid=False)
        for x, y in zip((s, b, c), (s2, b2, c2)):  # the optional intermediate does not change the result
            np.testing.assert_array_equal(x, y)


def test_nms_matches_the_numpy_oracle_with_ties_and_dead_slots(lib):
    rs = np.random.RandomState(5)
    B, N = 4, 300
    xy = rs.uniform(0, 200, (B, N, 2)).astype(F32)
    boxes = np.concatenate([xy, xy + rs.uniform(5, 80, (B, N, 2)).astype(F32)], 2)
    scores = np.round(rs.uniform(-0.2, 1, (B, N)), 2).astype(F32)  # ties, zeros and negatives
    scores[0, :10] = np.nan
    classes = rs.randint(0, 3, (B, N)).astype(F32)
    for diou in (True, False):
        for ndet in (1, 30, 300):
            want = O.nms(scores, boxes, classes, 0.45, ndet, diou)
            got = c_nms(lib, scores, boxes, classes, 0.45, ndet, diou)
            for x, y in zip(got, want):
                np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("radius", [0.0, 1.5])
def test_match_targets_matches_the_numpy_oracle(lib, radius):
    rs = np.random.RandomState(3)
    B, Gt, C, stride, size = 3, 9, 7, 16, (9, 11)
    anchors = O.generate_anchors(stride, [1, 2, 0.5], [2.0, 2.828])
    t = np.full((B, Gt, 5), -1, F32)
    for b, n in enumerate((9, 4, 0)):  # a full image, a padded one, an empty one
        t[b, :n, :2] = rs.uniform(0, 120, (n, 2))
        t[b, :n, 2:4] = rs.uniform(8, 90, (n, 2))
        t[b, :n, 4] = rs.randint(0, C, n)
    want = O.extract_targets(t, OrderedDict([(stride, anchors)]), C, stride, size, (0.5, 0.4), radius)
    got = c_match(lib, t, anchors, C, stride, size, (0.5, 0.4), radius)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(got[2], want[2])
    want = O.extract_targets(t, OrderedDict([(stride, anchors)]), C, stride, size, [[0.5, 4.0]], radius)
    got = c_match(lib, t, anchors, C, stride, size, (0.5, 4.0), radius, by_scale=True)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(got[2], want[2])
