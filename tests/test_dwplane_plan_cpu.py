"""Host logic of the whole-row depthwise training kernels (csrc/ssdk_dwplane.hip), without a GPU: the plan the library
reports through ssdk_dwconv_plan keeps its invariants over a sweep of shapes, and a numpy model of the kernels' index
arithmetic -- staging into the padded LDS rows, (piece, row, segment) units, the stride-2 input gradient's parity rules --
run on THAT plan reproduces a direct depthwise convolution and its two gradients (reference: torchvision-style
InvertedResidual depthwise 3x3, mobilenet.py:56-76 of the reference, under F.conv2d semantics)."""
import ctypes
import itertools

import numpy as np
import pytest

from ssds import _native as N

F32, BF16 = 0, 1
NAMES = ("G", "T", "TR", "seg", "LD", "SR", "CH", "UP", "nwg", "groups", "lds", "Ht")


def plan(kind, n, c, h, w, s, dtype):
    out = (ctypes.c_int32 * 12)()
    rc = N.lib.ssdk_dwconv_plan(kind, n, c, h, w, s, dtype, out)
    assert rc in (0, 1)
    return dict(zip(NAMES, out)) if rc == 0 else None


def conv_out(h, s):
    return (h + 2 - 3) // s + 1


SHAPES = [(64, 32, 150, 150), (64, 96, 150, 150), (64, 144, 75, 75), (64, 192, 38, 38), (64, 384, 19, 19), (64, 960, 10, 10),
          (64, 32, 256, 256), (64, 144, 128, 128), (64, 960, 16, 16), (3, 5, 1, 1), (2, 8, 5, 130), (1, 2, 300, 40),
          (7, 3, 2, 3), (1, 1, 7, 1000), (2, 2, 3, 3000), (1, 1, 700, 9)]


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("kind,s", [(0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (2, 2)])
def test_plan_invariants(kind, s, dtype):
    for n, c, h, w in SHAPES:
        p = plan(kind, n, c, h, w, s, dtype)
        ho, wo = conv_out(h, s), conv_out(w, s)
        ht, wt = (h, w) if kind == 1 else (ho, wo)
        hs, ws = (ho, wo) if kind == 1 else (h, w)
        if p is None:  # only rows too wide for the LDS budget fall back to the tiled kernels
            assert max(w, wo) > 1000, (kind, s, n, c, h, w)
            continue
        esz = 4 if dtype == F32 else 2
        assert p["Ht"] == ht and p["seg"] == (wt + 7) // 8 and p["UP"] == p["TR"] * p["seg"]
        assert p["lds"] == p["G"] * p["SR"] * p["LD"] * esz <= 32768
        assert 1 <= p["G"] * p["UP"] <= 1024                       # at most four units per thread
        assert p["T"] * p["TR"] >= ht and (p["T"] - 1) * p["TR"] < ht    # the bands cover the plane, none is empty
        assert p["G"] == 1 or p["T"] == 1                          # several images per workgroup only for whole planes
        assert p["groups"] == -(-n // p["G"]) * p["T"] and p["nwg"] == p["groups"] * c
        if kind == 1 and s == 2 and p["T"] > 1:
            assert p["TR"] % 2 == 0                                # the stride-2 input gradient pairs rows
        x0 = 1 if dtype == F32 else 8                              # LDS column of pixel 0
        assert p["LD"] >= ws + x0 + 1                              # a zero column right of the row
        if dtype == F32:
            assert p["LD"] % 8 == 1 and p["CH"] * 8 == p["LD"] - 1
        else:
            assert p["LD"] % 8 == 0 and p["CH"] * 8 == p["LD"] - 8
        sr = (p["TR"] >> 1) + 2 if (kind == 1 and s == 2) else (1 if kind == 1 else s) * (p["TR"] - 1) + 3
        assert p["SR"] == sr


# ---- a numpy model of the kernels on the library's plan ---------------------------------------------------------------
def _stage(p, a, n0, c, ys0, hs, ws, x0col):
    lds = np.full((p["G"] * p["SR"], p["LD"]), np.nan, np.float32)   # NaN: a column nobody staged shows up in the result
    for lr in range(p["G"] * p["SR"]):
        i, sr = divmod(lr, p["SR"])
        ys = ys0 + sr
        lds[lr, :x0col] = 0.0
        row = np.zeros(p["CH"] * 8, np.float32)
        if n0 + i < a.shape[0] and 0 <= ys < hs:
            row[:ws] = a[n0 + i, c, ys]
        lds[lr, x0col:x0col + p["CH"] * 8] = row
    return lds


def _units(p, n, n0, r0):
    for u in range(p["G"] * p["UP"]):
        i, ru = divmod(u, p["UP"])
        r, g = divmod(ru, p["seg"])
        if n0 + i < n and r0 + r < p["Ht"]:
            yield i, r, g


def _workgroups(p, c):
    for grp in range(p["groups"]):
        nb, band = divmod(grp, p["T"])
        for ch in range(c):
            yield grp, ch, nb * p["G"], band * p["TR"]


def model_fwd(kind, a, w, s, h, wd, dtype):
    """kind 0: y from x; kind 1: dx from dy (a = dy; h, wd = the INPUT dims)."""
    n, c = a.shape[:2]
    p = plan(kind, n, c, h, wd, s, dtype)
    assert p is not None
    x0 = 1 if dtype == F32 else 8
    hs, ws = a.shape[2:]
    wt = wd if kind == 1 else conv_out(wd, s)
    out = np.full((n, c, p["Ht"], wt), np.nan, np.float32)
    for grp, ch, n0, r0 in _workgroups(p, c):
        if kind == 1 and s == 2:
            lds = _stage(p, a, n0, ch, (r0 >> 1) - 1, hs, ws, x0)
            for i, r, g in _units(p, n, n0, r0):
                iy = r0 + r
                hi = lds[i * p["SR"] + ((iy + 1) >> 1) - (r0 >> 1) + 1]
                lo = lds[i * p["SR"] + ((iy + 1) >> 1) - (r0 >> 1)]
                whi = w[ch, 0:3] if iy & 1 else w[ch, 3:6]
                wlo = w[ch, 6:9] if iy & 1 else np.zeros(3, np.float32)
                col = 4 * g + x0
                for e in range(8):
                    if 8 * g + e >= wt:
                        continue
                    if e & 1:
                        v = (whi[0] * hi[col + (e + 1) // 2] + whi[2] * hi[col + (e - 1) // 2] + wlo[0] * lo[col + (e + 1) // 2]
                             + wlo[2] * lo[col + (e - 1) // 2])
                    else:
                        v = whi[1] * hi[col + e // 2] + wlo[1] * lo[col + e // 2]
                    out[n0 + i, ch, iy, 8 * g + e] = v
        else:
            st = 1 if kind == 1 else s
            lds = _stage(p, a, n0, ch, st * r0 - 1, hs, ws, x0)
            for i, r, g in _units(p, n, n0, r0):
                for e in range(8):
                    if 8 * g + e >= wt:
                        continue
                    v = 0.0
                    for ky in range(3):
                        row = lds[i * p["SR"] + st * r + ky]
                        for kx in range(3):
                            t = ky * 3 + kx
                            v += w[ch, 8 - t if kind == 1 else t] * row[st * (8 * g + e) + kx + x0 - 1]
                    out[n0 + i, ch, r0 + r, 8 * g + e] = v
    return out


def model_wgrad(x, dy, s, dtype):
    n, c, h, wd = x.shape
    p = plan(2, n, c, h, wd, s, dtype)
    assert p is not None
    x0 = 1 if dtype == F32 else 8
    part = np.zeros((p["groups"], c, 9), np.float64)
    for grp, ch, n0, r0 in _workgroups(p, c):
        lds = _stage(p, x, n0, ch, s * r0 - 1, h, wd, x0)
        for i, r, g in _units(p, n, n0, r0):
            for e in range(8):
                if 8 * g + e >= dy.shape[3]:
                    continue
                for ky in range(3):
                    row = lds[i * p["SR"] + s * r + ky]
                    for kx in range(3):
                        part[grp, ch, ky * 3 + kx] += dy[n0 + i, ch, r0 + r, 8 * g + e] * row[s * (8 * g + e) + kx + x0 - 1]
    return part.sum(0)


def direct(x, w, dy, s):
    """depthwise 3x3, pad 1: y, dx, dw by the definitions (float64)."""
    n, c, h, wd = x.shape
    ho, wo = conv_out(h, s), conv_out(wd, s)
    xp = np.zeros((n, c, h + 2, wd + 2))
    xp[:, :, 1:-1, 1:-1] = x
    y = np.zeros((n, c, ho, wo))
    dxp = np.zeros_like(xp)
    dw = np.zeros((c, 9))
    for ky, kx in itertools.product(range(3), range(3)):
        win = xp[:, :, ky:ky + s * (ho - 1) + 1:s, kx:kx + s * (wo - 1) + 1:s]
        y += w[None, :, ky * 3 + kx, None, None] * win
        dw[:, ky * 3 + kx] = (win * dy).sum((0, 2, 3))
        dxp[:, :, ky:ky + s * (ho - 1) + 1:s, kx:kx + s * (wo - 1) + 1:s] += w[None, :, ky * 3 + kx, None, None] * dy
    return y, dxp[:, :, 1:-1, 1:-1], dw


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("n,c,h,w,s", [(2, 2, 20, 24, 1), (2, 1, 33, 31, 2), (1, 1, 150, 150, 1), (1, 1, 150, 150, 2), (2, 1, 75, 75, 2),
                                       (5, 2, 19, 19, 1), (45, 1, 10, 10, 1), (3, 1, 10, 10, 2), (2, 1, 2, 3, 2), (1, 1, 1, 1, 1),
                                       (1, 1, 7, 1000, 1), (1, 1, 300, 40, 2)])
def test_index_model_on_the_library_plan_matches_a_direct_convolution(n, c, h, w, s, dtype):
    rng = np.random.default_rng(n * 1000 + h * 10 + s)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = rng.standard_normal((c, 9)).astype(np.float32)
    dy = rng.standard_normal((n, c, conv_out(h, s), conv_out(w, s))).astype(np.float32)
    y, dx, dw = direct(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), s)
    np.testing.assert_allclose(model_fwd(0, x, wt, s, h, w, dtype), y, rtol=0, atol=2e-5)
    np.testing.assert_allclose(model_fwd(1, dy, wt, s, h, w, dtype), dx, rtol=0, atol=2e-5)
    np.testing.assert_allclose(model_wgrad(x, dy, s, dtype), dw, rtol=1e-6, atol=1e-4)
