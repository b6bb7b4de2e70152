"""The weight image of the row-pair block kernel (include/ssdk.h, ssdk_mbconv_desc.w_image; built by MbPack.image() with
vectorised index arithmetic) against a restatement of the DOCUMENTED layout in plain loops: sampled entries of the weight
fragments, every entry of the constant blocks, the size the library reports, and the shapes no instance takes.  Host logic
only: runs without a GPU (the library's size query is a host function)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ssds.pytorch_amd")]

# cin, cout, stride, input width -- one per instance of csrc/ssdk_mbk.hip
SHAPES = [(160, 160, 1, 16), (160, 320, 1, 16), (96, 160, 2, 32), (64, 64, 1, 32), (64, 96, 1, 32), (96, 96, 1, 32),
          (32, 64, 2, 64), (32, 32, 1, 64)]


def _pack(cin, cout, stride, dtype):
    import torch
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import InvertedResidual

    torch.manual_seed(cin * 7 + cout + stride)
    blk = InvertedResidual(cin, cout, stride, 6).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    return FC.MbPack(groups_of(blk.conv), blk.use_res_connect, dtype)


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("cin,cout,stride,w_in", SHAPES)
def test_image_follows_the_documented_layout(cin, cout, stride, w_in, dtype_name):
    import torch
    from ssds import _native as N

    dtype = getattr(torch, dtype_name)
    pk = _pack(cin, cout, stride, dtype)
    got = pk.image(w_in)
    assert got is not None, "no instance for a shape the kernel lists"
    nw, img = got
    wo = w_in // stride
    nfo_c = ctypes.c_int(0)
    need = int(N.lib.ssdk_mbk_image_bytes(cin, pk.chid, cout, stride, wo, nw, ctypes.byref(nfo_c)))
    nfo = nfo_c.value
    assert need == img.numel() * 2 and img.dtype == torch.int16 and img.is_contiguous()
    im = img.numpy().view(np.uint16)
    ks, nch = cin // 32, pk.chid // 16
    nchw = (nch + nw - 1) // nw
    npair, halves = (nchw + 1) // 2, cout // (16 * nfo)
    pair_words = (2 * ks + nfo) * 512  # 16-bit words of one (half, slice, pair) block: 2 KS + NFO KiB
    we = pk.e.w.reshape(pk.chid, cin).view(torch.int16).numpy().view(np.uint16)   # expand weights (BN scale folded in), model dtype
    wp = pk.wp.reshape(cout, pk.chid).view(torch.int16).numpy().view(np.uint16)  # projection weights, fp16
    rs = np.random.RandomState(cin + cout)

    def expand_entry(w, t, cc, k, lane, j):
        loc = 2 * t + cc
        chunk = w * nchw + loc
        if loc >= nchw or chunk >= nch:
            return 0
        return int(we[chunk * 16 + (lane & 15), 32 * k + 8 * (lane >> 4) + j])

    def project_entry(h, w, t, f, lane, j):
        loc = 2 * t + j // 4  # (the k-permutation: element j of a lane <-> chunk 2t + j / 4, channel 4 fg + j % 4)
        chunk = w * nchw + loc
        if loc >= nchw or chunk >= nch:
            return 0
        return int(wp[h * 16 * nfo + 16 * f + (lane & 15), chunk * 16 + 4 * (lane >> 4) + j % 4])

    for _ in range(4000):
        h, w, t = rs.randint(halves), rs.randint(nw), rs.randint(npair)
        base = ((h * nw + w) * npair + t) * pair_words
        lane, j = rs.randint(64), rs.randint(8)
        if rs.randint(2):
            cc, k = rs.randint(2), rs.randint(ks)
            at = base + ((cc * ks + k) * 64 + lane) * 8 + j
            assert int(im[at]) == expand_entry(w, t, cc, k, lane, j), ("expand", h, w, t, cc, k, lane, j)
        else:
            f = rs.randint(nfo)
            at = base + 2 * ks * 512 + (f * 64 + lane) * 8 + j
            assert int(im[at]) == project_entry(h, w, t, f, lane, j), ("project", h, w, t, f, lane, j)

    # ---- per-slice constants: [NCHW][4 fg] f32x4 expand bias | [NCHW][9 taps][4 fg] 4 x fp16 taps | [NCHW][4 fg] 4 x fp16 bias / 6
    misc_words = ((nchw * 384 + 1023) // 1024) * 512
    misc0 = halves * nw * npair * pair_words
    be = pk.e.bias.float().numpy()
    wd = pk.wd.reshape(9, pk.chid).numpy()                                  # fp16 [tap][channel]
    bd6 = (pk.bd.float() * torch.tensor(1.0 / 6.0)).to(torch.float16).numpy()
    for w in range(nw):
        blk = im[misc0 + w * misc_words: misc0 + (w + 1) * misc_words]
        be_got = blk[: nchw * 32].view(np.float32).reshape(nchw, 4, 4)
        wd_got = blk[nchw * 32: nchw * 32 + nchw * 144].view(np.float16).reshape(nchw, 9, 4, 4)
        bd_got = blk[nchw * 176: nchw * 176 + nchw * 16].view(np.float16).reshape(nchw, 4, 4)
        for c in range(nchw):
            chunk = w * nchw + c
            for g in range(4):
                for q in range(4):
                    ch = chunk * 16 + 4 * g + q
                    live = chunk < nch
                    assert be_got[c, g, q] == (be[ch] if live else 0.0)
                    assert bd_got[c, g, q] == (bd6[ch] if live else 0.0)
                    for tap in range(9):
                        assert wd_got[c, tap, g, q] == (wd[tap, ch] if live else 0.0)
        assert not blk[nchw * 192:].any()  # padding up to the KiB boundary

    # ---- projection BN per half: [NFO][4 fg][scale x 6 (4) | bias (4)] fp32, 2 KiB per half
    spb0 = misc0 + nw * misc_words
    sp6, bp = (pk.p.scale.float() * 6.0).numpy(), pk.p.bias.float().numpy()
    for h in range(halves):
        blk = im[spb0 + h * 1024: spb0 + (h + 1) * 1024].view(np.float32)
        for f in range(nfo):
            for g in range(4):
                for q in range(4):
                    co = h * 16 * nfo + 16 * f + 4 * g + q
                    assert blk[(f * 4 + g) * 8 + q] == sp6[co] and blk[(f * 4 + g) * 8 + 4 + q] == bp[co]
    assert spb0 + halves * 1024 == im.size


def test_shapes_without_an_instance_have_no_image():
    import torch

    pk = _pack(24, 32, 2, torch.bfloat16)      # Cin = 24: not a multiple of the MFMA's 32 k
    assert pk.image(128) is None
    pk = _pack(160, 160, 1, torch.bfloat16)
    assert pk.image(20) is None                # 20-pixel-wide map: not one of 16 / 32 / 64
    nfo = ctypes.c_int(-1)
    from ssds import _native as N
    assert N.lib.ssdk_mbk_image_bytes(160, 960, 160, 1, 20, 4, ctypes.byref(nfo)) == 0
