"""Numerics of the fused conv kernels (MFMA implicit GEMM, depthwise, stem) against a plain PyTorch fp32
reference of the same op on the same (bf16/f16-rounded) operands.  Tolerance: the kernels accumulate in
fp32 and round once to the 16-bit output type, so |err| <= ~2^-8 relative to the output magnitude (bf16)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GP = [  # cin, cout, stride, h, w, n, act, residual mode (None | "same" | "half" | "post")
    (256, 1024, 1, 40, 40, 8, "none", "post"),   # ResNet-50 layer3 expansion + block tail: eight n-tiles per pixel tile, four k-steps
    (1024, 256, 1, 40, 40, 8, "relu", None),      # the reduction: sixteen k-steps, two n-tiles
    (512, 256, 1, 48, 56, 6, "none", "half"),    # FPN lateral + nearest x2 top-down add; ragged last pixel tile
    (256, 512, 2, 80, 80, 4, "none", None),      # downsample branch: every second pixel of every second row
    (288, 132, 1, 64, 50, 4, "relu6", "same"),   # K tail of 32 (one k-substep), channel tail (132 = 128 + 4), plain residual
    (264, 128, 1, 100, 64, 4, "silu", None),     # K tail of 8, sigmoid-type activation
    (2048, 512, 1, 20, 20, 16, "relu", None),     # 32 k-steps, fewer tiles than CUs
]


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,stride,h,w,n,act,rmode", GP)
def test_pointwise_persistent_gemm_kernel(cin, cout, stride, h, w, n, act, rmode, dtype_name):
    """ssdk_gemmp.hip (round 6): 1x1 convolutions with Cin >= 256 as a persistent NT GEMM (256 x 128 x 64 tiles, three LDS
    stages by LDS-DMA, epilogue from the accumulator registers through a row permutation of the weights) against the fp32
    layer: every residual mode of ssdk_conv, both strides, K / channel / pixel tails, one and several tiles per workgroup."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin + cout + h + stride)
    conv = nn.Conv2d(cin, cout, 1, stride, 0, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda()
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    pack = FC.ConvPack(conv, bn, act, dtype)
    if rmode is None:
        want = _ref(x, conv, bn, act)
        y = FC.conv_native(x.cuda(), pack)
    elif rmode == "post":
        res = torch.randn(n, cout, ho, wo).to(dtype)
        lin = _ref(x, conv, bn, "none").to(dtype).float() + res.float()
        want = lin.clamp(min=0)
        y = FC.conv_native(x.cuda(), FC.ConvPack(conv, bn, "relu", dtype), residual=res.cuda(), res_mode=2)
    elif rmode == "same":
        res = torch.randn(n, cout, ho, wo).to(dtype)
        want = _ref(x, conv, bn, act, residual=res)
        y = FC.conv_native(x.cuda(), pack, residual=res.cuda())
    else:
        res = torch.randn(n, cout, ho // 2, wo // 2).to(dtype)
        want = _ref(x, conv, bn, act).to(dtype).float() + F.interpolate(res.float(), scale_factor=2, mode="nearest")
        y = FC.conv_native(x.cuda(), pack, residual=res.cuda(), res_mode=1)
    assert N.last_kernel() == "conv_gemmp_kernel", N.last_kernel()
    _check(y, want, dtype, "gemmp %d->%d s%d %s" % (cin, cout, stride, rmode), floor=1.0 if rmode else 0.125)
    y2 = FC.conv_native(x.cuda(), pack) if rmode is None else None  # (bit-reproducible: no atomics, fixed k order)
    assert y2 is None or torch.equal(y, y2)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(x, conv, bn, act, residual=None):
    import torch
    import torch.nn.functional as F

    w = conv.weight.detach().float().cpu()
    y = F.conv2d(x.float().cpu(), w, None, conv.stride, conv.padding, 1, conv.groups)
    from ssds.modeling.layers.fused_conv import fold_bn

    scale, bias = fold_bn(conv, bn)
    y = y * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    if act == "relu":
        y = y.clamp(min=0)
    elif act == "relu6":
        y = y.clamp(0, 6)
    elif act == "silu":
        y = y * torch.sigmoid(y)
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    if residual is not None:
        y = y.to(x.dtype).float() + residual.float().cpu()
    return y


def _check(got, want, dtype, what, floor=0.125):
    """Per ELEMENT: |got - want| <= tol * max(|want|, floor * max|want|) + 1e-3, tol = 4 half-ulps of the output dtype.
    Single layers (floor 1/8): the reference is the fp32 layer on the same rounded inputs, so what is left is the output
    rounding (relative to the element) plus the accumulation order (relative to the sum of products, hence the floor).
    Round 2 bounded only the maximum error by tol * max|want|, which let a small output be wrong by its whole magnitude.
    floor=1 keeps that bound for results that are SUMS of rounded terms of full magnitude -- a residual added in the model
    dtype, the fused blocks with their fp16 internal tensors -- where an element near zero carries the rounding of the
    operands that cancelled."""
    import torch

    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9
    g = got.float().cpu().contiguous()
    assert g.shape == want.shape, (what, g.shape, want.shape)
    scale = max(float(want.abs().max()), 1e-3)
    excess = (g - want).abs() - (tol * want.abs().clamp(min=floor * scale) + 1e-3)
    worst = int(excess.argmax())
    assert float(excess.max()) <= 0, "%s: element %d got %.6g want %.6g (scale %.3g)" % (
        what, worst, float(g.flatten()[worst]), float(want.flatten()[worst]), scale)


DENSE = [
    # cin, cout, k, stride, h, w, n, act
    (16, 96, 1, 1, 20, 24, 2, "relu6"),
    (24, 144, 1, 1, 9, 7, 3, "relu6"),      # Cin not a multiple of the 32-wide k chunk
    (96, 24, 1, 1, 17, 17, 2, "none"),      # small N tile
    (144, 32, 1, 1, 8, 8, 1, "none"),
    (32, 16, 1, 1, 33, 31, 2, "none"),      # N = 16 tile
    (96, 504, 3, 1, 19, 19, 2, "none"),     # SSD head geometry (level 0 @300)
    (320, 256, 1, 1, 10, 10, 2, "relu"),    # extras 1x1
    (256, 512, 3, 2, 10, 10, 2, "relu"),    # extras 3x3 stride 2
    (64, 64, 3, 1, 5, 5, 1, "silu"),
    (128, 40, 3, 2, 7, 9, 2, "sigmoid"),    # odd sizes, Cout not a multiple of 16
    (256, 256, 3, 1, 40, 40, 1, "relu"),    # FPN tower geometry
]


@pytest.mark.parametrize("cin,cout,k,stride,h,w,n,act", DENSE)
@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_dense_conv(cin, cout, k, stride, h, w, n, act, dtype_name):
    import torch
    import torch.nn as nn
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin * 131 + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
    bn = nn.BatchNorm2d(cout)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    conv, bn = conv.cuda(), bn.cuda()
    pack = FC.ConvPack(conv, bn, act, dtype)
    assert pack.kind == "dense"
    y = FC.conv_native(x.cuda(), pack)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == dtype
    _check(y, _ref(x, conv, bn, act), dtype, "dense nhwc")
    y2 = FC.conv_native(x.cuda(), pack, nchw_out=True)  # NCHW epilogue (head layout)
    assert y2.is_contiguous()
    _check(y2, _ref(x, conv, bn, act), dtype, "dense nchw")


HALO, G256, SMALLMAP, SHORT = "conv3x3_halo_kernel", "conv_gemm256_kernel", "conv_smallmap_kernel", "conv3x3_short_kernel"
LARGE = [
    # cin, cout, k, stride, h, w, n, act, kernel the layer must be dispatched to
    (96, 504, 3, 1, 32, 32, 8, "none", HALO),      # SSD head L0: 16x16 patches, last slab holds 32 channels
    (40, 128, 3, 1, 32, 32, 24, "relu", HALO),     # Cin < 64: one partial slab, masked lanes
    (320, 504, 3, 1, 16, 16, 32, "sigmoid", HALO), # SSD head L1: one whole map per tile
    (64, 720, 3, 1, 19, 19, 16, "relu", HALO),     # odd map: ragged patches, scalar NCHW stores
    (256, 256, 3, 1, 40, 40, 8, "relu", HALO),     # FPN tower P4: 6x40 patches
    (128, 256, 3, 1, 10, 10, 128, "relu", HALO),   # two whole maps per tile
    (384, 504, 3, 1, 8, 8, 96, "none", HALO),      # four whole maps per tile (Cin the small-map kernel has no instance for)
    (512, 504, 3, 1, 8, 8, 96, "none", SMALLMAP),  # SSD head L2: 128 pixels (two maps) x 64 channels per workgroup
    (512, 504, 3, 1, 8, 8, 5, "sigmoid", SMALLMAP),   # few images: 64 pixels per workgroup, a ragged last group
    (256, 504, 3, 1, 4, 4, 66, "none", SMALLMAP),  # SSD head L3: four maps per workgroup, two surplus images
    (256, 504, 3, 1, 2, 2, 35, "relu", SMALLMAP),  # SSD head L4: sixteen maps per workgroup
    (128, 504, 3, 1, 1, 1, 64, "silu", SMALLMAP),  # SSD head L5: 1x1 maps, only the centre tap sees data
    (256, 40, 3, 1, 2, 4, 9, "relu6", SMALLMAP),   # 2x4 map, a single partial channel range
    (256, 256, 3, 1, 5, 5, 32, "relu", SMALLMAP),  # FPN tower on the 5x5 level: 25 pixels do not divide 64 (two maps + 14 idle slots)
    (256, 720, 3, 1, 7, 7, 16, "sigmoid", SMALLMAP),  # BiFPN head on the 7x7 level: one map per workgroup
    (128, 504, 3, 1, 3, 5, 9, "none", SMALLMAP),   # 15 pixels: four maps per workgroup, a ragged last group
    # conv3x3_short_kernel: (kernel of the NHWC call, kernel of the NCHW call) -- NHWC takes none / relu / relu6, NCHW none / sigmoid / silu
    (96, 504, 3, 1, 32, 32, 64, "sigmoid", (HALO, SHORT)), # SSD head L0 at bench size: a 128-pixel patch x 4 channel tiles per workgroup
    (128, 256, 3, 1, 40, 40, 20, "relu", (SHORT, HALO)),   # ragged patches (40 = 2.5 x 16), two channel tiles
    (32, 132, 3, 1, 64, 64, 8, "none", SHORT),             # one 32-channel slice, a ragged second channel tile
    (64, 160, 3, 1, 24, 48, 32, "silu", (HALO, SHORT)),    # non-square map
    (256, 256, 3, 1, 40, 40, 32, "relu", (SHORT, HALO)),   # FPN tower layer: one workgroup per CU, A fragments one k-step ahead
    (256, 720, 3, 1, 24, 24, 48, "sigmoid", (HALO, SHORT)),# FPN head: six channel tiles, the last one ragged
    (256, 512, 3, 2, 16, 16, 64, "relu", SMALLMAP),   # first SSD extra: stride 2 from a 16x16 onto an 8x8 map (skewed input rows)
    (128, 256, 3, 2, 8, 8, 33, "relu6", SMALLMAP),    # stride 2 onto 4x4: four images per workgroup, a ragged last group
    (128, 100, 3, 2, 4, 4, 70, "silu", SMALLMAP),     # stride 2 onto 2x2, a partial channel range
    (1024, 256, 1, 1, 40, 40, 24, "relu", G256),   # wide 1x1 with a long K (the flat-K 256 x 256 kernel takes K >= 1024 only)
    (128, 320, 3, 2, 64, 64, 32, "relu6", G256),   # 3x3 stride 2
    (1064, 200, 1, 1, 45, 50, 16, "silu", G256),   # ragged everything
    # wide 1x1, short K: NHWC streams through pwflow_kernel (round 4), NCHW takes the 128-row GEMM kernel (several workgroups per CU)
    (64, 256, 1, 1, 64, 64, 8, "relu", ("pwflow_kernel", "conv_gemm_kernel")),
]


@pytest.mark.parametrize("cin,cout,k,stride,h,w,n,act,kernel", LARGE)
@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_large_tile_kernels(cin, cout, k, stride, h, w, n, act, kernel, dtype_name):
    """The 256-row tile kernels (halo-tile 3x3, 256x256 flat-K) only take layers with enough tiles to fill the
    chip, the small-map kernel the 3x3 layers on maps of <= 64 pixels; these shapes are sized to reach them, and the
    dispatch is asserted."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin * 7 + cout + h)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
    bn = nn.BatchNorm2d(cout)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    conv, bn = conv.cuda(), bn.cuda()
    pack = FC.ConvPack(conv, bn, act, dtype)
    want = _ref(x, conv, bn, act)
    k_nhwc, k_nchw = kernel if isinstance(kernel, tuple) else (kernel, kernel)
    y = FC.conv_native(x.cuda(), pack)
    assert N.last_kernel() == k_nhwc, N.last_kernel()
    _check(y, want, dtype, "large nhwc")
    y2 = FC.conv_native(x.cuda(), pack, nchw_out=True)
    assert N.last_kernel() == k_nchw, N.last_kernel()
    _check(y2, want, dtype, "large nchw")


@pytest.mark.parametrize("cin,h,w,n", [(512, 8, 8, 70), (256, 4, 4, 66), (256, 2, 2, 35), (128, 1, 1, 64), (256, 2, 4, 9)])
def test_small_map_kernel_on_krsc_and_on_fragment_major_weights(cin, h, w, n, monkeypatch):
    """conv_smallmap reads its weights either from the KRSC tensor or from the fragment-major image ConvPack.frag() builds
    (ssdk.h ssdk_weight_frag_bytes); both paths against torch, and against each other (same products, another k order)."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(cin + h)
    conv = nn.Conv2d(cin, 504, 3, 1, 1, bias=True)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    conv = conv.cuda()
    want = _ref(x, conv, None, "none")
    outs = []
    for use in (True, False):
        monkeypatch.setattr(FC, "USE_WFRAG", use)
        pack = FC.ConvPack(conv, None, "none", dtype)
        assert (pack.frag() is not None) == use
        if use:  # the image itself: block (g, ks) lane (fg, fr) holds w[16g + fr][32ks + 8fg .. +7]
            img, w2d = pack.frag(), pack.w.reshape(504, -1)
            assert img.shape == (32, 9 * cin // 32, 4, 16, 8) and img.numel() * 2 == N.lib.ssdk_weight_frag_bytes(504, 9 * cin)
            assert torch.equal(img[3, 5, 2, 7], w2d[16 * 3 + 7, 32 * 5 + 16:32 * 5 + 24])
            assert float(img[31, :, :, 8:].float().abs().max()) == 0.0  # rows 504..511
        y = FC.conv_native(x.cuda(), pack, nchw_out=True, split=24, act2="sigmoid")
        assert N.last_kernel() == SMALLMAP
        got = torch.cat([y[0].float(), y[1].float()], 1)
        ref = torch.cat([want[:, :24], torch.sigmoid(want[:, 24:])], 1)
        _check(got, ref, dtype, "small map, frag=%s" % use)
        outs.append(got)
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_short_k_kernel_split_heads(dtype_name):
    """SSD head L0 (96 -> 24 loc | 480 conf, NCHW split, sigmoid on conf only) on conv3x3_short_kernel, and the same layer on
    the halo kernel (no fragment-major image: the short kernel does not apply)."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(5)
    conv = nn.Conv2d(96, 504, 3, 1, 1, bias=True)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(64, 96, 32, 32).to(dtype)
    conv = conv.cuda()
    want = _ref(x, conv, None, "none")
    ref = torch.cat([want[:, :24], torch.sigmoid(want[:, 24:])], 1)
    pack = FC.ConvPack(conv, None, "none", dtype)
    y = FC.conv_native(x.cuda(), pack, nchw_out=True, split=24, act2="sigmoid")
    assert N.last_kernel() == SHORT, N.last_kernel()
    assert y[0].shape == (64, 24, 32, 32) and y[1].shape == (64, 480, 32, 32) and y[0].is_contiguous() and y[1].is_contiguous()
    got = torch.cat([y[0].float(), y[1].float()], 1)
    _check(got, ref, dtype, "short-K split heads")
    mp = pytest.MonkeyPatch()
    try:
        mp.setattr(FC, "USE_WFRAG", False)
        plain = FC.ConvPack(conv, None, "none", dtype)
        z = FC.conv_native(x.cuda(), plain, nchw_out=True, split=24, act2="sigmoid")
        assert N.last_kernel() == HALO, N.last_kernel()
    finally:
        mp.undo()
    got2 = torch.cat([z[0].float(), z[1].float()], 1)
    _check(got2, ref, dtype, "halo split heads")
    assert float((got - got2).abs().max()) <= 2e-2


def test_halo_kernel_residual_and_split_heads():
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(11)
    conv = nn.Conv2d(256, 256, 3, padding=1, bias=False).cuda()
    bn = nn.BatchNorm2d(256).cuda()
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(48, 256, 16, 16).to(dtype)
    res = torch.randn(48, 256, 16, 16).to(dtype)
    y = FC.conv_native(x.cuda(), FC.ConvPack(conv, bn, "none", dtype), residual=res.cuda())
    assert N.last_kernel() == HALO
    _check(y, _ref(x, conv, bn, "none", res), dtype, "halo residual", floor=1.0)
    loc = nn.Conv2d(96, 24, 3, padding=1).cuda()
    conf = nn.Conv2d(96, 480, 3, padding=1).cuda()
    for m in (loc, conf):
        m.weight.data = (m.weight.data * 3).to(dtype).float()
        m.bias.data.normal_(0, 0.5)
    f = torch.randn(8, 96, 32, 32).to(dtype)
    l, c = FC.conv_native(f.cuda(), FC.pack_heads(loc, conf, dtype), act="none", nchw_out=True, split=24,
                          act2="sigmoid")
    assert N.last_kernel() == HALO
    _check(l, _ref(f, loc, None, "none"), dtype, "halo split loc")
    _check(c, _ref(f, conf, None, "sigmoid"), dtype, "halo split conf")


@pytest.mark.parametrize("cin,cout,k,h,w,n,kernel", [
    (64, 256, 1, 16, 20, 2, "conv_gemm_kernel"),       # FPN lateral + top-down add, tiled kernel
    (1024, 256, 1, 40, 60, 16, G256),                  # same on the 256x256 kernel (K >= 1024; 300 tiles of 256 x 128: the persistent GEMM declines a launch of 1.2 rounds)
    (1024, 256, 1, 64, 64, 16, "conv_gemmp_kernel"),   # same on the persistent GEMM (round 6)
    (64, 64, 1, 2, 2, 8, "conv_wave_kernel"),          # same on the wave kernel
])
def test_conv_half_resolution_residual(cin, cout, k, h, w, n, kernel):
    """lateral(x) + upsample2x_nearest(coarse) as ONE launch (reference fpn.py:80-87)."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(h * 3 + cout)
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=True).cuda()
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    coarse = torch.randn(n, cout, h // 2, w // 2).to(dtype)
    want = _ref(x, conv, None, "none").to(dtype).float() + F.interpolate(coarse.float(), scale_factor=2, mode="nearest")
    y = FC.conv_native(x.cuda(), FC.ConvPack(conv, None, "none", dtype), residual=coarse.cuda(), res_mode=1)
    assert N.last_kernel() == kernel, N.last_kernel()
    _check(y, want, dtype, "half-resolution residual", floor=1.0)


@pytest.mark.parametrize("cin,cout,k,h,w,n,act,kernel", [
    (64, 64, 3, 14, 14, 2, "relu", "conv_gemm_kernel"),
    (128, 256, 3, 16, 16, 48, "relu", HALO),
    (1024, 512, 1, 40, 32, 16, "relu", G256),
    (1024, 512, 1, 64, 32, 16, "relu", "conv_gemmp_kernel"),
    (64, 64, 3, 2, 2, 8, "relu6", "conv_wave_kernel"),
])
def test_conv_activation_after_residual(cin, cout, k, h, w, n, act, kernel):
    """ResNet block tail: relu(bn(conv(x)) + identity) as one launch (res_mode bit 1)."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(cin + h)
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda()
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    res = torch.randn(n, cout, h, w).to(dtype)
    lin = _ref(x, conv, bn, "none").to(dtype).float() + res.float()
    want = lin.clamp(min=0) if act == "relu" else lin.clamp(0, 6)
    y = FC.conv_native(x.cuda(), FC.ConvPack(conv, bn, act, dtype), residual=res.cuda(), res_mode=2)
    assert N.last_kernel() == kernel, N.last_kernel()
    _check(y, want, dtype, "activation after residual", floor=1.0)


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("h,w,dtype_name", [(64, 96, "bf16"), (75, 53, "bf16"), (40, 40, "f16")])
def test_resnet_stem_and_maxpool(layout, h, w, dtype_name):
    """conv1 7x7/2 + bn1 + relu on the MFMA im2col kernel and maxpool 3x3/2 (nets/resnet.py:41-46) vs torch fp32."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(h + w)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(3, 3, h, w).to(dtype)
    want = _ref(x, conv, bn, "relu")
    conv, bn = conv.cuda(), bn.cuda()
    assert FC.StemPack.supported(conv, bn)
    xin = x.cuda() if layout == "nchw" else x.cuda().contiguous(memory_format=torch.channels_last)
    y = FC.stem7_native(xin, FC.StemPack(conv, bn, "relu", dtype))
    assert y.is_contiguous(memory_format=torch.channels_last)
    _check(y, want, dtype, "stem 7x7")
    pooled = FC.maxpool_native(y)
    assert torch.equal(pooled.float().cpu(), F.max_pool2d(y.float().cpu(), 3, 2, 1)), "maxpool must be exact"


@pytest.mark.parametrize("c,stride,h,w,n", [(64, 1, 20, 24, 2), (128, 2, 33, 31, 3), (288, 1, 7, 9, 2), (672, 2, 14, 14, 1),
                                            (288, 1, 56, 56, 2), (208, 2, 40, 24, 3)])
@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_grouped_conv_16_per_group(c, stride, h, w, n, dtype_name):
    """RegNetX bottleneck conv: 3x3, groups = C/16 (+ BN + ReLU) vs torch fp32; stride 1 and 2, ragged maps, channel counts
    that leave a partial block of groups (288 = 18 groups, 208 = 13).  Output maps at least 8 pixels wide run from an LDS halo
    tile (gconv3x3_g16_tile_kernel), narrower ones straight from L2 (gconv3x3_g16_kernel)."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(c + stride)
    conv = nn.Conv2d(c, c, 3, stride, 1, groups=c // 16, bias=False)
    bn = nn.BatchNorm2d(c)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, c, h, w).to(dtype)
    want = _ref(x, conv, bn, "relu")
    conv, bn = conv.cuda(), bn.cuda()
    pack = FC.ConvPack(conv, bn, "relu", dtype)
    assert pack.kind == "g16"
    y = FC.conv_native(x.cuda(), pack)
    wo = (w + 2 - 3) // stride + 1
    assert N.last_kernel() == ("gconv3x3_g16_tile_kernel" if wo >= 8 else "gconv3x3_g16_kernel"), N.last_kernel()
    _check(y, want, dtype, "grouped conv")


@pytest.mark.parametrize("net,outs,depth", [("ResNet18", [3, 4, 5], [128, 256, 512]), ("ResNet50", [4, 5], [1024, 2048]),
                                            ("RegNetX008", [2, 3, 4], [128, 288, 672])])
def test_ssd_on_resnet_plan_matches_torch(net, outs, depth):
    """SSD heads on a ResNet backbone: image -> heads as one recorded plan (stem kernel, maxpool, residual blocks
    with the ReLU after the add) vs the fp32 module."""
    import torch
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import fused_conv as FC

    torch.manual_seed(4)
    fl = [outs + ["Conv:S"], depth + [256]]
    nets_outputs, extras, hd = ssds.SSD.add_extras(fl, [6] * (len(outs) + 1), 5)
    model = ssds.SSD(getattr(nets, net)(outputs=nets_outputs), extras, hd, 5).eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05)
            m.running_var.uniform_(0.9, 1.1)
    x = torch.rand(2, 3, 96, 128)
    with torch.no_grad():
        rl, rc = model(x)
    model = model.cuda().to(torch.bfloat16)
    plans = FC.STATS["plan_runs"]
    with torch.no_grad():
        loc, conf = model(x.cuda().to(torch.bfloat16))
    assert FC.STATS["plan_runs"] == plans + 1, "the ResNet forward did not run as one plan"
    # a random-init 18/50-layer network amplifies the per-layer bf16 rounding (2^-8 relative) to a few percent; a
    # structural error (wrong residual, stride, layout) shows up as O(100 %): bound the relative L2 error
    for l, a, c, b in zip(loc, rl, conf, rc):
        assert l.shape == a.shape and c.shape == b.shape
        for got, want in ((l, a), (c, b)):
            err = float((got.float().cpu() - want).norm() / want.norm().clamp(min=1e-6))
            assert err < 0.06, err


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_fuse_kernel(dtype_name):
    """BiFPN weighted fusions (reference bifpn.py:41-62) against torch fp32."""
    import torch
    import torch.nn.functional as F
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(3)
    a = torch.randn(3, 64, 14, 10).to(dtype)
    up = torch.randn(3, 64, 7, 5).to(dtype)
    big = torch.randn(3, 64, 29, 21).to(dtype)  # odd: max_pool2d floors to 14 x 10
    skip = torch.randn(3, 64, 14, 10).to(dtype)
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    y = FC.fuse_native(cl(a), cl(up), None, (0.25, 0.75, 0.0), N.FUSE_UP2)
    want = 0.25 * a.float() + 0.75 * F.interpolate(up.float(), scale_factor=2, mode="nearest")
    _check(y, want, dtype, "top-down fusion")
    y = FC.fuse_native(cl(a), cl(big), cl(skip), (0.5, 0.3, 0.2), N.FUSE_POOL2, N.FUSE_SAME)
    want = 0.5 * a.float() + 0.3 * F.max_pool2d(big.float(), 2) + 0.2 * skip.float()
    _check(y, want, dtype, "bottom-up fusion")
    with pytest.raises(N.SsdkError):
        FC.fuse_native(cl(a), cl(torch.randn(3, 64, 9, 5).to(dtype)), None, (1, 1, 0), N.FUSE_POOL2)


def test_dense_conv_residual_and_split():
    import torch
    import torch.nn as nn
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(5)
    conv = nn.Conv2d(192, 32, 1, bias=False).cuda()
    bn = nn.BatchNorm2d(32).cuda()
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(2, 192, 13, 11).to(dtype)
    res = torch.randn(2, 32, 13, 11).to(dtype)
    y = FC.conv_native(x.cuda(), FC.ConvPack(conv, bn, "none", dtype), residual=res.cuda())
    _check(y, _ref(x, conv, bn, "none", res), dtype, "residual", floor=1.0)
    # loc | conf of one level as ONE GEMM, two NCHW outputs, sigmoid only on the conf part
    loc = nn.Conv2d(96, 24, 3, padding=1).cuda()
    conf = nn.Conv2d(96, 480, 3, padding=1).cuda()
    for m in (loc, conf):
        m.weight.data = (m.weight.data * 3).to(dtype).float()
        m.bias.data.normal_(0, 0.5)
    f = torch.randn(3, 96, 19, 19).to(dtype)
    pk = FC.pack_heads(loc, conf, dtype)
    l, c = FC.conv_native(f.cuda(), pk, act="none", nchw_out=True, split=24, act2="sigmoid")
    assert tuple(l.shape) == (3, 24, 19, 19) and tuple(c.shape) == (3, 480, 19, 19) and l.is_contiguous()
    _check(l, _ref(f, loc, None, "none"), dtype, "split loc")
    _check(c, _ref(f, conf, None, "sigmoid"), dtype, "split conf")


@pytest.mark.parametrize("c,stride,h,w", [(32, 1, 16, 16), (96, 2, 17, 15), (144, 1, 9, 9), (576, 2, 8, 8),
                                           (960, 1, 5, 3), (8, 2, 6, 6)])
def test_depthwise(c, stride, h, w):
    import torch
    import torch.nn as nn
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(c)
    conv = nn.Conv2d(c, c, 3, stride, 1, groups=c, bias=False).cuda()
    bn = nn.BatchNorm2d(c).cuda()
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(2, c, h, w).to(dtype)
    pack = FC.ConvPack(conv, bn, "relu6", dtype)
    assert pack.kind == "dw"
    _check(FC.conv_native(x.cuda(), pack), _ref(x, conv, bn, "relu6"), dtype, "depthwise")


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_stem(layout):
    import torch
    import torch.nn as nn
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16
    torch.manual_seed(1)
    conv = nn.Conv2d(3, 32, 3, 2, 1, bias=False).cuda()
    bn = nn.BatchNorm2d(32).cuda()
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    x = torch.rand(2, 3, 33, 30).to(dtype)
    xin = x.cuda() if layout == "nchw" else x.cuda().contiguous(memory_format=torch.channels_last)
    pack = FC.ConvPack(conv, bn, "relu6", dtype)
    assert pack.kind == "stem"
    # the stem keeps fp32 weights: reference with fp32 weights
    _check(FC.conv_native(xin, pack), _ref(x, conv, bn, "relu6"), dtype, "stem " + layout)


def test_ssd_mobilenetv2_plan_matches_torch():
    """Whole network (BASELINE config 1 geometry): recorded plan on the HIP kernels vs the same module in
    fp32 on torch.  bf16 activations through ~55 layers: compare with a loose relative tolerance, and the
    confidence maps (what decode thresholds) tightly in absolute terms."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import fused_conv as FC

    config.reset_cfg()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.cfg_from_file(os.path.join(root, "experiments", "cfgs", "ssd_mobilenetv2_300.yml"))
    torch.manual_seed(3)
    model = model_builder.create_model(cfg.MODEL).eval()
    for m in model.modules():  # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.8, 1.2)
    x = torch.rand(2, 3, 300, 300)
    with torch.no_grad():
        ref_loc, ref_conf = model.float()(x)  # CPU fp32 (torch path)
    model = model.cuda().to(torch.bfloat16)
    before = dict(FC.STATS)
    with torch.no_grad():
        loc, conf = model(x.cuda().to(torch.bfloat16))
        loc2, conf2 = model(x.cuda().to(torch.bfloat16))
    assert FC.STATS["plan_runs"] == before["plan_runs"] + 2, "the recorded plan did not run"
    assert FC.STATS["torch_fallback_layers"] == before["torch_fallback_layers"]
    for a, b in zip(conf, conf2):
        assert torch.equal(a, b)
    for l, rl, c, rc in zip(loc, ref_loc, conf, ref_conf):
        assert l.shape == rl.shape and c.shape == rc.shape and l.is_contiguous() and c.is_contiguous()
        assert float((c.float().cpu() - rc).abs().max()) < 2e-3   # sigmoid around 0.01
        scale = max(float(rl.abs().max()), 1e-2)
        assert float((l.float().cpu() - rl).abs().max()) < 0.05 * scale + 5e-3
    # A/B switch: the torch path on the same device agrees too
    os.environ["SSDK_FUSED_CONV"] = "0"
    try:
        with torch.no_grad():
            tl, tc = model(x.cuda().to(torch.bfloat16))
    finally:
        os.environ["SSDK_FUSED_CONV"] = "1"
    for c, t in zip(conf, tc):
        assert float((c.float() - t.float()).abs().max()) < 4e-3


MB = [  # cin, cout, stride, h, w  (hidden = 6*cin)
    (16, 24, 2, 32, 32), (24, 24, 1, 16, 16), (24, 32, 2, 19, 17), (32, 32, 1, 9, 11), (32, 64, 2, 16, 16),
    (64, 64, 1, 8, 8), (64, 96, 1, 10, 6), (96, 96, 1, 8, 8), (96, 160, 2, 13, 13), (160, 160, 1, 5, 7),
    (160, 320, 1, 8, 8),
]


@pytest.mark.parametrize("variant", ["tiled", "flow"])
@pytest.mark.parametrize("cin,cout,stride,h,w", MB + [(16, 24, 2, 61, 45), (24, 24, 1, 33, 30), (24, 32, 2, 40, 29), (32, 32, 1, 31, 17)])
def test_fused_inverted_residual_block(cin, cout, stride, h, w, variant):
    """The one-kernel MobileNetV2 block against the torch fp32 block on bf16-rounded weights: the LDS-tiled kernel
    and, where it exists (Cin <= 32), the register-flow kernel (ragged strips, several row segments)."""
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import InvertedResidual

    dtype = torch.bfloat16
    torch.manual_seed(cin * 7 + cout + stride)
    blk = InvertedResidual(cin, cout, stride, 6).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.randn(3, cin, h, w).to(dtype)
    with torch.no_grad():
        # reference with the same intermediate roundings the kernel applies (bf16 E and D tensors)
        y = x.float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(dtype).float()
        y = mods[1](y).to(dtype).float()
        y = mods[3](mods[2](y))
        if blk.use_res_connect:
            y = y.to(dtype).float() + x.float()
    blk = blk.cuda()
    groups = groups_of(blk.conv)
    assert FC.MbPack.supported(groups, blk.use_res_connect)
    pk = FC.MbPack(groups, blk.use_res_connect, dtype)
    got = FC.mbconv_native(x.cuda(), pk, variant=1 if variant == "flow" else -1)
    name = N.last_kernel()
    assert ("mbflow" in name) == (variant == "flow" and cin <= 32), name
    assert got.is_contiguous(memory_format=torch.channels_last)
    _check(got, y, dtype, "mbconv %d->%d s%d (%s)" % (cin, cout, stride, name), floor=1.0)


PW = [  # cin, cout, stride, h, w, n, act, residual mode (None | "same" | "half" | "post")
    (64, 256, 1, 64, 64, 4, "relu", None),       # ResNet-50 layer1 expansion: 16 fragments per wave, two k-steps
    (64, 256, 1, 72, 57, 4, "none", "post"),     # block tail: relu(bn(conv) + identity); ragged last pixel group
    (128, 512, 1, 40, 52, 8, "none", "post"),    # layer2 expansion: two slices of 256 channels, four k-steps
    (128, 64, 1, 128, 128, 1, "relu", None),     # a reduction: one slice of four fragments
    (128, 256, 2, 160, 160, 3, "none", None),    # stride 2: every second pixel of every second row
    (96, 192, 1, 128, 128, 1, "relu6", "same"),  # Cin not a multiple of 64 (three of four k-steps used), plain residual
    (128, 256, 1, 80, 96, 3, "none", "half"),    # lateral + nearest x2 top-down add
    (64, 64, 1, 160, 160, 1, "silu", None),
]


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,stride,h,w,n,act,rmode", PW)
def test_pointwise_streaming_kernel(cin, cout, stride, h, w, n, act, rmode, dtype_name):
    """ssdk_pwflow.hip: 1x1 convolutions with Cin <= 256 on large maps (B operand straight from global memory, weights
    resident in LDS, 8 consecutive channels per lane through a row permutation of the weights) against the fp32 layer,
    every residual mode of ssdk_conv and both strides."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin + cout + h + stride)
    conv = nn.Conv2d(cin, cout, 1, stride, 0, bias=False).cuda()
    bn = nn.BatchNorm2d(cout).cuda()
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    conv.weight.data = conv.weight.data.to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    pack = FC.ConvPack(conv, bn, act, dtype)
    if rmode is None:
        want = _ref(x, conv, bn, act)
        y = FC.conv_native(x.cuda(), pack)
    elif rmode == "post":
        res = torch.randn(n, cout, ho, wo).to(dtype)
        lin = _ref(x, conv, bn, "none").to(dtype).float() + res.float()
        want = lin.clamp(min=0)
        y = FC.conv_native(x.cuda(), FC.ConvPack(conv, bn, "relu", dtype), residual=res.cuda(), res_mode=2)
    elif rmode == "same":
        res = torch.randn(n, cout, ho, wo).to(dtype)
        want = _ref(x, conv, bn, act, residual=res)
        y = FC.conv_native(x.cuda(), pack, residual=res.cuda())
    else:
        res = torch.randn(n, cout, ho // 2, wo // 2).to(dtype)
        want = _ref(x, conv, bn, act).to(dtype).float() + F.interpolate(res.float(), scale_factor=2, mode="nearest")
        y = FC.conv_native(x.cuda(), pack, residual=res.cuda(), res_mode=1)
    assert N.last_kernel() == "pwflow_kernel", N.last_kernel()
    _check(y, want, dtype, "pwflow %d->%d s%d %s" % (cin, cout, stride, rmode), floor=1.0 if rmode else 0.125)


@pytest.mark.parametrize("variant", ["tiled", "flow"])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("h,w", [(64, 64), (37, 45), (130, 121), (70, 90), (129, 122), (66, 60)])
def test_fused_stem_block(layout, h, w, variant):
    """Stem (3x3/s2, BN, ReLU6) + expand-free first block (dw 3x3, pw 32->16) as ONE launch from the image.  NCHW images of
    even width take the register-flow kernel's dword loads (three per lane, row and strip; the left tap comes from the
    neighbour lane): widths that end inside a strip / on a strip boundary, odd heights (the masking instance runs the bottom
    segment on the 2-byte gather), several strips per wave and an odd number of strips."""
    import torch
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import ConvBNReLU6, InvertedResidual

    dtype = torch.bfloat16
    torch.manual_seed(11)
    stem = ConvBNReLU6(3, 32, stride=2).eval()
    blk = InvertedResidual(32, 16, 1, 1).eval()
    for m in list(stem.modules()) + list(blk.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.rand(2, 3, h, w).to(dtype)
    with torch.no_grad():
        y = stem(x.float()).to(dtype).float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(dtype).float()
        y = mods[2](mods[1](y))
    stem, blk = stem.cuda(), blk.cuda()
    sg, bg = groups_of(stem), groups_of(blk.conv)
    assert FC.MbPack.stem_supported(sg, bg)
    pk = FC.MbPack(bg, False, dtype, stem_group=sg[0])
    xin = x.cuda() if layout == "nchw" else x.cuda().contiguous(memory_format=torch.channels_last)
    from ssds import _native as N
    got = FC.mbconv_native(xin, pk, variant=1 if variant == "flow" else -1)
    name = N.last_kernel()
    assert ("mbflow" in name) == (variant == "flow"), name
    _check(got, y, dtype, "stem block %s (%s)" % (layout, name), floor=1.0)


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cmid,cout,h,w,n", [(512, 128, 256, 8, 8, 5), (256, 128, 256, 4, 4, 64), (256, 64, 128, 2, 2, 7),
                                                (256, 128, 256, 3, 5, 3), (128, 64, 128, 1, 1, 4)])
def test_extra_layer_pair_in_one_launch(cin, cmid, cout, h, w, n, dtype_name):
    """An SSD extra layer (Conv 1x1 + BN + ReLU -> Conv 3x3 / stride 2 + BN + ReLU) on a small map as one launch
    (ssdk_xpair) against the torch fp32 layers with the intermediate map rounded to the model dtype, and against the two
    single-layer launches it replaces."""
    import torch
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.basic_layers import ConvBNReLU
    from ssds.modeling.layers.planner import groups_of

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin + cmid + h)
    layer = torch.nn.Sequential(ConvBNReLU(cin, cmid, 1), ConvBNReLU(cmid, cout, 3, stride=2)).eval()
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    with torch.no_grad():
        y = layer[1](layer[0](x.float()).to(dtype).float())
    layer = layer.cuda()
    (c1, b1, a1), (c2, b2, a2) = groups_of(layer)
    p1, p2 = FC.ConvPack(c1, b1, a1, dtype), FC.ConvPack(c2, b2, a2, dtype)
    assert FC.xpair_supported(p1, p2, h, w)
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    got = FC.xpair_native(xc, p1, p2)
    assert got.shape == y.shape and got.is_contiguous(memory_format=torch.channels_last)
    _check(got, y, dtype, "extra layer %d>%d>%d @%dx%d" % (cin, cmid, cout, h, w))
    two = FC.conv_native(FC.conv_native(xc, p1), p2)
    assert float((got.float() - two.float()).abs().max()) <= 2e-2 * max(1.0, float(y.abs().max()))
    # the same launch on the KRSC weights (no fragment-major images)
    import pytest as _pytest
    assert p1.frag() is not None and p2.frag() is not None
    mp = _pytest.MonkeyPatch()
    try:
        mp.setattr(FC, "USE_WFRAG", False)
        q1, q2 = FC.ConvPack(c1, b1, a1, dtype), FC.ConvPack(c2, b2, a2, dtype)
        assert q1.frag() is None
        plain = FC.xpair_native(xc, q1, q2)
    finally:
        mp.undo()
    _check(plain, y, dtype, "extra layer on KRSC weights")
    assert float((got.float() - plain.float()).abs().max()) <= 2e-2 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("cin,cout,stride,h,w", [(16, 24, 2, 70, 45), (24, 24, 1, 47, 33), (32, 64, 2, 36, 36)])
def test_register_flow_block_fp16(cin, cout, stride, h, w):
    """The register-flow kernel in fp16 (expand on the f16 MFMA), ragged strips and several row segments, against the
    torch fp32 block with fp16-rounded intermediates; and the same block through the LDS-tiled kernel."""
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import InvertedResidual

    dtype = torch.float16
    torch.manual_seed(cin + cout + stride)
    blk = InvertedResidual(cin, cout, stride, 6).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.randn(5, cin, h, w).to(dtype)
    with torch.no_grad():
        y = x.float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(dtype).float()
        y = mods[1](y).to(dtype).float()
        y = mods[3](mods[2](y))
        if blk.use_res_connect:
            y = y.to(dtype).float() + x.float()
    blk = blk.cuda()
    pk = FC.MbPack(groups_of(blk.conv), blk.use_res_connect, dtype)
    outs = {}
    for variant, code in (("flow", 1), ("tiled", -1)):
        outs[variant] = FC.mbconv_native(x.cuda(), pk, variant=code)
        name = N.last_kernel()
        assert ("mbflow" in name) == (variant == "flow"), name
        _check(outs[variant], y, dtype, "fp16 block %d->%d s%d (%s)" % (cin, cout, stride, name), floor=1.0)
    # the two kernels run the same arithmetic in the same order: they agree far inside the tolerance against torch
    assert float((outs["flow"].float() - outs["tiled"].float()).abs().max()) <= 2e-2 * max(1.0, float(y.abs().max()))


SPLIT = [  # cin, cout, stride, h, w, n: the block shapes ssdk_mbsplit.hip is instantiated for (hidden = 6 * cin)
    (24, 32, 2, 70, 45, 3),    # three waves x 3 chunks, merged stride-2 strips, ragged strips / segments
    (32, 32, 1, 31, 47, 3),    # four waves x 3 chunks, residual, two output fragments per strip
    (32, 32, 1, 64, 64, 5),    # the bench's map: three strip pairs, full segments
    (32, 64, 2, 36, 41, 4),    # stride 2 with four output fragments
]


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,stride,h,w,n", SPLIT)
def test_split_register_flow_block(cin, cout, stride, h, w, n, dtype_name):
    """ssdk_mbsplit.hip (hidden channels of a strip pair split over the waves of a workgroup, partial projections
    exchanged through LDS) against the torch fp32 block with 16-bit-rounded intermediates, against the LDS-tiled kernel,
    and bit-reproducible from run to run (the exchange adds the partial sums in wave order)."""
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import InvertedResidual

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin * 11 + cout + stride + h)
    blk = InvertedResidual(cin, cout, stride, 6).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    with torch.no_grad():
        y = x.float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(dtype).float()
        y = mods[1](y).to(dtype).float()
        y = mods[3](mods[2](y))
        if blk.use_res_connect:
            y = y.to(dtype).float() + x.float()
    blk = blk.cuda()
    pk = FC.MbPack(groups_of(blk.conv), blk.use_res_connect, dtype)
    got = FC.mbconv_native(x.cuda(), pk, variant=2)
    name = N.last_kernel()
    assert "mbsplit" in name, name
    _check(got, y, dtype, "split block %d->%d s%d" % (cin, cout, stride), floor=1.0)
    again = FC.mbconv_native(x.cuda(), pk, variant=2)
    assert torch.equal(got, again), "the exchange is not bit-reproducible"
    tiled = FC.mbconv_native(x.cuda(), pk, variant=-1)
    assert "mbsplit" not in N.last_kernel() and "mbflow" not in N.last_kernel()
    assert float((got.float() - tiled.float()).abs().max()) <= 2e-2 * max(1.0, float(y.abs().max()))


def test_flow_row_pairs_equal_single_rows_bit_for_bit(tmp_path):
    """mbflow_kernel<..., PAIR> (two input rows per visit of the chunks: every weight read serves two rows) performs the same
    operations per element in the same order as the single-row instance: the outputs must be EQUAL.  The instance is chosen
    from the environment once per process, so the two run in subprocesses (odd height: the last pair is half empty; several
    segments; residual)."""
    import subprocess
    import sys

    script = r"""
import os, sys, torch
sys.path[:0] = [%r, %r]
from ssds import _native as N
from ssds.modeling.layers import fused_conv as FC
from ssds.modeling.layers.planner import groups_of
from ssds.modeling.nets.mobilenet import InvertedResidual
torch.manual_seed(7)
blk = InvertedResidual(24, 24, 1, 6).eval()
for m in blk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
x = torch.randn(3, 24, 45, 37).to(torch.bfloat16).cuda()
pk = FC.MbPack(groups_of(blk.cuda().conv), blk.use_res_connect, torch.bfloat16)
y = FC.mbconv_native(x, pk, variant=1)
assert "mbflow" in N.last_kernel(), N.last_kernel()
torch.save(y.float().cpu(), sys.argv[1])
""" % (ROOT, os.path.join(ROOT, "ssds.pytorch_amd"))
    outs = []
    for v in ("0", "1"):
        f = str(tmp_path / ("y%s.pt" % v))
        env = dict(os.environ, SSDK_FLOW_PAIR=v)
        r = subprocess.run([sys.executable, "-c", script, f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(f)
    import torch

    a, b = torch.load(outs[0]), torch.load(outs[1])
    assert a.abs().max() > 0.1
    assert torch.equal(a, b)


MBK = [
    # cin, cout, stride, h, w, n -- the blocks of MobileNetV2 from the 32x32 maps down (SSD-MobileNetV2@512, blocks 8-17)
    (160, 160, 1, 16, 16, 3), (160, 320, 1, 16, 16, 2),   # 16x16 maps: one strip; 320 columns = two halves
    (160, 160, 1, 5, 16, 2), (160, 320, 1, 1, 16, 3),     # an odd number of rows (a pair with one output row), a single row
    (96, 160, 2, 32, 32, 2), (96, 160, 2, 10, 32, 3),     # stride 2, 32 -> 16 columns (even / odd column fragments); Ho = 5
    (64, 64, 1, 32, 32, 2), (64, 96, 1, 6, 32, 3),        # 32-wide maps: two strips per row, cross-strip taps
    (96, 96, 1, 32, 32, 2), (96, 96, 1, 3, 32, 2),
    (32, 64, 2, 64, 64, 2), (32, 64, 2, 6, 64, 3),        # stride 2 with two output strips (four input fragments per row)
    (32, 32, 1, 64, 64, 2), (32, 32, 1, 5, 64, 2),        # 64-wide maps: four strips per row
]


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,stride,h,w,n", MBK)
def test_row_pair_block_kernel(cin, cout, stride, h, w, n, dtype_name):
    """ssdk_mbk.hip (16- and 32-pixel-wide maps: a work item is a pair of output rows, the hidden channels are split over the
    waves, every wave streams its own weights from the fragment-major image into MFMA operands; mobilenet.py:56, 84-89)
    against the torch fp32 block with 16-bit-rounded intermediates, against the LDS-tiled kernel, and bit-reproducible from
    run to run."""
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import InvertedResidual

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(cin * 7 + cout + h + stride)
    blk = InvertedResidual(cin, cout, stride, 6).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = (m.weight.data * 2).to(dtype).float()
    x = torch.randn(n, cin, h, w).to(dtype)
    with torch.no_grad():
        y = x.float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(dtype).float()
        y = mods[1](y).to(dtype).float()
        y = mods[3](mods[2](y))
        if blk.use_res_connect:
            y = y.to(dtype).float() + x.float()
    blk = blk.cuda()
    pk = FC.MbPack(groups_of(blk.conv), blk.use_res_connect, dtype)
    im = pk.image(w)
    assert im is not None and im[0] == 4
    got = FC.mbconv_native(x.cuda(), pk, variant=3)
    name = N.last_kernel()
    assert "mbk" in name, name
    _check(got, y, dtype, "row-pair block %d->%d s%d %dx%d" % (cin, cout, stride, h, w), floor=1.0)
    again = FC.mbconv_native(x.cuda(), pk, variant=3)
    assert torch.equal(got, again), "the exchange is not bit-reproducible"
    tiled = FC.mbconv_native(x.cuda(), pk, variant=-1)
    assert "mbk" not in N.last_kernel()
    assert float((got.float() - tiled.float()).abs().max()) <= 2e-2 * max(1.0, float(y.abs().max()))
    # a map of another width never reaches the kernel
    x8 = torch.randn(n, cin, 8, 8).to(dtype).cuda()
    FC.mbconv_native(x8, pk, variant=0)
    assert "mbk" not in N.last_kernel()


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_fused_blocks_16x16_tiles(dtype_name):
    """Stride-1 blocks on maps large enough for >= 512 tiles run on the 16x16-tile instantiations (strip-tiled
    depthwise, batched expand fragments): a residual block and the stem block, ragged sizes included."""
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers.planner import groups_of
    from ssds.modeling.nets.mobilenet import ConvBNReLU6, InvertedResidual

    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    torch.manual_seed(21)

    def randomize(mods):
        for m in mods:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
            if isinstance(m, torch.nn.Conv2d):
                m.weight.data = (m.weight.data * 2).to(dtype).float()

    # residual block 24 -> 144 -> 24 on 10 images of 120 x 104 (8 x 7 tiles of 16 x 16, ragged right/bottom edges)
    blk = InvertedResidual(24, 24, 1, 6).eval()
    randomize(blk.modules())
    x = torch.randn(10, 24, 120, 104).to(dtype)
    with torch.no_grad():
        mods = list(blk.conv.children())
        y = mods[0](x.float()).to(torch.float16).float()
        y = mods[1](y).to(torch.float16).float()
        y = mods[3](mods[2](y)).to(dtype).float() + x.float()
    blk = blk.cuda()
    got = FC.mbconv_native(x.cuda(), FC.MbPack(groups_of(blk.conv), True, dtype))
    assert "16x16" in N.last_kernel(), N.last_kernel()
    _check(got, y, dtype, "16x16 residual block", floor=1.0)
    # stem + first block from a 9 x 3 x 250 x 246 image (stem grid 125 x 123)
    stem = ConvBNReLU6(3, 32, stride=2).eval()
    blk = InvertedResidual(32, 16, 1, 1).eval()
    randomize(list(stem.modules()) + list(blk.modules()))
    img = torch.rand(9, 3, 250, 246).to(dtype)
    with torch.no_grad():
        y = stem(img.float()).to(torch.float16).float()
        mods = list(blk.conv.children())
        y = mods[0](y).to(torch.float16).float()
        y = mods[2](mods[1](y))
    stem, blk = stem.cuda(), blk.cuda()
    pk = FC.MbPack(groups_of(blk.conv), False, dtype, stem_group=groups_of(stem)[0])
    got = FC.mbconv_native(img.cuda(), pk)
    assert "16x16" in N.last_kernel(), N.last_kernel()
    _check(got, y, dtype, "16x16 stem block", floor=1.0)


@pytest.mark.parametrize("head,net,outs,depth", [("SSDFPN", "ResNet18", [3, 4, 5], [128, 256, 512]),
                                                 ("SSDBiFPN", "RegNetX002", [2, 3, 4], [56, 152, 368])])
def test_fpn_bifpn_eval_on_device(head, net, outs, depth):
    """BASELINE configs 3 / 5 families: backbone on PyTorch-ROCm, ConvBNReLU blocks and the shared towers on
    the fused MFMA kernel (per-layer calls), NCHW head outputs with the sigmoid fused; vs the fp32 module."""
    import torch
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import fused_conv as FC

    cls = getattr(ssds, head)
    torch.manual_seed(2)
    fl = [outs + ["Conv:S", "Conv:S"], depth + [depth[-1], 256]]
    nets_outputs, extras, hd = cls.add_extras(fl, [9] * 5, 6)
    model = cls(getattr(nets, net)(outputs=nets_outputs), extras, hd, 6).eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05)
            m.running_var.uniform_(0.9, 1.1)
    x = torch.rand(2, 3, 128, 160)
    with torch.no_grad():
        rl, rc = model(x)
    model = model.cuda().to(torch.bfloat16)
    before, plans = FC.STATS["native_layers"], FC.STATS["plan_runs"]
    with torch.no_grad():
        loc, conf = model(x.cuda().to(torch.bfloat16))
    assert FC.STATS["native_layers"] - before >= 5 * 10, "towers did not run on the fused kernels"
    assert FC.STATS["plan_runs"] == plans + 1, "neck + towers did not run as one recorded plan"
    for l, a, c, b in zip(loc, rl, conf, rc):
        assert l.shape == a.shape and c.shape == b.shape and l.is_contiguous() and c.is_contiguous()
        assert float((c.float().cpu() - b).abs().max()) < 3e-3
        assert float((l.float().cpu() - a).abs().max()) < 0.05 * max(float(a.abs().max()), 1e-2) + 5e-3
    # the small heads run on the executor's side stream: replays must be bit-identical (a buffer re-used by the main
    # lane while a side-lane head still reads it shows up here)
    for _ in range(8):
        with torch.no_grad():
            loc2, conf2 = model(x.cuda().to(torch.bfloat16))
        for l, l2, c, c2 in zip(loc, loc2, conf, conf2):
            assert torch.equal(l, l2) and torch.equal(c, c2)


def test_hipgraph_captured_inference_matches_eager():
    """forward + decode + NMS captured once as a hipGraph (BASELINE config 5 mode); replays on new inputs equal
    the eager pipeline bit for bit."""
    import torch
    from collections import OrderedDict
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import box
    from ssds.modeling.layers.decoder import Decoder
    from ssds.utils.graph import GraphedInference

    torch.manual_seed(6)
    fl = [[5, 7, "Conv:S"], [96, 320, 256]]
    nets_outputs, extras, hd = ssds.SSD.add_extras(fl, [6, 6, 6], 7)
    model = ssds.SSD(nets.MobileNetV2(outputs=nets_outputs), extras, hd, 7).eval().cuda().to(torch.bfloat16)
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in (16, 32, 64))
    dec = Decoder(0.005, 0.6, 50, 100, True, True)
    x0 = torch.rand(4, 3, 128, 128, device="cuda").to(torch.bfloat16)
    g = GraphedInference(model, dec, anchors, x0)
    for seed in (1, 2):
        torch.manual_seed(seed)
        x = torch.rand(4, 3, 128, 128, device="cuda").to(torch.bfloat16)
        got = [t.clone() for t in g(x)]
        with torch.no_grad():
            want = dec(*model(x), anchors)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_tail_stream_decoder_under_forward_load():
    """Decoder.enable_tail_stream() in a serving loop: level + NMS kernels of batch i run on the tail stream while
    the plan of batch i+1 executes on the main stream.  540 batches must equal the in-line detections bit for bit."""
    import torch
    from collections import OrderedDict
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import box
    from ssds.modeling.layers.decoder import Decoder

    torch.manual_seed(8)
    fl = [[5, 7, "Conv:S", "Conv:S", "Conv:S"], [96, 320, 256, 128, 128]]
    nets_outputs, extras, hd = ssds.SSD.add_extras(fl, [6] * 5, 7)
    model = ssds.SSD(nets.MobileNetV2(outputs=nets_outputs), extras, hd, 7).eval().cuda().to(torch.bfloat16)
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in (16, 32, 64, 128, 256))
    dec = Decoder(0.005, 0.6, 50, 100, True, True)
    xs = [torch.rand(16, 3, 256, 256, device="cuda").to(torch.bfloat16) for _ in range(9)]
    with torch.no_grad():
        want = [tuple(t.clone() for t in dec(*model(x), anchors)) for x in xs]
        piped = Decoder(0.005, 0.6, 50, 100, True, True).enable_tail_stream()
        for rep in range(60):
            got = [piped(*model(x), anchors) for x in xs]
            piped.wait()
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                for a, b in zip(g, w):
                    assert torch.equal(a, b)


def test_soak_three_queues_against_in_line_with_poisoned_lds():
    """>= 5000 batches through main lane + side lane + tail stream, LDS poisoned in front of every
    kernel (SSDK_LDS_POISON=1 is read once per process, hence the subprocess), every batch equal to the in-line path."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSDK_LDS_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_multiqueue.py"), "--batches", "5000"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["batches"] >= 5000 and rec["mismatching_batches"] == 0 and rec["lds_poison"] == "1", rec


def test_grouped_small_map_launch_respects_buffer_reuse():
    """The executor puts neighbouring small-map layers that are independent into ONE launch.  On an FPN level of 8x8 at
    batch 96 (6144 output pixels: the heads stay on lane 0) the loc head reads the last tower buffer, that buffer is released
    and the next op -- the conf tower's first layer, which reads the level input -- may be given it as its OUTPUT: head and
    layer are not independent (write-after-read), whichever buffer the arena picks.  The plan's outputs must equal the same
    layers launched one by one, and the plan must have grouped something."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC
    from ssds.modeling.layers import planner

    torch.manual_seed(5)
    dtype = torch.float16

    def pack(cout, act, bn=True):
        conv = nn.Conv2d(256, cout, 3, 1, 1, bias=not bn)
        conv.weight.data.mul_(0.5)
        b = nn.BatchNorm2d(cout) if bn else None
        if bn:
            b.running_mean.normal_(0, 0.1)
            b.running_var.uniform_(0.5, 1.5)
            b.eval()
        return FC.ConvPack(conv.cuda(), b.cuda() if bn else None, act, dtype)

    loc = ([pack(256, "relu") for _ in range(4)], pack(36, "none", bn=False))
    conf = ([pack(256, "relu") for _ in range(4)], pack(180, "none", bn=False))
    x = torch.randn(96, 256, 8, 8).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    plan = FC.ConvPlan(x.device, dtype)
    xx = plan.add_input(x.shape)
    planner._record_towers(plan, xx, loc, conf)
    plan.finalize()
    assert all(not L.get("lane") for L in plan.layers), "the heads of this level are expected in line"
    (l_plan,), (c_plan,) = plan.run(x)
    (l_again,), (c_again,) = plan.run(x)
    plan.ctx.set_op_profiling(True)
    plan.run(x)
    torch.cuda.synchronize()
    kernels = [k for k, _ in plan.ctx.op_timings()]
    plan.ctx.set_op_profiling(False)
    assert any("group" in k for k in kernels), kernels

    def chain(body_final, act):
        body, final = body_final
        cur = x
        for pk in body:
            cur = FC.conv_native(cur, pk)
            assert N.last_kernel() == "conv_smallmap_kernel", N.last_kernel()
        return FC.conv_native(cur, final, act=act, nchw_out=True)

    l_ref, c_ref = chain(loc, "none"), chain(conf, "sigmoid")
    assert torch.equal(l_plan, l_ref) and torch.equal(c_plan, c_ref)
    assert torch.equal(l_again, l_ref) and torch.equal(c_again, c_ref)

