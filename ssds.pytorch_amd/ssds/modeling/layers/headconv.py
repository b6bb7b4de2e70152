"""The 3x3 loc | conf convolutions of an SSD level (reference ssds/modeling/ssds/ssd.py:100-103, called at :67-70) inside the
TRAINING step: forward on the inference kernels (one split-output GEMM per level: csrc/ssdk_conv3x3s.hip, ssdk_conv3x3.hip,
ssdk_smallmap.hip -- the kernels the eval plan uses); the INPUT gradient of the pair is one more such convolution (of the
concatenated output gradient with the transposed, flipped weights); the weight / bias gradients stay on the framework's
convolution backward, per module as before.  Rounds 2-6 left these twelve
convolutions to MIOpen: ~1 ms of implicit-GEMM forward kernels + layout transposes per step at SSD-MobileNetV2@512, batch 64
(profiles/r06_train_kernel_split_final_v1.txt).

The weights change every step, so the kernels' layouts (KRSC 16-bit rows, the fragment-major image, fp32 biases) are rebuilt per
call from the fp32 master tensors by ONE launch (ssdk_pack_conv3x3).  HIP tensors in a 16-bit autocast dtype only; everything else
goes through the modules' own forward."""
import os

import torch

from ssds import _native as N
from ssds.modeling.layers import fused_conv as FC


def enabled():
    return os.environ.get("SSDK_HEAD_PAIR", "1") != "0"


def supported(x, loc, conf):
    """x: the level's feature map as the heads see it; loc / conf: the two nn.Conv2d."""
    if not (x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()):
        return False
    dt = torch.get_autocast_dtype("cuda")
    if dt not in (torch.bfloat16, torch.float16):
        return False
    for m in (loc, conf):
        if not (type(m) is torch.nn.Conv2d and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros" and m.weight.dtype == torch.float32
                and m.weight.is_cuda):
            return False
    return loc.in_channels == conf.in_channels == int(x.shape[1]) and loc.in_channels % 8 == 0 and (loc.bias is None) == (conf.bias is None)


class _Pack(object):
    """The part of fused_conv.ConvPack that fill_desc / wants_frag read."""

    __slots__ = ("kind", "w", "scale", "bias", "cin", "cout", "k", "stride", "groups", "act", "_img")

    def frag(self):
        return self._img


class _HeadPair3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wl, bl, wc, bc):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        if x.dtype != dt:
            x = x.to(dt)
        n, cin, h, w = (int(v) for v in x.shape)
        nl, nc = int(wl.shape[0]), int(wc.shape[0])
        rows, kel = nl + nc, 9 * cin
        dev = x.device
        xcl = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        krsc = torch.empty((rows, 3, 3, cin), device=dev, dtype=dt)
        bias = torch.empty(rows, device=dev, dtype=torch.float32)
        pk = _Pack()
        pk.kind, pk.w, pk.scale, pk.bias, pk.cin, pk.cout, pk.k, pk.stride, pk.groups, pk.act = "dense", krsc, None, bias, cin, rows, 3, 1, 1, "none"
        pk._img = None
        want_img = FC.USE_WFRAG and kel % 32 == 0 and cin % 32 == 0 and FC.wants_frag(pk, h, w, False)
        img = torch.empty(int(N.lib.ssdk_weight_frag_bytes(rows, kel)) // 2, device=dev, dtype=dt) if want_img else None
        pk._img = img
        y = torch.empty((n, nl, h, w), device=dev, dtype=dt)
        y2 = torch.empty((n, nc, h, w), device=dev, dtype=dt)
        with torch.cuda.device(dev):
            sp = N.stream_ptr(dev)
            N.check(N.lib.ssdk_pack_conv3x3(wl.data_ptr(), None if bl is None else bl.data_ptr(), nl, wc.data_ptr(),
                                            None if bc is None else bc.data_ptr(), nc, cin, krsc.data_ptr(),
                                            None if img is None else img.data_ptr(), bias.data_ptr(), N.dtype_code(x), sp), "pack_conv3x3")
            d = FC.fill_desc(N.ConvDesc(), xcl.data_ptr(), n, h, w, pk, N.dtype_code(x), "none", y.data_ptr(), N.NHWC, N.NCHW, None,
                             y2.data_ptr(), nl, "none")
            need = int(N.lib.ssdk_conv_workspace_bytes(n, cin, h, w, rows, 3, 1, N.dtype_code(x)))
            if need:
                ws = FC._splitk_ws(dev, need)
                wptr = (ws.data_ptr() + 255) & ~255
                rc = N.lib.ssdk_conv(ctypes_byref(d), wptr, ws.numel() - (wptr - ws.data_ptr()), sp)
            else:
                rc = N.lib.ssdk_conv(ctypes_byref(d), None, 0, sp)
            N.check(rc, "conv (head pair)")
        ctx.save_for_backward(x, wl, wc)
        ctx.has_bias = bl is not None
        return y, y2

    @staticmethod
    def backward(ctx, gl, gc):
        x, wl, wc = ctx.saved_tensors
        dt = x.dtype
        want_gx = ctx.needs_input_grad[0]
        native_gx = want_gx and dgrad_enabled()
        # weight / bias gradients: im2col + ssdk_pw_wgrad per module (tools/head_wgrad_probe.py: 198 / 146 / 160 us against the
        # library's 369 / 208 / 217 on the 32^2 / 16^2 / 8^2 levels; the smaller levels with the batch folded into the pixel
        # dimension, pointwise.FOLD_BELOW -- per image they were slower than the library).  Library (WGRAD_MIN_PIXELS):
        # two calls, as autograd would make them for the two modules (ONE call on the concatenated 504 channels was measured and is
        # 2 ms per step slower: the library picks k-tile-8 kernels for it; tools/run/r06_s23.sh)
        n, cin, h, w_ = (int(v) for v in x.shape)
        native_gw = wgrad_enabled() and h * w_ >= WGRAD_MIN_PIXELS and x.is_contiguous()
        if native_gw:
            (gwl, gbl), (gwc, gbc) = _weight_gradients(x, ((gl, wl), (gc, wc)), ctx.has_bias)
            gx1 = gx2 = None
            if want_gx and not native_gx:
                gx1 = sum(torch.ops.aten.convolution_backward(g.to(dt), x, w.to(dt), [int(w.shape[0])], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [True, False, False])[0] for g, w in ((gl, wl), (gc, wc)))
                gx2 = 0
        else:
            out = []
            for g, w in ((gl, wl), (gc, wc)):
                out.append(torch.ops.aten.convolution_backward(g.to(dt), x, w.to(dt), [int(w.shape[0])], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                               [want_gx and not native_gx, True, ctx.has_bias]))
            (gx1, gwl, gbl), (gx2, gwc, gbc) = out
        gx = None
        if native_gx:
            gx = _input_gradient(gl, gc, wl, wc, x)
        elif want_gx:
            gx = gx1 + gx2
        if not ctx.has_bias:
            gbl = gbc = None
        else:
            gbl, gbc = gbl.to(wl.dtype), gbc.to(wc.dtype)
        return gx, gwl.to(wl.dtype), gbl, gwc.to(wc.dtype), gbc


WGRAD_MIN_PIXELS = 0  # (64 -- ssds/utils/train_ddp.py with SSDK_CONV3_NATIVE=0 --: the levels below 64 pixels on the library's weight gradient)


def wgrad_enabled():
    return True  # (round 6: the SSDK_HEAD_PAIR_WGRAD switch is gone, its A/B is settled -- tools/run/r06_s29.sh: 17.7 vs 17.9 ms per step)


def _weight_gradients(x, pairs, has_bias):
    """[(dweight, dbias)] of 3x3 / stride 1 / pad 1 convolutions that share the input x [N, Cin, H, W] (contiguous, 16 bit): ONE
    im2col of x, then ssdk_pw_wgrad per (dy, weight) pair -- dW = dy col^T contracts over pixels (csrc/ssdk_pwtrain.hip)."""
    from ssds.modeling.layers import pointwise as PW

    n, cin, h, w = (int(v) for v in x.shape)
    dev, dt, hw = x.device, x.dtype, h * w
    out = []
    with torch.cuda.device(dev):
        sp = N.stream_ptr(dev)
        fold = n > 1 and hw < PW.FOLD_BELOW and (n * hw) % 8 == 0  # small levels: the batch as ONE image of n * hw pixels (PW._fold)
        col = PW._im2col(x.detach(), 1, fold)
        kp = int(col.shape[1])
        gn, ghw = (1, n * hw) if fold else (n, hw)
        for g, wt in pairs:
            g = g.to(dt).contiguous()
            cout = int(wt.shape[0])
            g2 = PW._fold(g.view(n, cout, hw)) if fold else g
            need = int(N.lib.ssdk_pw_wgrad_workspace_bytes(gn, cout, kp, ghw))
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            gw32 = torch.empty((cout, kp), device=dev, dtype=torch.float32)
            N.check(N.lib.ssdk_pw_wgrad(g2.data_ptr(), col.data_ptr(), gw32.data_ptr(), ws.data_ptr(), need, gn, cout, kp, ghw, N.dtype_code(x), sp),
                    "pw_wgrad (head pair)")
            gw = gw32[:, : cin * 9].reshape(cout, cin, 3, 3)
            gb = g.sum((0, 2, 3), dtype=torch.float32) if has_bias else None
            out.append((gw, gb))
    PW.release_col_cache()
    return out


def dgrad_enabled():
    return True  # (round 6: the SSDK_HEAD_PAIR_DGRAD switch is gone, its A/B is settled -- tools/run/r06_s25.sh: 17.8 vs 18.4 ms per step)


_ZERO_BIAS = {}


def _input_gradient(gl, gc, wl, wc, x):
    """dx of the pair as ONE 3x3 convolution of the concatenated output gradient with the transposed, flipped weights
    (ssdk_pack_conv3x3_dgrad) on the inference kernels: dy [N, opad, H, W] in channels_last memory (the 504 channels zero-padded to
    512: conv_smallmap_kernel has instances for 128 / 256 / 512 input channels), dx [N, Cin, H, W] contiguous."""
    dt = x.dtype
    n, cin, h, w = (int(v) for v in x.shape)
    nl, nc = int(wl.shape[0]), int(wc.shape[0])
    rows = nl + nc
    opad = next((c for c in (128, 256, 512) if c >= rows), (rows + 31) // 32 * 32)
    dev = x.device
    g = torch.empty((n, opad, h, w), device=dev, dtype=dt, memory_format=torch.channels_last)
    gl, gc = gl.to(dt).contiguous(), gc.to(dt).contiguous()
    with torch.cuda.device(dev):
        N.check(N.lib.ssdk_concat_nchw_to_nhwc(gl.data_ptr(), nl, gc.data_ptr(), nc, g.data_ptr(), opad, n, h * w, N.dtype_code(x),
                                               N.stream_ptr(dev)), "concat_nchw_to_nhwc")
    krsc = torch.empty((cin, 3, 3, opad), device=dev, dtype=dt)
    zb = _ZERO_BIAS.get((dev.index, cin))
    if zb is None:
        zb = _ZERO_BIAS[(dev.index, cin)] = torch.zeros(cin, device=dev, dtype=torch.float32)
    pk = _Pack()
    pk.kind, pk.w, pk.scale, pk.bias, pk.cin, pk.cout, pk.k, pk.stride, pk.groups, pk.act = "dense", krsc, None, zb, opad, cin, 3, 1, 1, "none"
    pk._img = None
    want_img = FC.USE_WFRAG and opad % 32 == 0 and FC.wants_frag(pk, h, w, False)
    img = torch.empty(int(N.lib.ssdk_weight_frag_bytes(cin, 9 * opad)) // 2, device=dev, dtype=dt) if want_img else None
    pk._img = img
    gx = torch.empty((n, cin, h, w), device=dev, dtype=dt)
    with torch.cuda.device(dev):
        sp = N.stream_ptr(dev)
        N.check(N.lib.ssdk_pack_conv3x3_dgrad(wl.data_ptr(), nl, wc.data_ptr(), nc, cin, opad, krsc.data_ptr(),
                                              None if img is None else img.data_ptr(), N.dtype_code(x), sp), "pack_conv3x3_dgrad")
        d = FC.fill_desc(N.ConvDesc(), g.data_ptr(), n, h, w, pk, N.dtype_code(x), "none", gx.data_ptr(), N.NHWC, N.NCHW)
        need = int(N.lib.ssdk_conv_workspace_bytes(n, opad, h, w, cin, 3, 1, N.dtype_code(x)))
        if need:
            ws = FC._splitk_ws(dev, need)
            wptr = (ws.data_ptr() + 255) & ~255
            rc = N.lib.ssdk_conv(ctypes_byref(d), wptr, ws.numel() - (wptr - ws.data_ptr()), sp)
        else:
            rc = N.lib.ssdk_conv(ctypes_byref(d), None, 0, sp)
        N.check(rc, "conv (head pair, input gradient)")
    return gx


def ctypes_byref(d):
    import ctypes

    return ctypes.byref(d)


def head_pair(x, loc, conf):
    """(loc(x), conf(x)) for the two 3x3 convolutions of one level; differentiable in x and the four parameters."""
    with torch.autocast("cuda", enabled=False):
        return _HeadPair3x3.apply(x, loc.weight, loc.bias, conf.weight, conf.bias)


def use_head_pairs(model):
    """Mark every SSD head of ``model`` (modules with ``loc`` / ``conf`` lists of bare 3x3 convolutions, ssd.py) so that its
    TRAINING forward runs each level's loc | conf pair through ``head_pair``.  SSDK_HEAD_PAIR=0: no effect.  -> heads marked."""
    n = 0
    if not enabled():
        return n
    for m in model.modules():
        if type(m).__name__ == "SSD" and hasattr(m, "loc") and hasattr(m, "conf"):
            m.__dict__["_ssdk_head_pair"] = True
            n += 1
    return n
