"""Host side of the fused conv + folded-BN + activation kernels (C-ABI ``ssdk_conv`` /
``ssdk_conv_sequence``, csrc/ssdk_conv.hip).

* ``pack_conv(conv, bn, dtype)`` folds the BatchNorm running statistics into a per-channel fp32
  (scale, bias) pair and re-packs the weight once into the KRSC layout the kernels read.
* ``conv_native(x, pack, ...)`` launches one layer (activations are NHWC = torch ``channels_last``).
* ``ConvPlan`` records a whole network as an array of descriptors with pre-assigned activation buffers and
  replays it with ONE host call per forward (``ssdk_conv_sequence``): ~2 us of host time per layer instead
  of a Python round trip per layer, and trivially hipGraph-capturable.
* ``FusedSequentialMixin`` lets the reference-shaped nn.Sequential blocks (basic_layers.py) use the fused
  kernels in eval mode without changing their state_dict layout.

Switch: ``SSDK_FUSED_CONV`` = "1" (default: HIP kernels in eval mode on a HIP device) or "0" (torch/MIOpen
everywhere; for A/B measurements only, never chosen silently).
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ssds import _native as N

_ACT_OF = {nn.ReLU: "relu", nn.ReLU6: "relu6", nn.SiLU: "silu", nn.Sigmoid: "sigmoid"}


def _act_name(m):
    """Activation name of module ``m`` or None.  Looked up along the MRO: the training Solver swaps the classes of
    activations behind a kernel-backed BatchNorm to subclasses (batchnorm.FusedAwayReLU6 / FusedAwayReLU), and a model
    that went through it must still record its eval plan."""
    for klass in type(m).__mro__:
        if klass in _ACT_OF:
            return _ACT_OF[klass]
    return None

STATS = {"native_layers": 0, "plan_runs": 0, "torch_fallback_layers": 0}


def fused_enabled():
    return os.environ.get("SSDK_FUSED_CONV", "1") != "0"


def fold_bn(conv, bn):
    """-> (scale[Cout], bias[Cout]) fp32 such that bn(conv(x)) == conv_nobias(x) * scale + bias."""
    cout = conv.out_channels
    dev = conv.weight.device
    if bn is None:
        scale = torch.ones(cout, device=dev, dtype=torch.float32)
        bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout, device=dev)
        return scale, bias.contiguous()
    var = bn.running_var.detach().float()
    mean = bn.running_mean.detach().float()
    gamma = bn.weight.detach().float() if bn.affine else torch.ones_like(var)
    beta = bn.bias.detach().float() if bn.affine else torch.zeros_like(var)
    scale = gamma / torch.sqrt(var + bn.eps)
    b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(mean)
    bias = (b0 - mean) * scale + beta
    return scale.contiguous(), bias.contiguous()


def conv_kind(conv):
    """'dense' | 'dw' | 'g16' | 'stem' | None (None: not covered by the HIP kernels -> torch, reported)."""
    k = conv.kernel_size
    if not (k[0] == k[1] and k[0] in (1, 3) and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.padding == (k[0] // 2, k[0] // 2) and conv.dilation == (1, 1)
            and conv.padding_mode == "zeros"):
        return None
    if conv.groups == 1:
        if conv.in_channels <= 4:
            return "stem" if (k[0] == 3 and conv.out_channels in (16, 32, 64)) else None
        return "dense" if conv.in_channels % 8 == 0 else None
    if conv.groups == conv.in_channels == conv.out_channels and k[0] == 3 and conv.in_channels % 8 == 0:
        return "dw"
    if conv.in_channels == conv.out_channels == conv.groups * 16 and k[0] == 3:
        return "g16"  # 16 channels per group (RegNetX bottlenecks): csrc/ssdk_gconv.hip
    return None


# Fragment-major weight images for the kernels that stream weights into operand registers (ConvPack.frag); False keeps
# those kernels on the KRSC tensor (A/B runs and the test of that path).
USE_WFRAG = os.environ.get("SSDK_WFRAG", "1") != "0"


class ConvPack(object):
    """Weights of one fused layer in kernel layout (built once per model/dtype)."""

    __slots__ = ("kind", "w", "scale", "bias", "cin", "cout", "k", "stride", "groups", "act", "_frag")

    def frag(self):
        """Fragment-major image of the dense weights (include/ssdk.h ``ssdk_weight_frag_bytes``) for the kernels that stream
        weights straight into MFMA operand registers (conv_smallmap, xpair); None where no such kernel applies.  Built once
        per weight tensor (pure layout: [rows][K] -> [rows/16][K/32][4][16][8], rows zero-padded to a multiple of 16)."""
        k_elems = self.k * self.k * self.cin
        if (not USE_WFRAG or self.kind != "dense" or self.w.dtype not in (torch.bfloat16, torch.float16) or k_elems % 32
                or self.cin % 32):
            return None
        cached = getattr(self, "_frag", None)
        if cached is not None and cached[0] == (self.w.data_ptr(), self.w._version):
            return cached[1]
        rows = self.cout
        g = (rows + 15) // 16
        w2d = self.w.reshape(rows, k_elems)
        if g * 16 != rows:
            w2d = torch.cat([w2d, w2d.new_zeros((g * 16 - rows, k_elems))], 0)
        img = w2d.view(g, 16, k_elems // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
        assert img.numel() * 2 == N.lib.ssdk_weight_frag_bytes(rows, k_elems)
        self._frag = ((self.w.data_ptr(), self.w._version), img)
        return img

    def __init__(self, conv, bn, act, dtype, extra_cout=None):
        self._frag = None
        self.kind = conv_kind(conv)
        if self.kind is None:
            raise N.SsdkError("conv {} is not covered by the HIP kernels".format(conv))
        scale, bias = fold_bn(conv, bn)
        w = conv.weight.detach().float()
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.k, self.stride, self.groups, self.act = conv.kernel_size[0], conv.stride[0], conv.groups, act
        if self.kind == "stem":  # fp32 weights with the BN scale folded in, KRSC
            self.w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous()
            self.scale = None
        elif self.kind == "dw":  # [C,1,3,3] -> [3][3][C]
            self.w = w[:, 0].permute(1, 2, 0).contiguous().to(dtype)
            self.scale = scale
        else:  # [Cout,Cin,kh,kw] -> [Cout][kh][kw][Cin]
            self.w = w.permute(0, 2, 3, 1).contiguous().to(dtype)
            self.scale = scale if bn is not None else None
        self.bias = bias


class StemPack(object):
    """7x7 / stride 2 / pad 3 stem conv on a 3-channel image + BN + activation (ResNet conv1/bn1/relu) packed for
    csrc/ssdk_stem.hip: weights [Cout][ky 7][kx padded to 8][ci padded to 4], zeros in the padding slots."""

    __slots__ = ("w", "scale", "bias", "cin", "cout", "act")

    @staticmethod
    def supported(conv, bn):
        return (bn is not None and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
                and conv.groups == 1 and conv.in_channels == 3 and conv.out_channels in (32, 64))

    def __init__(self, conv, bn, act, dtype):
        self.scale, self.bias = fold_bn(conv, bn)
        w = conv.weight.detach().float().permute(0, 2, 3, 1)  # [Cout][ky][kx][ci]
        wp = torch.zeros((conv.out_channels, 7, 8, 4), device=w.device, dtype=torch.float32)
        wp[:, :, :7, :3] = w
        self.w = wp.to(dtype).contiguous()
        self.cin, self.cout, self.act = 3, conv.out_channels, act


def stem7_native(x, pack):
    """[N,3,H,W] image (NCHW contiguous or channels_last) -> channels_last [N,Cout,H/2,W/2]."""
    N.require_device(x, "conv_stem7")
    n, c, h, w = (int(v) for v in x.shape)
    layout = N.NCHW
    if not x.is_contiguous():
        x = x.contiguous(memory_format=torch.channels_last)
        layout = N.NHWC
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    y = torch.empty((n, pack.cout, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    d = N.StemDesc()
    d.x, d.w, d.scale, d.bias, d.y = x.data_ptr(), pack.w.data_ptr(), pack.scale.data_ptr(), pack.bias.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.act, d.dtype, d.in_layout = n, h, w, c, pack.cout, N.ACT[pack.act], N.dtype_code(x), layout
    with torch.cuda.device(x.device):
        rc = N.lib.ssdk_conv_stem7(ctypes.byref(d), N.stream_ptr(x.device))
    N.check(rc, "conv_stem7")
    return y


def maxpool_native(x):
    """F.max_pool2d(x, 3, 2, 1) on a channels_last tensor."""
    N.require_device(x, "maxpool3x3s2")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    n, c, h, w = (int(v) for v in x.shape)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = torch.empty((n, c, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    d = N.PoolDesc()
    d.x, d.y = x.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.C, d.dtype = n, h, w, c, N.dtype_code(x)
    with torch.cuda.device(x.device):
        rc = N.lib.ssdk_maxpool3x3s2(ctypes.byref(d), N.stream_ptr(x.device))
    N.check(rc, "maxpool3x3s2")
    return y


def pack_heads(loc_conv, conf_conv, dtype):
    """loc | conf of one SSD level as ONE GEMM with N = A*(4+C): concatenated KRSC weights + biases."""
    p = ConvPack(loc_conv, None, "none", dtype)
    q = ConvPack(conf_conv, None, "none", dtype)
    p.w = torch.cat([p.w, q.w], 0).contiguous()
    p.bias = torch.cat([p.bias, q.bias], 0).contiguous()
    p.cout = loc_conv.out_channels + conf_conv.out_channels
    return p


class MbPack(object):
    """One MobileNetV2 inverted-residual block (expand 1x1 -> depthwise 3x3 -> project 1x1) packed for the
    fused block kernel (csrc/ssdk_mbconv.hip).  ``groups`` = [(conv, bn, act)] x 3 as produced by
    ``sequential_groups`` on the flattened block."""

    __slots__ = ("e", "d", "p", "we", "wd", "bd", "wp", "cin", "chid", "cout", "stride", "residual", "stem", "_image")

    @staticmethod
    def supported(groups, residual):
        if len(groups) != 3:
            return False
        (ce, _, ae), (cd, _, ad), (cp, _, ap) = groups
        return (conv_kind(ce) == "dense" and ce.kernel_size == (1, 1) and ce.stride == (1, 1) and ae == "relu6"
                and conv_kind(cd) == "dw" and ad == "relu6"
                and conv_kind(cp) == "dense" and cp.kernel_size == (1, 1) and cp.stride == (1, 1) and ap == "none"
                and ce.in_channels <= 160 and cp.out_channels <= 320 and cp.out_channels % 8 == 0
                and ce.out_channels == cd.in_channels == cp.in_channels
                and all(bn is not None for _, bn, _ in groups)
                and (not residual or (cd.stride[0] == 1 and ce.in_channels == cp.out_channels)))

    @staticmethod
    def stem_supported(stem_groups, block_groups):
        """network stem (3x3/s2 conv on <=3 channels, BN, ReLU6) + an expand-free block (dw, pw-linear)."""
        if len(stem_groups) != 1 or len(block_groups) != 2:
            return False
        (cs, bs, a_s), (cd, bdn, ad), (cp, bpn, ap) = stem_groups[0], block_groups[0], block_groups[1]
        return (cs.kernel_size == (3, 3) and cs.stride == (2, 2) and cs.padding == (1, 1) and cs.groups == 1
                and 9 * cs.in_channels <= 32 and cs.out_channels % 8 == 0 and a_s == "relu6" and bs is not None
                and conv_kind(cd) == "dw" and ad == "relu6" and cd.in_channels == cs.out_channels
                and conv_kind(cp) == "dense" and cp.kernel_size == (1, 1) and cp.stride == (1, 1) and ap == "none"
                and cp.out_channels <= 64 and cp.out_channels % 8 == 0 and bdn is not None and bpn is not None)

    def __init__(self, groups, residual, dtype, stem_group=None):
        self.stem = 0
        self._image = None
        if stem_group is not None:  # the stem conv plays the role of the expand conv
            cs, bs, _ = stem_group
            (cd, bd, _), (cp, bp, _) = groups
            scale, bias = fold_bn(cs, bs)
            # K layout of the stem GEMM = (ky, kx padded 3 -> 8, ci padded -> 4): a k-step is one kernel row and a
            # lane's 8 k-values are 2 neighbouring patch pixels x 4 channels (csrc/ssdk_mbconv.hip stem mode)
            w = cs.weight.detach().float().permute(0, 2, 3, 1)  # [Cout][ky][kx][ci]
            wpad = torch.zeros((cs.out_channels, 3, 8, 4), device=w.device, dtype=torch.float32)
            wpad[:, :, :3, : w.shape[3]] = w
            self.e = ConvPack.__new__(ConvPack)
            # the BN scale goes into the weights BEFORE they are rounded (the block kernels then apply a scale of exactly 1:
            # csrc/ssdk_mbflow.hip starts its expand MFMA from the BN bias and has no BN arithmetic left in the row loop)
            wpad = wpad * scale.view(-1, 1, 1, 1)
            self.e.w, self.e.scale, self.e.bias = wpad.reshape(cs.out_channels, 96).to(dtype).contiguous(), torch.ones_like(scale), bias
            self.d = ConvPack(cd, bd, "relu6", dtype)
            self.p = ConvPack(cp, bp, "none", dtype)
            self.cin, self.chid, self.cout = cs.in_channels, cs.out_channels, cp.out_channels
            self.stride, self.residual, self.stem = cd.stride[0], False, 1
            self._fold_tail(cd, bd, cp)
            return
        (ce, be, _), (cd, bd, _), (cp, bp, _) = groups
        self.e = ConvPack(ce, be, "relu6", dtype)
        # expand BN scale folded into the weights before rounding (see the stem case above); KRSC [Chid][1][1][Cin]
        w_e = ce.weight.detach().float().permute(0, 2, 3, 1) * self.e.scale.view(-1, 1, 1, 1)
        self.e.w, self.e.scale = w_e.contiguous().to(dtype), torch.ones_like(self.e.scale)
        self.d = ConvPack(cd, bd, "relu6", dtype)
        self.p = ConvPack(cp, bp, "none", dtype)
        self.cin, self.chid, self.cout = ce.in_channels, ce.out_channels, cp.out_channels
        self.stride, self.residual = cd.stride[0], bool(residual)
        self._fold_tail(cd, bd, cp)

    def _fold_tail(self, cd, bd, cp):
        # internal tensors are fp16: depthwise weights with the BN scale folded in, fp16 bias, fp16 projection
        sd, bdv = fold_bn(cd, bd)
        wdw = cd.weight.detach().float()[:, 0] * sd.view(-1, 1, 1)  # [C,3,3]
        self.wd = wdw.permute(1, 2, 0).contiguous().to(torch.float16)
        self.bd = bdv.to(torch.float16).contiguous()
        self.wp = cp.weight.detach().float().permute(0, 2, 3, 1).contiguous().to(torch.float16)


    def image(self, w_in, nw=None):
        """(nw, tensor) -- the image of the two 1x1 weight matrices and the per-channel constants for csrc/ssdk_mbk.hip
        (layout: include/ssdk.h ``ssdk_mbconv_desc.w_image``) on a map ``w_in`` pixels wide, or None where no instance of that
        kernel takes the block.  Built once per pack and output width: pure permutations of ``e.w`` (activation dtype, BN
        scale folded in), ``wp`` / ``wd`` (fp16) and the folded biases, carried as 16-bit words.  ``nw``: slices of the hidden
        channels = waves per work item (4)."""
        if self.stem or self.stride not in (1, 2):
            return None
        nw = 4 if nw is None else int(nw)
        wo = (w_in + 2 - 3) // self.stride + 1
        if w_in != wo * self.stride:
            return None
        key = (nw, wo)
        if self._image is not None and self._image[0] == key:
            return self._image[1]
        nfo_c = ctypes.c_int(0)
        need = int(N.lib.ssdk_mbk_image_bytes(self.cin, self.chid, self.cout, self.stride, wo, nw, ctypes.byref(nfo_c)))
        if need == 0:
            self._image = (key, None)
            return None
        ks, nch, nfo = self.cin // 32, self.chid // 16, int(nfo_c.value)
        nchw = (nch + nw - 1) // nw
        npair = (nchw + 1) // 2
        halves = self.cout // (16 * nfo)
        dev = self.wp.device
        we = self.e.w.reshape(self.chid, self.cin).view(torch.int16)
        wp = self.wp.reshape(self.cout, self.chid).view(torch.int16)
        we = torch.cat([we, we.new_zeros((1, self.cin))], 0)      # row chid = the zero row of a chunk beyond the slice / Chid
        wp = torch.cat([wp, wp.new_zeros((self.cout, 1))], 1)     # column chid likewise
        ar = lambda n: torch.arange(n, device=dev)  # noqa: E731
        lane = ar(64)
        fr, fg = lane & 15, lane >> 4
        # expand fragments [slice w][pair t][chunk cc][k-step][lane][8]
        w_, t_, cc_, ks_, j_ = ar(nw).view(-1, 1, 1, 1, 1, 1), ar(npair).view(1, -1, 1, 1, 1, 1), ar(2).view(1, 1, -1, 1, 1, 1), \
            ar(ks).view(1, 1, 1, -1, 1, 1), ar(8).view(1, 1, 1, 1, 1, -1)
        loc = 2 * t_ + cc_
        chunk = w_ * nchw + loc
        ok = (loc < nchw) & (chunk < nch)
        row = torch.where(ok, chunk * 16 + fr.view(1, 1, 1, 1, -1, 1), torch.full_like(chunk, self.chid))
        col = (32 * ks_ + 8 * fg.view(1, 1, 1, 1, -1, 1) + j_).expand(nw, npair, 2, ks, 64, 8)
        exp = we[row.expand(nw, npair, 2, ks, 64, 8), col]                                  # [nw, np, 2, ks, 64, 8]
        # projection fragments [half h][slice w][pair t][fragment f][lane][8]
        h_, w2, t2, f_, j2 = ar(halves).view(-1, 1, 1, 1, 1, 1), ar(nw).view(1, -1, 1, 1, 1, 1), ar(npair).view(1, 1, -1, 1, 1, 1), \
            ar(nfo).view(1, 1, 1, -1, 1, 1), ar(8).view(1, 1, 1, 1, 1, -1)
        loc2 = 2 * t2 + j2 // 4
        chunk2 = w2 * nchw + loc2
        ok2 = (loc2 < nchw) & (chunk2 < nch)
        hid = torch.where(ok2, chunk2 * 16 + 4 * fg.view(1, 1, 1, 1, -1, 1) + j2 % 4, torch.full_like(chunk2 + fg.view(1, 1, 1, 1, -1, 1), self.chid))
        co = (h_ * (16 * nfo) + 16 * f_ + fr.view(1, 1, 1, 1, -1, 1)).expand(halves, nw, npair, nfo, 64, 8)
        prj = wp[co, hid.expand(halves, nw, npair, nfo, 64, 8)]                               # [halves, nw, np, nfo, 64, 8]
        wts = torch.cat([exp.reshape(1, nw, npair, 2 * ks * 512).expand(halves, -1, -1, -1),
                         prj.reshape(halves, nw, npair, nfo * 512)], 3).reshape(-1)
        # per-slice constants, laid out as the kernel reads them: bias_expand fp32 | taps fp16 | bias_dw / 6 fp16
        c_, g_, q_ = ar(nchw).view(1, -1, 1, 1), ar(4).view(1, 1, -1, 1), ar(4).view(1, 1, 1, -1)
        gc = ar(nw).view(-1, 1, 1, 1) * nchw + c_
        ch = torch.where(gc < nch, gc * 16 + 4 * g_ + q_, torch.full_like(gc + g_ + q_, self.chid))   # [nw, nchw, 4, 4]
        zf = lambda t: torch.cat([t, t.new_zeros((1,) + tuple(t.shape[1:]))], 0)  # noqa: E731 -- index chid = zeros
        be = zf(self.e.bias.float())[ch]                                                           # fp32 [nw, nchw, 4, 4]
        wd = zf(self.wd.reshape(9, self.chid).t().contiguous())[ch]                                # fp16 [nw, nchw, 4, 4, 9]
        wd = wd.permute(0, 1, 4, 2, 3).contiguous()                                                # [nw, nchw, 9, 4, 4]
        bd6 = zf((self.bd.float() * torch.tensor(1.0 / 6.0, dtype=torch.float32, device=dev)).to(torch.float16))[ch]
        misc_kb = (nchw * 384 + 1023) // 1024
        misc = torch.cat([be.contiguous().view(torch.int16).reshape(nw, -1), wd.view(torch.int16).reshape(nw, -1),
                          bd6.contiguous().view(torch.int16).reshape(nw, -1)], 1)
        misc = torch.cat([misc, misc.new_zeros((nw, misc_kb * 512 - misc.shape[1]))], 1).reshape(-1)
        # projection BN per half: [nfo][4 g][scale * 6 (4 q) | bias (4 q)] fp32
        cof = (ar(halves).view(-1, 1, 1, 1) * (16 * nfo) + 16 * ar(nfo).view(1, -1, 1, 1) + 4 * ar(4).view(1, 1, -1, 1)
               + ar(4).view(1, 1, 1, -1))
        spb = torch.cat([(self.p.scale.float() * 6.0)[cof], self.p.bias.float()[cof]], 3).contiguous()  # [halves, nfo, 4, 8]
        spb = spb.view(torch.int16).reshape(halves, -1)
        spb = torch.cat([spb, spb.new_zeros((halves, 1024 - spb.shape[1]))], 1).reshape(-1)
        img = torch.cat([wts, misc, spb]).contiguous()
        assert img.numel() * 2 == need, (img.numel() * 2, need)
        self._image = (key, (nw, img))
        return self._image[1]

    def fp16_safe(self):
        """The block kernel keeps the expanded tensor, the depthwise weights / bias / output and the projection weights
        in fp16 (11-bit mantissa: more precise than bf16 inside its range, but the range is 65504).  Everything the
        kernel rounds to fp16 is bounded here from the folded weights: the expanded tensor is clamped to [0, 6], so a
        depthwise output is at most 6 * sum|w| + |b| before its own clamp.  False -> the planner records the block as
        three layer launches (bf16 / fp32 accumulators) instead."""
        if not (torch.isfinite(self.wd).all() and torch.isfinite(self.bd).all() and torch.isfinite(self.wp).all()):
            return False
        bound = 6.0 * self.wd.float().abs().sum((0, 1)) + self.bd.float().abs()  # per channel
        return bool(bound.max() < 3.0e4)


def fill_mb_desc(d, x_ptr, y_ptr, n, h, w, pk, dtype_code):
    d.x, d.y = x_ptr, y_ptr
    d.w_expand, d.scale_expand, d.bias_expand = pk.e.w.data_ptr(), pk.e.scale.data_ptr(), pk.e.bias.data_ptr()
    d.w_dw, d.bias_dw = pk.wd.data_ptr(), pk.bd.data_ptr()
    d.w_project, d.scale_project, d.bias_project = pk.wp.data_ptr(), pk.p.scale.data_ptr(), pk.p.bias.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Chid, d.Cout = n, h, w, pk.cin, pk.chid, pk.cout
    d.stride, d.residual, d.dtype, d.stem = pk.stride, int(pk.residual), dtype_code, pk.stem
    d.image_nw, d.w_image, d.w_image_bytes = 0, None, 0
    if w in (16, 32, 64) and not pk.stem:  # ssdk_mbk.hip: the blocks on 16- to 64-pixel-wide maps
        im = pk.image(w)
        if im is not None:
            d.image_nw, d.w_image, d.w_image_bytes = im[0], im[1].data_ptr(), im[1].numel() * 2
    return d


def xpair_supported(p1, p2, h, w):
    """An SSD extra layer (1x1 + BN + act, then 3x3 / stride 2 / pad 1 + BN + act) that csrc/ssdk_xpair.hip runs as one
    launch: small map, channel counts of its instances."""
    return (p1.kind == "dense" and p2.kind == "dense" and p1.k == 1 and p1.stride == 1 and p2.k == 3 and p2.stride == 2
            and p1.cout == p2.cin
            # exactly what ssdk_xpair() accepts: maps of <= 16 pixels (one fragment) or exactly 64 (four full fragments)
            and (h * w <= 16 or h * w == 64)
            and ((h * w + 15) // 16, p1.cin, p1.cout, p2.cout) in ((4, 512, 128, 256), (1, 256, 128, 256), (1, 256, 64, 128),
                                                                   (1, 128, 64, 128))
            and p1.act in ("none", "relu", "relu6") and p2.act in ("none", "relu", "relu6")
            and p1.scale is not None and p2.scale is not None and os.environ.get("SSDK_XPAIR", "1") != "0")


def fill_xpair_desc(d, x_ptr, y_ptr, n, h, w, p1, p2, dtype_code):
    d.x, d.y = x_ptr, y_ptr
    d.w1, d.scale1, d.bias1 = p1.w.data_ptr(), p1.scale.data_ptr(), p1.bias.data_ptr()
    d.w2, d.scale2, d.bias2 = p2.w.data_ptr(), p2.scale.data_ptr(), p2.bias.data_ptr()
    f1, f2 = p1.frag(), p2.frag()
    d.w1_frag, d.w2_frag = (f1.data_ptr(), f2.data_ptr()) if f1 is not None and f2 is not None else (None, None)
    d.N, d.H, d.W, d.Cin, d.Cmid, d.Cout = n, h, w, p1.cin, p1.cout, p2.cout
    d.act1, d.act2, d.dtype = N.ACT[p1.act], N.ACT[p2.act], dtype_code
    return d


def xpair_native(x, p1, p2):
    """Conv 1x1 + BN + act -> Conv 3x3 / stride 2 + BN + act on a small map in one launch; x channels_last."""
    N.require_device(x, "xpair")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    n, c, h, w = (int(v) for v in x.shape)
    ho, wo = _out_hw(h, w, 3, 2)
    y = torch.empty((n, p2.cout, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    d = fill_xpair_desc(N.XpairDesc(), x.data_ptr(), y.data_ptr(), n, h, w, p1, p2, N.dtype_code(x))
    with torch.cuda.device(x.device):
        rc = N.lib.ssdk_xpair(ctypes.byref(d), N.stream_ptr(x.device))
    N.check(rc, "xpair")
    STATS["native_layers"] += 2
    return y


def fuse_native(a, b, c=None, weights=(1.0, 1.0, 0.0), mode_b=N.FUSE_SAME, mode_c=N.FUSE_SAME):
    """y = w0*a + w1*R_b(b) [+ w2*R_c(c)] (BiFPN weighted fusion, csrc/ssdk_fuse.hip); channels_last tensors."""
    N.require_device(a, "fuse")
    ts = [t if t is None or t.is_contiguous(memory_format=torch.channels_last) else
          t.contiguous(memory_format=torch.channels_last) for t in (a, b, c)]
    a, b, c = ts
    n, ch, h, w = (int(v) for v in a.shape)
    y = torch.empty_like(a, memory_format=torch.channels_last)
    d = N.FuseDesc()
    d.a, d.b, d.c, d.y = a.data_ptr(), b.data_ptr(), (c.data_ptr() if c is not None else None), y.data_ptr()
    d.w0, d.w1, d.w2 = (float(v) for v in weights)
    d.mode_b, d.mode_c = mode_b, mode_c
    d.N, d.H, d.W, d.C = n, h, w, ch
    d.hb, d.wb = int(b.shape[2]), int(b.shape[3])
    if c is not None:
        d.hc, d.wc = int(c.shape[2]), int(c.shape[3])
    d.dtype = N.dtype_code(a)
    with torch.cuda.device(a.device):
        rc = N.lib.ssdk_fuse(ctypes.byref(d), N.stream_ptr(a.device))
    N.check(rc, "fuse")
    return y


def mbconv_native(x, pk, variant=0):
    """One fused inverted-residual block; x channels_last [N,Cin,H,W] -> channels_last [N,Cout,Ho,Wo].
    ``variant``: 0 automatic, 1 register-flow kernel wherever it exists, -1 LDS-tiled kernel only (ssdk_mbconv_desc)."""
    N.require_device(x, "mbconv")
    stem = pk.stem
    if stem and x.is_contiguous():
        stem = 1  # NCHW image
    else:
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        stem = 2 if stem else 0
    n, c, h, w = (int(v) for v in x.shape)
    hs, ws = _out_hw(h, w, 3, 2) if pk.stem else (h, w)
    ho, wo = _out_hw(hs, ws, 3, pk.stride)
    y = torch.empty((n, pk.cout, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    d = fill_mb_desc(N.MbConvDesc(), x.data_ptr(), y.data_ptr(), n, h, w, pk, N.dtype_code(x))
    d.stem = stem
    d.variant = int(variant)
    with torch.cuda.device(x.device):
        rc = N.lib.ssdk_mbconv(ctypes.byref(d), N.stream_ptr(x.device))
    N.check(rc, "mbconv")
    STATS["native_layers"] += 1
    return y


def _out_hw(h, w, k, stride):
    pad = k // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def wants_frag(pack, h, w, has_residual, in_layout=N.NHWC):
    """Whether one of the kernels that read the fragment-major weight image can take this layer -- the host-side mirror of the
    C dispatch (csrc/ssdk_smallmap.hip launch_conv_smallmap: 3x3 on maps of <= 64 output pixels, stride 1 | 2;
    csrc/ssdk_conv3x3s.hip launch_conv3x3_short: 3x3 / stride 1, NHWC input, Cin a multiple of 32 up to 128 -- or 256 on maps
    of <= 2500 pixels --, Cout >= 96, no residual, a map of >= 128 pixels and >= 8 columns).  Everything else (the ResNet /
    RegNet backbone 3x3s, the 80x80 tower layers) stays on kernels that read the KRSC tensor and carries no second copy."""
    if pack.kind != "dense" or pack.k != 3:
        return False
    ho, wo = _out_hw(h, w, pack.k, pack.stride)
    if ho * wo <= 64:  # conv_smallmap_kernel (stride 2: the first SSD extra, 16x16 -> 8x8)
        return True
    if pack.stride != 1 or has_residual or in_layout != N.NHWC:
        return False
    short_k = pack.cin % 32 == 0 and (pack.cin <= 128 or (pack.cin == 256 and h * w <= 2500))
    return short_k and pack.cout >= 96 and h * w >= 128 and w >= 8


def fill_desc(d, x_ptr, n, h, w, pack, dtype_code, act, y_ptr, in_layout=N.NHWC, out_layout=N.NHWC,
              residual_ptr=None, y2_ptr=None, split=None, act2=None):
    d.x, d.w = x_ptr, pack.w.data_ptr()
    frag = pack.frag() if wants_frag(pack, h, w, residual_ptr is not None, in_layout) else None
    d.w_frag = frag.data_ptr() if frag is not None else None
    d.scale = pack.scale.data_ptr() if pack.scale is not None else None
    d.bias = pack.bias.data_ptr()
    d.residual, d.y, d.y2 = residual_ptr, y_ptr, y2_ptr
    d.N, d.Cin, d.H, d.W, d.Cout = n, pack.cin, h, w, pack.cout
    d.k, d.stride, d.groups = pack.k, pack.stride, pack.groups
    d.act = N.ACT[act]
    d.act2 = N.ACT[act2 if act2 is not None else act]
    d.split = split if split is not None else pack.cout
    d.dtype, d.in_layout, d.out_layout = dtype_code, in_layout, out_layout
    return d


_SPLITK_WS = {}


def _splitk_ws(device, nbytes):
    """Zero-initialised split-K scratch per (device, stream); the kernels re-arm their counters, so it stays
    valid across calls as long as calls on it are stream-ordered."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SPLITK_WS.get(key)
    if buf is None or buf.numel() < nbytes + 256:
        buf = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
        _SPLITK_WS[key] = buf
    return buf


def conv_native(x, pack, act=None, residual=None, nchw_out=False, split=None, act2=None, res_mode=0):
    """One fused layer.  x: [N,C,H,W] tensor in channels_last memory (converted if not; the stem also takes
    plain NCHW).  Returns a channels_last tensor, or NCHW tensor(s) when ``nchw_out`` (heads)."""
    N.require_device(x, "conv")
    act = pack.act if act is None else act
    n, c, h, w = (int(v) for v in x.shape)
    in_layout = N.NHWC
    if pack.kind == "stem" and x.is_contiguous():
        in_layout = N.NCHW
    elif not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    ho, wo = _out_hw(h, w, pack.k, pack.stride)
    y2 = None
    if nchw_out:
        if split is not None:
            y = torch.empty((n, split, ho, wo), device=x.device, dtype=x.dtype)
            y2 = torch.empty((n, pack.cout - split, ho, wo), device=x.device, dtype=x.dtype)
        else:
            y = torch.empty((n, pack.cout, ho, wo), device=x.device, dtype=x.dtype)
    else:
        y = torch.empty((n, pack.cout, ho, wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
        residual = residual.contiguous(memory_format=torch.channels_last)
    d = fill_desc(N.ConvDesc(), x.data_ptr(), n, h, w, pack, N.dtype_code(x), act, y.data_ptr(), in_layout,
                  N.NCHW if nchw_out else N.NHWC, residual.data_ptr() if residual is not None else None,
                  y2.data_ptr() if y2 is not None else None, split, act2)
    d.res_mode = res_mode
    with torch.cuda.device(x.device):
        need = int(N.lib.ssdk_conv_workspace_bytes(n, pack.cin, h, w, pack.cout, pack.k, pack.stride, N.dtype_code(x)))
        if need:
            ws = _splitk_ws(x.device, need)
            wptr = (ws.data_ptr() + 255) & ~255
            rc = N.lib.ssdk_conv(ctypes.byref(d), wptr, ws.numel() - (wptr - ws.data_ptr()), N.stream_ptr(x.device))
        else:
            rc = N.lib.ssdk_conv(ctypes.byref(d), None, 0, N.stream_ptr(x.device))
    N.check(rc, "conv")
    STATS["native_layers"] += 1
    return (y, y2) if y2 is not None else y


class _Arena(object):
    """Greedy re-use of activation buffers inside one plan (the network is a chain: a buffer is free again
    once its last reader has been recorded)."""

    def __init__(self, device):
        self.device = device
        self.bufs = []  # [tensor(uint8), free]

    def get(self, nbytes):
        best = None
        for i, (t, free) in enumerate(self.bufs):
            if free and t.numel() >= nbytes and (best is None or t.numel() < self.bufs[best][0].numel()):
                best = i
        if best is None:
            self.bufs.append([torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device), False])
            return len(self.bufs) - 1
        self.bufs[best][1] = False
        return best

    def release(self, i):
        if i is not None:
            self.bufs[i][1] = True

    def ptr(self, i):
        return self.bufs[i][0].data_ptr()

    def total_bytes(self):
        return sum(t.numel() for t, _ in self.bufs)


class ExtBuf(object):
    """An external input of a plan (the image, or a backbone feature map computed outside the plan): the
    descriptors that read it are patched with the tensor's address on every run."""

    def __init__(self, index, shape):
        self.index, self.shape = index, tuple(shape)


class ConvPlan(object):
    """A recorded forward: ops (fused convs, fused MobileNet blocks, BiFPN fusions) on arena buffers.
    ``record`` by walking the model once for given input shapes; ``run(*inputs)`` replays it with ONE C call.
    Head outputs are allocated per call (they are returned to the caller); all other activations live in the
    plan's arena.  A "value" is ``(buffer index | ExtBuf, n, c, h, w)``."""

    def __init__(self, device, dtype, in_shape=None):
        self.device, self.dtype = device, dtype
        self.dtype_code = N._DTYPES[dtype]
        self.es = 2
        self.arena = _Arena(device)
        self.pinned = set()  # arena buffers read by side-lane ops: never re-used (see head())
        self.layers = []   # dicts with everything the descriptor fillers need
        self.heads = []    # (layer index, n, split | None, cout, ho, wo, tag)
        self.keep = []     # packs kept alive
        self.inputs = []   # ExtBuf
        self.ops = None
        self.report = []
        self.in_shape = tuple(in_shape) if in_shape is not None else None
        self._ctx = None  # this plan's side stream / events / op-profiling ring (include/ssdk.h "Contexts"): lazy
        self.side_chain = False  # tower chains recorded for the side stream: the plan turns its context's side lane on
        self.head_order = []     # per head: the level it belongs to (None: recording order)
        if in_shape is not None:
            self.add_input(in_shape)

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = N.Context(self.device)
        return self._ctx

    def add_input(self, shape):
        """Declare an external [N,C,H,W] input (channels_last memory at run time); returns its value."""
        e = ExtBuf(len(self.inputs), shape)
        self.inputs.append(e)
        n, c, h, w = e.shape
        return (e, n, c, h, w)

    def input_value(self):
        e = self.inputs[0]
        n, c, h, w = e.shape
        return (e, n, c, h, w)

    def conv(self, val, pack, act=None, residual=None, res_mode=0, role=None, lane=0):
        """res_mode bit 0: the residual value is half resolution (nearest x2 upsample, FPN top-down);
        bit 1: the activation follows the add (ResNet blocks).  ``role='tower'`` marks the 3x3 convs of the shared
        head towers (fpn.py:10-18) for the per-layer table: they are head convolutions, not neck.
        ``lane=2`` records the op for the executor's side stream (a chain of small-level tower layers next to the big
        levels, planner._record_extras_and_towers; include/ssdk.h: lane 2 = the in-line kernel choice): its input and
        output buffers are pinned like a side-lane head's."""
        buf, n, c, h, w = val
        assert c == pack.cin, (c, pack.cin)
        ho, wo = _out_hw(h, w, pack.k, pack.stride)
        out = self.arena.get(n * pack.cout * ho * wo * self.es)
        if lane:
            self.side_chain = True
            self.pinned.add(out)
            if not isinstance(buf, ExtBuf):
                self.pinned.add(buf)
        self.layers.append(dict(x=buf, n=n, h=h, w=w, pack=pack, act=pack.act if act is None else act, y=out,
                                res=residual[0] if residual is not None else None, res_mode=res_mode, nchw=False,
                                role=role, lane=lane))
        self.keep.append(pack)
        return (out, n, pack.cout, ho, wo)

    def mbconv(self, val, pk):
        buf, n, c, h, w = val
        assert c == pk.cin, (c, pk.cin)
        hs, ws = _out_hw(h, w, 3, 2) if pk.stem else (h, w)
        ho, wo = _out_hw(hs, ws, 3, pk.stride)
        out = self.arena.get(n * pk.cout * ho * wo * self.es)
        self.layers.append(dict(kind="mb", x=buf, n=n, h=h, w=w, pack=pk, y=out))
        self.keep.append(pk)
        return (out, n, pk.cout, ho, wo)

    def xpair(self, val, p1, p2, lane=0):
        buf, n, c, h, w = val
        assert c == p1.cin, (c, p1.cin)
        ho, wo = _out_hw(h, w, 3, 2)
        out = self.arena.get(n * p2.cout * ho * wo * self.es)
        if lane:  # a side-stream chain op: its buffers are never handed out again (see conv())
            self.side_chain = True
            self.pinned.add(out)
            if not isinstance(buf, ExtBuf):
                self.pinned.add(buf)
        self.layers.append(dict(kind="xpair", x=buf, n=n, h=h, w=w, pack=p1, pack2=p2, y=out, lane=lane))
        self.keep.extend([p1, p2])
        return (out, n, p2.cout, ho, wo)

    def stem7(self, val, pack):
        buf, n, c, h, w = val
        ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
        out = self.arena.get(n * pack.cout * ho * wo * self.es)
        self.layers.append(dict(kind="stem7", x=buf, n=n, h=h, w=w, pack=pack, y=out))
        self.keep.append(pack)
        return (out, n, pack.cout, ho, wo)

    def pool(self, val):
        buf, n, c, h, w = val
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        out = self.arena.get(n * c * ho * wo * self.es)
        self.layers.append(dict(kind="pool", x=buf, n=n, h=h, w=w, ch=c, y=out))
        return (out, n, c, ho, wo)

    def fuse(self, a, b, c=None, weights=(1.0, 1.0, 0.0), mode_b=N.FUSE_SAME, mode_c=N.FUSE_SAME):
        """y = w0*a + w1*R_b(b) [+ w2*R_c(c)] at the resolution of ``a`` (BiFPN weighted fusion)."""
        _, n, ch, h, w = a
        out = self.arena.get(n * ch * h * w * self.es)
        self.layers.append(dict(kind="fuse", a=a, b=b, c=c, w=tuple(float(v) for v in weights), mode_b=mode_b,
                                mode_c=mode_c, n=n, h=h, w_=w, ch=ch, y=out))
        return (out, n, ch, h, w)

    def head(self, val, pack, split=None, act="none", act2=None, tag="both", lane=None, position=None, level=None):
        """An NCHW output of the plan: loc|conf of one SSD level as one split GEMM (``tag='both'``) or the last
        conv of one shared tower (``tag='loc' | 'conf'``).  ``level``: the pyramid level of the output when the heads are
        recorded out of level order (small levels first, on the side stream); the outputs are returned in level order."""
        buf, n, c, h, w = val
        ho, wo = _out_hw(h, w, pack.k, pack.stride)
        # small heads are leaves of latency-bound work: they run on the executor's side stream next to the main
        # chain (SSDK_SIDE_STREAM); the big levels fill the chip on their own and stay in line.  A side-lane op
        # reads its input while the main lane keeps going, so that buffer must never be handed out again inside
        # this plan (the arena re-uses a buffer as soon as its last reader is RECORDED, which orders nothing
        # across streams): it is pinned.
        if lane is None:
            lane = 1 if n * ho * wo <= 4096 else 0
        if lane and not isinstance(buf, ExtBuf):
            self.pinned.add(buf)
        self.layers.append(dict(x=buf, n=n, h=h, w=w, pack=pack, act=act, y=None, res=None, res_mode=0, nchw=True,
                                split=split, act2=act2, lane=lane))
        entry = (len(self.layers) - 1, n, split, pack.cout, ho, wo, tag)
        if position is None:
            self.heads.append(entry)
            self.head_order.append(level)
        else:  # recorded out of level order (lane balancing): the outputs keep the level order
            self.heads.insert(position, entry)
            self.head_order.insert(position, level)
        self.keep.append(pack)

    def release(self, val):
        if not isinstance(val[0], ExtBuf) and val[0] not in self.pinned:
            self.arena.release(val[0])

    def _ptr(self, buf, patches, op_index, field):
        if buf is None:
            return 0
        if isinstance(buf, ExtBuf):
            patches.append((op_index, field, buf.index))
            return 0
        return self.arena.ptr(buf)

    def finalize(self):
        if any(o is not None for o in self.head_order):  # heads recorded out of level order: back into level order
            assert all(o is not None for o in self.head_order)
            order = sorted(range(len(self.heads)), key=lambda i: self.head_order[i])
            self.heads = [self.heads[i] for i in order]
            self.head_order = [self.head_order[i] for i in order]
        need = 0
        for L in self.layers:
            if L.get("kind") is None:
                pk = L["pack"]
                need = max(need, int(N.lib.ssdk_conv_workspace_bytes(L["n"], pk.cin, L["h"], L["w"], pk.cout, pk.k,
                                                                      pk.stride, self.dtype_code)))
        # split-K scratch (fp32 slabs + arrival counters): zero-initialised once, re-armed by the kernels.  The heads
        # may run on the executor's side stream concurrently with the main chain and get their own half.
        need = (need + 255) & ~255
        self.ws = torch.zeros(2 * need + 512, dtype=torch.uint8, device=self.device) if need else None
        self.ops = (N.Op * len(self.layers))()
        self.patches = []  # (op index, field name, external input index)
        for i, L in enumerate(self.layers):
            op = self.ops[i]
            kind = L.get("kind")
            if kind == "mb":
                op.kind = N.OP_MBCONV
                fill_mb_desc(op.mb, self._ptr(L["x"], self.patches, i, "mb.x"), self.arena.ptr(L["y"]), L["n"], L["h"],
                             L["w"], L["pack"], self.dtype_code)
                continue
            if kind == "xpair":
                op.kind = N.OP_XPAIR
                op.lane = L.get("lane", 0)
                fill_xpair_desc(op.xpair, self._ptr(L["x"], self.patches, i, "xpair.x"), self.arena.ptr(L["y"]), L["n"], L["h"],
                                L["w"], L["pack"], L["pack2"], self.dtype_code)
                continue
            if kind == "stem7":
                op.kind = N.OP_STEM7
                st, pk = op.stem, L["pack"]
                st.x = self._ptr(L["x"], self.patches, i, "stem.x")
                st.w, st.scale, st.bias, st.y = pk.w.data_ptr(), pk.scale.data_ptr(), pk.bias.data_ptr(), self.arena.ptr(L["y"])
                st.N, st.H, st.W, st.Cin, st.Cout = L["n"], L["h"], L["w"], 3, pk.cout
                st.act, st.dtype, st.in_layout = N.ACT[pk.act], self.dtype_code, N.NCHW
                continue
            if kind == "pool":
                op.kind = N.OP_POOL
                pl = op.pool
                pl.x, pl.y = self._ptr(L["x"], self.patches, i, "pool.x"), self.arena.ptr(L["y"])
                pl.N, pl.H, pl.W, pl.C, pl.dtype = L["n"], L["h"], L["w"], L["ch"], self.dtype_code
                continue
            if kind == "fuse":
                op.kind = N.OP_FUSE
                f = op.fuse
                f.a = self._ptr(L["a"][0], self.patches, i, "fuse.a")
                f.b = self._ptr(L["b"][0], self.patches, i, "fuse.b")
                f.c = self._ptr(L["c"][0], self.patches, i, "fuse.c") if L["c"] is not None else None
                f.y = self.arena.ptr(L["y"])
                f.w0, f.w1, f.w2 = L["w"]
                f.mode_b, f.mode_c = L["mode_b"], L["mode_c"]
                f.N, f.H, f.W, f.C = L["n"], L["h"], L["w_"], L["ch"]
                f.hb, f.wb = L["b"][3], L["b"][4]
                if L["c"] is not None:
                    f.hc, f.wc = L["c"][3], L["c"][4]
                f.dtype = self.dtype_code
                continue
            op.kind = N.OP_CONV
            op.lane = L.get("lane", 0)  # side stream for the small heads, decided (and its input pinned) in head()
            y_ptr = self.arena.ptr(L["y"]) if L["y"] is not None else 0
            res_ptr = self._ptr(L["res"], self.patches, i, "conv.residual") if L["res"] is not None else None
            fill_desc(op.conv, self._ptr(L["x"], self.patches, i, "conv.x"), L["n"], L["h"], L["w"], L["pack"],
                      self.dtype_code, L["act"], y_ptr, N.NHWC, N.NCHW if L["nchw"] else N.NHWC, res_ptr, None,
                      L.get("split"), L.get("act2"))
            op.conv.res_mode = L.get("res_mode", 0)
        return self

    def layer_table(self):
        """Geometry of every op of the plan: dicts with name, flops (2*MAC) and algorithmic HBM bytes
        (input + output + weights, each once) -- what the per-layer roofline numbers are computed from."""
        rows = []
        for L in self.layers:
            es = self.es
            if L.get("kind") == "stem7":
                pk, n, h, w = L["pack"], L["n"], L["h"], L["w"]
                ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
                rows.append(dict(name="stem7 3>%d @%dx%d" % (pk.cout, h, w), flops=2.0 * n * ho * wo * 147 * pk.cout,
                                 bytes=float(es * n * (3 * h * w + pk.cout * ho * wo)), kind="stem"))
                continue
            if L.get("kind") == "pool":
                n, h, w, ch = L["n"], L["h"], L["w"], L["ch"]
                ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
                rows.append(dict(name="maxpool %d @%dx%d" % (ch, h, w), flops=9.0 * n * ch * ho * wo,
                                 bytes=float(es * n * ch * (h * w + ho * wo)), kind="pool"))
                continue
            if L.get("kind") == "fuse":
                n, h, w, ch = L["n"], L["h"], L["w_"], L["ch"]
                srcs = [L["a"], L["b"]] + ([L["c"]] if L["c"] is not None else [])
                byt = es * (sum(v[1] * v[2] * v[3] * v[4] for v in srcs) + n * ch * h * w)
                rows.append(dict(name="fuse x%d %d @%dx%d" % (len(srcs), ch, h, w), flops=2.0 * len(srcs) * n * ch * h * w,
                                 bytes=float(byt), kind="fuse"))
                continue
            pk, n, h, w = L["pack"], L["n"], L["h"], L["w"]
            if L.get("kind") == "xpair":
                p2 = L["pack2"]
                ho, wo = _out_hw(h, w, 3, 2)
                macs = n * (h * w * pk.cin * pk.cout + ho * wo * 9 * p2.cin * p2.cout)
                byt = es * (n * (h * w * pk.cin + ho * wo * p2.cout) + pk.cin * pk.cout + 9 * p2.cin * p2.cout)
                rows.append(dict(name="extra %d>%d>%d k1,k3s2 @%dx%d" % (pk.cin, pk.cout, p2.cout, h, w), flops=2.0 * macs,
                                 bytes=float(byt), kind="conv"))
                continue
            if L.get("kind") == "mb":
                hs, ws = _out_hw(h, w, 3, 2) if pk.stem else (h, w)
                ho, wo = _out_hw(hs, ws, 3, pk.stride)
                if pk.stem:  # stem 3x3/s2 (cin -> chid) + depthwise + projection
                    macs = n * (hs * ws * 9 * pk.cin * pk.chid + ho * wo * 9 * pk.chid + ho * wo * pk.chid * pk.cout)
                else:
                    macs = n * (h * w * pk.cin * pk.chid + ho * wo * 9 * pk.chid + ho * wo * pk.chid * pk.cout)
                byt = es * n * (h * w * pk.cin + ho * wo * pk.cout)
                name = "mbconv%s %d>%d>%d s%d @%dx%d" % ("+stem" if pk.stem else "", pk.cin, pk.chid, pk.cout,
                                                         pk.stride, h, w)
                rows.append(dict(name=name, flops=2.0 * macs, bytes=float(byt), kind="mbconv"))
                continue
            ho, wo = _out_hw(h, w, pk.k, pk.stride)
            macs = n * ho * wo * pk.cout * (pk.cin // pk.groups) * pk.k * pk.k
            byt = es * (n * (h * w * pk.cin + ho * wo * pk.cout) + pk.cout * (pk.cin // pk.groups) * pk.k * pk.k)
            kind = "head" if L["nchw"] else ("tower" if L.get("role") == "tower" else
                                             ("dw" if pk.kind == "dw" else ("gconv" if pk.groups > 1 else "conv")))
            rows.append(dict(name="%s %d>%d k%d s%d @%dx%d" % (kind, pk.cin, pk.cout, pk.k, pk.stride, h, w),
                             flops=2.0 * macs, bytes=float(byt), kind=kind))
        return rows

    def _set_field(self, op_index, field, ptr):
        sub, name = field.split(".")
        setattr(getattr(self.ops[op_index], sub), name, ptr)

    def prepare(self, *inputs):
        """Points the recorded ops at this call's inputs and freshly allocated head outputs.  inputs: the tensors
        declared with ``add_input`` (the image: NCHW contiguous or channels_last; feature maps: converted to
        channels_last if needed).  Returns (loc tuple, conf tuple), NCHW -- filled once ``launch`` has run every op."""
        if len(inputs) != len(self.inputs):
            raise N.SsdkError("plan expects {} inputs, got {}".format(len(self.inputs), len(inputs)))
        held = []
        for e, x in zip(self.inputs, inputs):
            if tuple(x.shape) != e.shape or x.dtype != self.dtype:
                raise N.SsdkError("plan input {} was recorded as {} {}, got {} {}".format(
                    e.index, e.shape, self.dtype, tuple(x.shape), x.dtype))
        first_kind = self.ops[0].kind
        image_first = len(self.inputs) == 1 and self.inputs[0].shape[1] <= 4
        xs = list(inputs)
        if image_first:
            x = xs[0]
            if first_kind == N.OP_STEM7:  # ResNet stem reads the image in either layout
                first = self.ops[0].stem
                if x.is_contiguous():
                    first.in_layout = N.NCHW
                else:
                    x = x.contiguous(memory_format=torch.channels_last)
                    first.in_layout = N.NHWC
            elif first_kind == N.OP_MBCONV:  # stem + first block fused: the image is read by the block kernel
                first = self.ops[0].mb
                if x.is_contiguous():
                    first.stem = 1
                else:
                    x = x.contiguous(memory_format=torch.channels_last)
                    first.stem = 2
            else:
                first = self.ops[0].conv
                if x.is_contiguous():
                    first.in_layout = N.NCHW if self.layers[0]["pack"].kind == "stem" else N.NHWC
                    if first.in_layout == N.NHWC:
                        x = x.contiguous(memory_format=torch.channels_last)
                else:
                    x = x.contiguous(memory_format=torch.channels_last)
                    first.in_layout = N.NHWC
            xs[0] = x
        else:
            xs = [x if x.is_contiguous(memory_format=torch.channels_last) else
                  x.contiguous(memory_format=torch.channels_last) for x in xs]
        held.extend(xs)
        for (oi, field, ei) in self.patches:
            self._set_field(oi, field, xs[ei].data_ptr())
        loc, conf = [], []
        for (li, n, split, cout, ho, wo, tag) in self.heads:
            if tag == "both":
                l = torch.empty((n, split, ho, wo), device=self.device, dtype=self.dtype)
                c = torch.empty((n, cout - split, ho, wo), device=self.device, dtype=self.dtype)
                self.ops[li].conv.y, self.ops[li].conv.y2 = l.data_ptr(), c.data_ptr()
                loc.append(l)
                conf.append(c)
            else:
                t = torch.empty((n, cout, ho, wo), device=self.device, dtype=self.dtype)
                self.ops[li].conv.y = t.data_ptr()
                (loc if tag == "loc" else conf).append(t)
        if os.environ.get("SSDK_OPS_TRACE"):
            import sys
            for (li, n, split, cout, ho, wo, tag), t in zip(self.heads, [None] * len(self.heads)):
                sys.stderr.write("[plan] head op %d %s n=%d cout=%d %dx%d y=%#x\n" % (li, tag, n, cout, ho, wo, self.ops[li].conv.y or 0))
        self._held = held  # converted inputs stay alive until the next prepare()
        return tuple(loc), tuple(conf)

    def launch(self, lo=0, hi=None, stream=None):
        """Enqueue ops [lo, hi) of the plan on ``stream`` (default: the current stream) with ONE C call."""
        hi = len(self.layers) if hi is None else hi
        if hi <= lo:
            return
        ops = self.ops if lo == 0 else (N.Op * (hi - lo)).from_address(ctypes.addressof(self.ops) + lo * ctypes.sizeof(N.Op))
        sp = N.stream_ptr(self.device) if stream is None else ctypes.c_void_p(stream.cuda_stream)
        if self.side_chain:  # (SSDK_LEVEL_LANES=0 records no chains: planner._record_extras_and_towers)
            self.ctx.auto_side_lane()  # on by default; an explicit set_side_lane(False) / SSDK_SIDE_STREAM=0 is respected
        with torch.cuda.device(self.device):
            if self.ws is not None:
                wptr = (self.ws.data_ptr() + 255) & ~255
                rc = N.lib.ssdk_run_ops_ctx(self.ctx.ptr, ops, hi - lo, wptr,
                                            self.ws.numel() - (wptr - self.ws.data_ptr()), sp)
            else:
                rc = N.lib.ssdk_run_ops_ctx(self.ctx.ptr, ops, hi - lo, None, 0, sp)
        N.check(rc, "run_ops")
        STATS["native_layers"] += hi - lo

    def run(self, *inputs):
        """prepare + launch of the whole plan on the current stream.  Returns (loc tuple, conf tuple), NCHW."""
        out = self.prepare(*inputs)
        self.launch()
        STATS["plan_runs"] += 1
        return out


def sequential_groups(seq):
    """[Conv2d, (BatchNorm2d), (activation)]* -> [(conv, bn, act)], or None if the pattern does not match."""
    mods = list(seq.children())
    i, out = 0, []
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, nn.Conv2d):
            return None
        bn, act, j = None, "none", i + 1
        if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
            bn, j = mods[j], j + 1
        a = _act_name(mods[j]) if j < len(mods) else None
        if a is not None:
            act, j = a, j + 1
        out.append((conv, bn, act))
        i = j
    return out


class FusedSequentialMixin(object):
    """nn.Sequential of [Conv2d, (BatchNorm2d), (activation)]* groups: in eval mode on a HIP device each
    group is one fused launch; in training mode (or with SSDK_FUSED_CONV=0) the plain torch modules run."""

    def _packs(self, dtype):
        cache = getattr(self, "_ssdk_packs", None)
        if cache is None or cache[0] != dtype:
            groups = sequential_groups(self)
            packs = None
            if groups is not None and all(conv_kind(c) is not None for c, _, _ in groups):
                packs = [ConvPack(c, b, a, dtype) for c, b, a in groups]
            cache = (dtype, packs)
            object.__setattr__(self, "_ssdk_packs", cache)
        return cache[1]

    def train(self, mode=True):
        object.__setattr__(self, "_ssdk_packs", None)  # weights may change: re-fold on the next eval forward
        return nn.Sequential.train(self, mode)

    def _apply(self, fn, *a, **kw):
        object.__setattr__(self, "_ssdk_packs", None)
        return nn.Sequential._apply(self, fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        # reached for EVERY module of the tree whichever ancestor's load_state_dict() was called (a parent's
        # load_state_dict never calls a child's): the folded weights of this block are stale from here on
        object.__setattr__(self, "_ssdk_packs", None)
        return nn.Sequential._load_from_state_dict(self, *a, **kw)

    def forward(self, x):
        if self.training or not fused_enabled() or not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16):
            return nn.Sequential.forward(self, x)
        packs = self._packs(x.dtype)
        if packs is None:
            STATS["torch_fallback_layers"] += 1
            return nn.Sequential.forward(self, x)
        for p in packs:
            x = conv_native(x, p)
        return x
