"""Layer-by-layer audit of a recorded plan (shared by tests/test_gpu_plan_audit.py and tools/plan_trace.py).

Every op of the plan is launched ON ITS OWN and its output is compared with the same op computed in fp32 by PyTorch-ROCm on
the op's ACTUAL input (the plan's own activation: the errors of earlier layers do not count), so what is left is the
rounding of that one kernel -- median ~1e-3, 99.9th percentile ~1e-2 of the layer's RMS in bf16, an eighth of that in
fp16 -- whatever the depth of the network.  That is the comparison that discriminates in bf16: end to end, a random-weight
MobileNetV2 / ResNet-50 executed in bf16 is 0.5-0.9 RMS away from its fp32 self at the deep levels (PyTorch-ROCm's own
execution is), and a bar relative to that floor passes an all-zero head (Weak 2).

The op runs at the plan's full batch (the planner and the C dispatch pick kernels by shape); the fp32 reference is computed
for a SUBSET of the images (ops are independent per image), which keeps the audit of FPN-ResNet50@640 at a few seconds.

Reference arithmetic per op kind (the folded / rounded weights of the pack are the ground truth: folding is tested in
tests/test_model_cpu.py and end to end by the fixtures):
  conv        F.conv2d on the KRSC weights, * scale + bias, activation, residual (before / after the activation, same or
              half resolution: fused_conv.ConvPlan.conv), heads with their split activations          ssd.py:68-73, fpn.py:80-97
  stem7 pool  7x7/s2 stem, 3x3/s2 max pool                                                              resnet.py:41-47
  mb          expand 1x1 (or the 3x3/s2 stem) + bias, ReLU6 -> depthwise 3x3 + bias, ReLU6 -> project * scale + bias (+ x)
                                                                                                        mobilenet.py:56, 84-89
  xpair       1x1 + BN + act, rounded to the model dtype, 3x3/s2 + BN + act                             basic_layers.py:40-57
  fuse        w0 a + w1 R(b) [+ w2 R(c)], R = same | nearest x2 | max_pool2d(2)                         bifpn.py:41-62
"""
import torch
import torch.nn.functional as F

from ssds import _native as N
from ssds.modeling.layers import fused_conv as FC

# (median, p99.9, max) of |kernel - fp32| / rms(fp32) per op.  bf16 rounds at 2^-9 relative.  Measured in round 5 on every op
# of the three bench plans (profiles/r05_plan_audit_*.txt): bf16 <= 0.00097 / 0.015 / 0.065 (SSD-MobileNetV2@512, 28 ops incl.
# the fused inverted-residual blocks with their fp16 internal tensors; FPN-ResNet50@640, 112 ops), fp16 <= 0.00012 / 0.0019 /
# 0.005 (BiFPN-RegNetX008@896, 119 ops).  The bars leave a factor ~2; a wrong wire, tap order, fold or layout is >= 1, a
# kernel that reads an accumulator too early (ssdk_mbk.hip's first version) 0.2 - 0.35 in the median.
BARS = {
    torch.bfloat16: {"default": (2e-3, 0.03, 0.1), "fused": (2e-3, 0.03, 0.1)},
    torch.float16: {"default": (4e-4, 6e-3, 0.03), "fused": (4e-4, 6e-3, 0.03)},
}
MAX_ZERO_FRACTION = 0.95  # an fp32 output with more zeros than this is a dead layer: the comparison would decide nothing


def act_fn(y, act):
    if act == "relu":
        return y.clamp(min=0)
    if act == "relu6":
        return y.clamp(0, 6)
    if act == "silu":
        return y * torch.sigmoid(y)
    if act == "sigmoid":
        return torch.sigmoid(y)
    return y


def stats(got, want):
    """(median, p99.9, max) of |got - want| / rms(want), and the fraction of exact zeros in ``want``."""
    g, w = got.float(), want.float()
    rms = max(float(w.pow(2).mean().sqrt()), 1e-12)
    e = ((g - w).abs() / rms).flatten()
    if e.numel() > 20_000_000:  # kthvalue on a sample is enough
        e = e[torch.randint(0, e.numel(), (20_000_000,), device=e.device)]
    k = max(int(e.numel() * 0.999), 1)
    return float(e.median()), float(e.kthvalue(k).values), float(e.max()), float((w == 0).float().mean())


def default_images(n):
    """Indices of the audited images of a batch of ``n`` (see PlanAudit.__init__)."""
    return sorted(set(min(j, n - 1) for j in (0, n // 4 + 1, n // 2 + 2, n - 1)))


class PlanAudit(object):
    def __init__(self, plan, inputs, images=None):
        """``plan``: a finalized ConvPlan; ``inputs``: the tensors of one call (``plan.prepare(*inputs)`` is run here);
        ``images``: indices of the images the fp32 reference is computed for.  Default (round 6): FOUR images spread over the
        batch -- first, last and two whose index is 1 and 2 modulo 4 / 8 (n // 4 + 1, n // 2 + 2), i.e. INTERIOR to the image
        groups the small-map kernels hand to one workgroup (2, 4 or 8 whole images per workgroup: a fault confined to the
        middle of a group is invisible at images 0 and n - 1, which round 5 looked at)."""
        self.plan, self.dtype = plan, plan.dtype
        self.outs = plan.prepare(*inputs)
        self.inputs = list(plan._held)
        n = plan.inputs[0].shape[0]
        self.sel = sorted(set(min(max(int(j), 0), n - 1) for j in
                              (images if images is not None else default_images(n))))

    def view(self, buf, n, c, h, w):
        """logical [n, c, h, w] view of an arena buffer (NHWC memory) or of an external input"""
        if isinstance(buf, FC.ExtBuf):
            return self.inputs[buf.index]
        t = self.plan.arena.bufs[buf][0][: n * c * h * w * self.plan.es].view(self.dtype).view(n, h, w, c)
        return t.permute(0, 3, 1, 2)

    def take(self, buf, n, c, h, w):
        return self.view(buf, n, c, h, w)[self.sel].float().clone()

    # ---- fp32 references, on the selected images ------------------------------------------------------------------------------
    def _ref_conv(self, i, L):
        pk = L["pack"]
        xin = self.take(L["x"], L["n"], pk.cin, L["h"], L["w"])
        res = None
        if L["res"] is not None:
            rh, rw = ((L["h"] // 2, L["w"] // 2) if (L.get("res_mode", 0) & 1)
                      else FC._out_hw(L["h"], L["w"], pk.k, pk.stride))
            res = self.take(L["res"], L["n"], pk.cout, rh, rw)

        def run():
            if pk.kind == "stem":
                wt = pk.w.float().permute(0, 3, 1, 2).contiguous()
                want = F.conv2d(xin, wt, None, pk.stride, pk.k // 2) + pk.bias.view(1, -1, 1, 1)
            elif pk.kind == "dw":
                wt = pk.w.float().permute(2, 0, 1).unsqueeze(1).contiguous()
                want = F.conv2d(xin, wt, None, pk.stride, 1, 1, pk.cin)
                want = want * pk.scale.view(1, -1, 1, 1) + pk.bias.view(1, -1, 1, 1)
            else:
                wt = pk.w.float().permute(0, 3, 1, 2).contiguous()
                want = F.conv2d(xin, wt, None, pk.stride, pk.k // 2, 1, pk.groups)
                if pk.scale is not None:
                    want = want * pk.scale.view(1, -1, 1, 1)
                want = want + pk.bias.view(1, -1, 1, 1)
            rm = L.get("res_mode", 0)
            if L["nchw"]:  # a head: NCHW outputs of the plan (split | single)
                plan = self.plan
                hi = [hh for hh in plan.heads if hh[0] == i][0]
                pos = plan.heads.index(hi)
                split, tag = hi[2], hi[6]
                if tag == "both":
                    got = torch.cat([self.outs[0][pos], self.outs[1][pos]], 1)
                    want = torch.cat([act_fn(want[:, :split], L["act"]), act_fn(want[:, split:], L.get("act2") or L["act"])], 1)
                else:
                    idx = [hh for hh in plan.heads if hh[6] == tag].index(hi)
                    got = (self.outs[0] if tag == "loc" else self.outs[1])[idx]
                    want = act_fn(want, L["act"])
                return got[self.sel], want
            r = res
            if r is not None and (rm & 1):
                r = F.interpolate(r, scale_factor=2, mode="nearest")
            if r is not None and (rm & 2):
                want = act_fn(want.to(self.dtype).float() + r, L["act"])
            elif r is not None:
                want = act_fn(want, L["act"]).to(self.dtype).float() + r
            else:
                want = act_fn(want, L["act"])
            return self.view(L["y"], L["n"], pk.cout, want.shape[2], want.shape[3])[self.sel], want

        return run

    def _ref_mb(self, i, L):
        pk = L["pack"]
        xin = self.take(L["x"], L["n"], pk.cin, L["h"], L["w"])

        def run():
            if pk.stem:  # K layout (ky, kx padded 3 -> 8, ci padded -> 4), fused_conv.MbPack
                we = pk.e.w.float().view(pk.chid, 3, 8, 4)[:, :, :3, : pk.cin].permute(0, 3, 1, 2).contiguous()
                e = F.conv2d(xin, we, None, 2, 1)
            else:
                e = F.conv2d(xin, pk.e.w.float().reshape(pk.chid, pk.cin, 1, 1))
            e = (e * pk.e.scale.view(1, -1, 1, 1) + pk.e.bias.view(1, -1, 1, 1)).clamp(0, 6)
            wd = pk.wd.float().permute(2, 0, 1).unsqueeze(1).contiguous()
            d = (F.conv2d(e, wd, None, pk.stride, 1, 1, pk.chid) + pk.bd.float().view(1, -1, 1, 1)).clamp(0, 6)
            y = F.conv2d(d, pk.wp.float().reshape(pk.cout, pk.chid, 1, 1))
            y = y * pk.p.scale.view(1, -1, 1, 1) + pk.p.bias.view(1, -1, 1, 1)
            if pk.residual:
                y = y.to(self.dtype).float() + xin
            return self.view(L["y"], L["n"], pk.cout, y.shape[2], y.shape[3])[self.sel], y

        return run

    def _ref_xpair(self, i, L):
        p1, p2 = L["pack"], L["pack2"]
        xin = self.take(L["x"], L["n"], p1.cin, L["h"], L["w"])

        def run():
            m = F.conv2d(xin, p1.w.float().permute(0, 3, 1, 2).contiguous())
            m = act_fn(m * p1.scale.view(1, -1, 1, 1) + p1.bias.view(1, -1, 1, 1), p1.act).to(self.dtype).float()
            y = F.conv2d(m, p2.w.float().permute(0, 3, 1, 2).contiguous(), None, 2, 1)
            y = act_fn(y * p2.scale.view(1, -1, 1, 1) + p2.bias.view(1, -1, 1, 1), p2.act)
            return self.view(L["y"], L["n"], p2.cout, y.shape[2], y.shape[3])[self.sel], y

        return run

    def _ref_fuse(self, i, L):
        def src(v, mode, h, w):
            if v is None:
                return None
            t = self.take(*v)
            if mode == N.FUSE_UP2:
                t = F.interpolate(t, scale_factor=2, mode="nearest")
            elif mode == N.FUSE_POOL2:
                t = F.max_pool2d(t, kernel_size=2)
            assert tuple(t.shape[2:]) == (h, w), (t.shape, h, w)
            return t

        h, w = L["h"], L["w_"]
        a = src(L["a"], N.FUSE_SAME, h, w)
        b = src(L["b"], L["mode_b"], h, w)
        c = src(L["c"], L["mode_c"], h, w)

        def run():
            w0, w1, w2 = L["w"]
            y = w0 * a + w1 * b
            if c is not None:
                y = y + w2 * c
            return self.view(L["y"], L["n"], L["ch"], h, w)[self.sel], y

        return run

    def _ref_stem7(self, i, L):
        pk = L["pack"]
        xin = self.take(L["x"], L["n"], 3, L["h"], L["w"])

        def run():
            wt = pk.w.float()[:, :, :7, :3].permute(0, 3, 1, 2).contiguous()
            y = F.conv2d(xin, wt, None, 2, 3) * pk.scale.view(1, -1, 1, 1) + pk.bias.view(1, -1, 1, 1)
            y = act_fn(y, pk.act)
            return self.view(L["y"], L["n"], pk.cout, y.shape[2], y.shape[3])[self.sel], y

        return run

    def _ref_pool(self, i, L):
        xin = self.take(L["x"], L["n"], L["ch"], L["h"], L["w"])

        def run():
            y = F.max_pool2d(xin, 3, 2, 1)
            return self.view(L["y"], L["n"], L["ch"], y.shape[2], y.shape[3])[self.sel], y

        return run

    def plan_kernels(self):
        """Kernel name per op as the plan's OWN single call launches them (op profiling on, side lane off): neighbouring
        small-map heads run as members of ONE conv_smallmap_group launch there, while the op-by-op audit below launches each
        of them alone as conv_smallmap (round-5 review: the audit's kernel column named a kernel the timed plan does not run).
        The group's launch is booked on its first member; the other members report the group's name too."""
        ctx = self.plan.ctx
        ctx.set_side_lane(False)
        ctx.set_op_profiling(True)
        try:
            with torch.no_grad():
                self.plan.launch()
            torch.cuda.synchronize()
            names = [k for k, _ in ctx.op_timings()]
        finally:
            ctx.set_op_profiling(False)
            ctx.set_side_lane(None)
        out, last = [], ""
        for k in names:  # members 2.. of a grouped launch carry an empty / repeated name: give them the group's
            last = k if k else last
            out.append((k or last).replace("_kernel", ""))
        return out

    def run(self):
        """Launches the ops one by one.  -> list of dicts (index, name, kernel, plan_kernel, kind, median, p999, max, zeros) in
        plan order; ``kernel`` ran in the audit, ``plan_kernel`` is what the plan's single call runs for that op."""
        plan = self.plan
        table = plan.layer_table()
        refs = {None: self._ref_conv, "mb": self._ref_mb, "xpair": self._ref_xpair, "fuse": self._ref_fuse,
                "stem7": self._ref_stem7, "pool": self._ref_pool}
        rows = []
        tf32 = torch.backends.cudnn.allow_tf32
        torch.backends.cudnn.allow_tf32 = False
        try:
            for i, L in enumerate(plan.layers):
                kind = L.get("kind")
                with torch.no_grad():
                    ref = refs[kind](i, L)  # copies the op's inputs BEFORE the launch (the arena may recycle them later)
                    plan.launch(i, i + 1)
                    torch.cuda.synchronize()
                    kern = N.last_kernel()
                    got, want = ref()
                    med, p999, mx, zeros = stats(got, want)
                rows.append(dict(index=i, name=table[i]["name"], kernel=kern.replace("_kernel", ""),
                                 kind="fused" if kind in ("mb", "xpair") else "default",
                                 median=med, p999=p999, max=mx, zeros=zeros))
        finally:
            torch.backends.cudnn.allow_tf32 = tf32
        try:
            pk = self.plan_kernels()
        except Exception:  # (a plan without a context of its own: the column stays empty)
            pk = []
        for r, k in zip(rows, pk + [""] * (len(rows) - len(pk))):
            r["plan_kernel"] = k
        return rows


def format_rows(rows):
    out = ["%3s %-38s %-26s %-26s %9s %9s %9s %6s" % ("#", "layer", "kernel (audited alone)", "kernel (in the plan's call)",
                                                      "median", "p99.9", "max", "zeros")]
    for r in rows:
        pk = r.get("plan_kernel", "")
        out.append("%3d %-38s %-26s %-26s %9.5f %9.5f %9.5f %6.3f" % (
            r["index"], r["name"], r["kernel"], "=" if pk == r["kernel"] else pk, r["median"], r["p999"], r["max"], r["zeros"]))
    return "\n".join(out)


def failures(rows, dtype):
    """rows that miss their bar, as strings (empty: the plan is at its rounding level everywhere)"""
    bad = []
    for r in rows:
        b = BARS[dtype][r["kind"]]
        if r["zeros"] > MAX_ZERO_FRACTION:
            bad.append("op %d %s: fp32 output is %.1f %% zeros (dead layer: re-seed)" % (r["index"], r["name"], 100 * r["zeros"]))
        elif not (r["median"] <= b[0] and r["p999"] <= b[1] and r["max"] <= b[2]):
            bad.append("op %d %s on %s: median / p99.9 / max = %.5f / %.5f / %.5f of the layer's RMS, bar %s"
                       % (r["index"], r["name"], r["kernel"], r["median"], r["p999"], r["max"], b))
    return bad
