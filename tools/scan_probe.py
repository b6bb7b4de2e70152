"""Splits the time of scan_kernel on the headline shapes (SSD-MobileNetV2@512, batch 64, 6 anchors x 80 classes,
levels 32..1) into its streaming part and its candidate handling: the same tensors decoded with
  all-equal scores above the threshold (the bench's random-init case: everything ties at the focal prior),
  uniform random scores with ~half above the threshold, sparse scores (0.1 % above), and nothing above.
Usage: python tools/scan_probe.py"""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch

from ssds import _native as N
from ssds.modeling.layers import box

# PROBE_SHAPE=bifpn896: config 5's heads (batch 16, 9 anchors, levels 112..7); PROBE_DTYPE=f16: fp16 heads
DT = torch.float16 if os.environ.get("PROBE_DTYPE", "bf16") == "f16" else torch.bfloat16
if os.environ.get("PROBE_SHAPE", "ssd512") == "bifpn896":
    B, A, C = 16, 9, 80
    sizes = [112, 56, 28, 14, 7]
    strides = [8, 16, 32, 64, 128]
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.52, 3.175])) for s in strides)
elif os.environ.get("PROBE_SHAPE") == "fpn640":
    B, A, C = 32, 9, 80
    sizes = [80, 40, 20, 10, 5]
    strides = [8, 16, 32, 64, 128]
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.52, 3.175])) for s in strides)
else:
    B, A, C = 64, 6, 80
    sizes = [32, 16, 8, 4, 2, 1]
    strides = [16, 32, 64, 128, 256, 512]
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
torch.manual_seed(0)
loc = [torch.randn(B, A * 4, h, h, device="cuda").mul(0.1).to(DT) for h in sizes]
nbytes = sum(B * A * C * h * h * 2 for h in sizes)


ctx = N.Context(torch.device("cuda", 0))
STAGE_BYTES = nbytes + sum(B * A * 4 * h * h * 2 for h in sizes) + B * (2 * 24 * len(sizes) * 300 + 24 * 100)


def run(name, make, thr=0.01, reps=20):
    conf = [make(B, A * C, h, h).to(DT) for h in sizes]
    for _ in range(3):
        box.decode_nms(loc, conf, anchors, thr, 300, True, 0.6, 100, True, ctx=ctx)
    ctx.set_profiling(True)
    for _ in range(reps):
        box.decode_nms(loc, conf, anchors, thr, 300, True, 0.6, 100, True, ctx=ctx)
    torch.cuda.synchronize()
    t = [ctx.timings_ms(i) for i in range(reps)]
    ctx.set_profiling(False)
    scan = sum(x[0] for x in t) / reps
    lvl = sum(x[1] for x in t) / reps
    nms = sum(x[2] for x in t) / reps
    ctx.set_profiling(2)  # one interval around the whole stage
    for _ in range(reps):
        box.decode_nms(loc, conf, anchors, thr, 300, True, 0.6, 100, True, ctx=ctx)
    torch.cuda.synchronize()
    tot = sum(ctx.timings_ms(i)[0] for i in range(reps)) / reps
    ctx.set_profiling(False)
    extra = ""
    if os.environ.get("SSDK_TAIL_STAMPS") and os.environ.get("SSDK_DECODE_FUSED", "1") != "0":
        st = ctx.tail_stamps()
        d = lambda a, b: (st[b] - st[a]) / 1e3  # noqa: E731
        extra = ("\n      levelsel wg(0,0) (kcycles): load+bound %.1f | hist %.1f cut+scatter %.1f resolve+order %.1f | decode %.1f = %.1f"
                 "\n      nmswalk wg0 (kcycles): keys %.1f | C zero+hist %.1f scan %.1f scatter %.1f order+gather %.1f | D walk %.1f | E %.1f"
                 " = %.1f  (round of %d, kept %d)" % (
                     d(0, 1), d(1, 21), d(21, 22), d(22, 2), d(2, 16), d(0, 16),
                     d(3, 6), d(6, 7), d(7, 8), d(8, 17), d(17, 9), d(9, 4), d(4, 5), d(3, 5), st[11], st[12]))
        extra += "\n      decode detail (kcycles): gather issue %.1f decode+store(first) %.1f second %.1f" % (d(2, 13), d(13, 15), d(15, 16))
        sc = st[24:]
        e = lambda a, b: (sc[b] - sc[a]) / 1e3  # noqa: E731
        if sc[11]:  # scan16_kernel
            extra += ("\n      scan16 wg0 (kcycles): H top2 %.1f hist %.1f cut %.1f tie-prefix %.1f | stream %.1f | extract %.1f "
                      "select %.1f = %.1f  fast=%d winners=%d" % (
                          e(0, 8), e(8, 9), e(9, 10), e(10, 1), e(1, 2), e(2, 11), e(11, 3), e(0, 4), sc[5] >> 32,
                          sc[5] & 0xffffffff))
            W = 4096
            tl = ctx.tail_stamps(48 + 2 * W)[48:]
            se = [(tl[2 * i], tl[2 * i + 1]) for i in range(W) if tl[2 * i + 1]]
            if se:
                slow = sorted(((tl[2 * i + 1] - tl[2 * i]) / 100.0, i, tl[2 * i + 1] & 1) for i in range(W) if tl[2 * i + 1])[-6:]
                nfb = sum(1 for i in range(W) if tl[2 * i + 1] & 1)
                flags = [sum(1 for i in range(W) if tl[2 * i + 1] & b) for b in (2, 4, 8, 16)]
                flags.append(sum(1 for i in range(W) if (tl[2 * i + 1] & 21) == 5))
                extra += ("\n      scan: near-tie rule: %d units counted their sample, %d ended on a predicted cut, %d settled a tie; %d units "
                          "overflowed a key segment, %d predictions came back short" % tuple(flags))
                extra += "\n      scan: %d workgroups took the exact fallback; slowest (us, block %% B = image, block // B = unit, fallback): %s" % (
                    nfb, " ".join("(%.0f,%d,%d,%d)" % (d_, i % B, i // B, f) for d_, i, f in slow))
                dbg = tl[2 * 3000:2 * 3000 + 4]
                if dbg[0]:
                    extra += ("\n      scan overflow debug: block %d cut16 0x%04x ntiles %d | wave key counts %d %d %d %d | S %d sstride %d eq %d above %d"
                              " cb %d tie_rich %d thr16 0x%04x" % (dbg[0] >> 32, (dbg[0] >> 16) & 0xffff, dbg[0] & 0xffff, dbg[1] >> 48, (dbg[1] >> 32) & 0xffff,
                                                                   (dbg[1] >> 16) & 0xffff, dbg[1] & 0xffff, dbg[2] >> 48, (dbg[2] >> 32) & 0xffff,
                                                                   (dbg[2] >> 16) & 0xffff, dbg[2] & 0xffff, dbg[3] >> 32, dbg[3] & 1, (dbg[3] >> 8) & 0xffff))
                t0 = min(a for a, _ in se)
                st_ = sorted((a - t0) / 100.0 for a, _ in se)
                en_ = sorted((b - t0) / 100.0 for _, b in se)
                du_ = sorted((b - a) / 100.0 for a, b in se)
                q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]  # noqa: E731
                extra += ("\n      scan timeline of %d workgroups (us from the first start): starts p50 %.1f p90 %.1f max %.1f | ends p10 "
                          "%.1f p50 %.1f p90 %.1f max %.1f | durations p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
                              len(se), q(st_, .5), q(st_, .9), st_[-1], q(en_, .1), q(en_, .5), q(en_, .9), en_[-1],
                              q(du_, .1), q(du_, .5), q(du_, .9), du_[-1]))
        else:
            extra += "\n      scan wg0 (kcycles): " + " ".join("%.1f" % ((b - a) / 1e3) for a, b in zip(sc[0:5], sc[1:5]))
            extra += "  fast=%d winners=%d" % (sc[5] >> 32, sc[5] & 0xffffffff)
    print("%-28s scan %6.1f us (%5.2f TB/s)  levelsel %5.1f us  nmswalk %5.1f us | stage in ONE interval %6.1f us = %.3f of 8 TB/s%s" % (
        name, scan * 1e3, nbytes / scan / 1e9, lvl * 1e3, nms * 1e3, tot * 1e3, STAGE_BYTES / tot / 1e9 / 8000, extra),
        flush=True)


if os.environ.get("PROBE_CFG"):  # the bench's own input: reference-init network of that configuration on torch.rand images
    from ssds.core import config as _C
    from ssds.modeling import model_builder as _mb

    _cfg = _C.cfg_from_file(os.environ["PROBE_CFG"])
    torch.manual_seed(1234)
    _model = _mb.create_model(_cfg.MODEL).eval().cuda().to(DT)
    _S = int(_cfg.MODEL.IMAGE_SIZE[0])
    with torch.no_grad():
        _loc, _conf = _model(torch.rand(B, 3, _S, _S, device="cuda").to(DT))
    _by_h = {int(c.shape[-1]): c.float() for c in _conf}
    assert sorted(_by_h) == sorted(sizes), (sorted(_by_h), sizes)
    run("bench input of " + os.path.basename(os.environ["PROBE_CFG"]), lambda b, ac, h, w: _by_h[h])
run("SURVEY 8d heads sigmoid(N(-4.6,1.5))", lambda *s: torch.sigmoid(torch.randn(*s, device="cuda") * 1.5 - 4.6))
run("all equal (focal prior)", lambda *s: torch.full(s, 0.01, device="cuda"))
# the reference-init network behind a SHARED tower (FPN / BiFPN bench input): logits -log(99) +- a little, i.e. a handful of
# adjacent 16-bit values around the threshold, each occurring millions of times
run("near ties: sigmoid(N(-4.595, 0.02))", lambda *s: torch.sigmoid(torch.randn(*s, device="cuda") * 0.02 - 4.595))
run("uniform, half above thr", lambda *s: torch.rand(*s, device="cuda") * 0.02)
run("sparse, 0.1% above thr", lambda *s: torch.rand(*s, device="cuda") * 0.01001)
run("nothing above thr", lambda *s: torch.rand(*s, device="cuda") * 0.009)
run("trained-like (1% above .05)", lambda *s: torch.rand(*s, device="cuda").pow(8), thr=0.05)
