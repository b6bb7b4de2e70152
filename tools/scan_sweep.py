"""A/B sweep of the decode stage's environment knobs on the headline shapes (SSD-MobileNetV2@512, batch 64, bf16):
SSDK_SCAN_REG (sample tiles in registers) x SSDK_SCAN_PF (ring depth) x SSDK_TARGET_WGS (unit size), on SURVEY 8d's
realistic heads and on the all-equal heads of the reference-init network.  Usage: python tools/scan_sweep.py"""
import itertools
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch

from ssds import _native as N
from ssds.modeling.layers import box

B, A, C = 64, 6, 80
sizes = [32, 16, 8, 4, 2, 1]
strides = [16, 32, 64, 128, 256, 512]
anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
torch.manual_seed(0)
loc = [torch.randn(B, A * 4, h, h, device="cuda").mul(0.5).to(torch.bfloat16) for h in sizes]
real = [torch.sigmoid(torch.randn(B, A * C, h, h, device="cuda") * 1.5 - 4.6).to(torch.bfloat16) for h in sizes]
ties = [torch.full((B, A * C, h, h), 0.01, device="cuda").to(torch.bfloat16) for h in sizes]
nbytes = sum(B * A * C * h * h * 2 for h in sizes)


def run(conf, reps=30):
    ctx = N.Context(torch.device("cuda", 0))
    for _ in range(3):
        box.decode_nms(loc, conf, anchors, 0.01, 300, True, 0.6, 100, True, ctx=ctx)
    ctx.set_profiling(True)
    for _ in range(reps):
        box.decode_nms(loc, conf, anchors, 0.01, 300, True, 0.6, 100, True, ctx=ctx)
    torch.cuda.synchronize()
    t = [ctx.timings_ms(i) for i in range(reps)]
    return sum(x[0] for x in t) / reps * 1e3, sum(x[1] for x in t) / reps * 1e3


combos = list(itertools.product(os.environ.get("SWEEP_REG", "8,16,32").split(","), os.environ.get("SWEEP_PF", "4,8").split(","),
                                os.environ.get("SWEEP_WGS", "512,640,768,1024").split(",")))
for reg, pf, wgs in combos:
    os.environ.update(SSDK_SCAN_REG=reg, SSDK_SCAN_PF=pf, SSDK_TARGET_WGS=wgs)
    s, t = run(real)
    s2, t2 = run(ties)
    print("REG=%-2s PF=%s WGS=%-4s  realistic: scan %5.1f us (%.2f TB/s) tail %5.1f us | all-equal: scan %5.1f tail %5.1f" % (
        reg, pf, wgs, s, nbytes / s / 1e6, t, s2, t2), flush=True)
