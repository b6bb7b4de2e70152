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

B, A, C = 64, 6, 80
sizes = [32, 16, 8, 4, 2, 1]
strides = [16, 32, 64, 128, 256, 512]
anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
torch.manual_seed(0)
loc = [torch.randn(B, A * 4, h, h, device="cuda").mul(0.1).to(torch.bfloat16) for h in sizes]
nbytes = sum(B * A * C * h * h * 2 for h in sizes)


def run(name, make, thr=0.01, reps=20):
    conf = [make(B, A * C, h, h).to(torch.bfloat16) for h in sizes]
    for _ in range(3):
        box.decode_nms(loc, conf, anchors, thr, 300, True, 0.6, 100, True)
    N.set_profiling(True)
    for _ in range(reps):
        box.decode_nms(loc, conf, anchors, thr, 300, True, 0.6, 100, True)
    torch.cuda.synchronize()
    t = [N.timings_ms(i) for i in range(reps)]
    N.set_profiling(False)
    scan = sum(x[0] for x in t) / reps
    lvl = sum(x[1] for x in t) / reps
    nms = sum(x[2] for x in t) / reps
    print("%-28s scan %6.1f us (%5.2f TB/s)  level %5.1f us  nms %5.1f us" % (name, scan * 1e3, nbytes / scan / 1e9, lvl * 1e3, nms * 1e3), flush=True)


run("all equal (focal prior)", lambda *s: torch.full(s, 0.01, device="cuda"))
run("uniform, half above thr", lambda *s: torch.rand(*s, device="cuda") * 0.02)
run("sparse, 0.1% above thr", lambda *s: torch.rand(*s, device="cuda") * 0.01001)
run("nothing above thr", lambda *s: torch.rand(*s, device="cuda") * 0.009)
run("trained-like (1% above .05)", lambda *s: torch.rand(*s, device="cuda").pow(8), thr=0.05)
