"""Per-layer roofline report: joins a rocprofv3 kernel trace of bench.py with the layer geometry of the
SSD-MobileNetV2@512 plan (computed on CPU from the module structure)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
import torch.nn as nn

from ssds.core import config
from ssds.modeling import model_builder

trace = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
model = model_builder.create_model(cfg.MODEL).eval()
layers = []


def hook(m, inp, out):
    x = inp[0]
    layers.append((m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.groups, x.shape[2], x.shape[3],
                   out.shape[2], out.shape[3]))


hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, nn.Conv2d)]
with torch.no_grad():
    model(torch.rand(1, 3, 512, 512))
for h in hs:
    h.remove()
# heads: loc+conf of one level are one launch
n_heads = len(model.loc)
body, heads = layers[:-2 * n_heads], layers[-2 * n_heads:]
merged = list(body)
for i in range(n_heads):
    l, c = heads[2 * i], heads[2 * i + 1]
    merged.append((l[0], l[1] + c[1], l[2], l[3], 1, l[5], l[6], l[7], l[8]))

rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nms = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
step = rows[nms[len(nms) // 2 - 1] + 1: nms[len(nms) // 2] + 1]
convs = [r for r in step if any(k in r["Kernel_Name"] for k in ("conv_gemm", "dwconv", "conv_first"))]
assert len(convs) == len(merged), (len(convs), len(merged))
tot = 0
print("%3s %-22s %5s %5s %9s %8s %8s %7s" % ("#", "layer", "HxW", "k/s", "us", "GB/s", "TFLOP/s", "MB"))
for i, (r, L) in enumerate(zip(convs, merged)):
    cin, cout, k, s, g, h, w, ho, wo = L
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    bytes_ = 2 * B * (cin * h * w + cout * ho * wo) + 2 * cout * (cin // g) * k * k
    flops = 2 * B * ho * wo * cout * (cin // g) * k * k
    kind = "dw" if g > 1 else ("stem" if cin <= 4 else "gemm")
    print("%3d %-22s %5s %5s %9.1f %8.0f %8.1f %7.1f" % (i, "%s %d->%d" % (kind, cin, cout), "%dx%d" % (h, w),
          "%d/%d" % (k, s), us, bytes_ / us / 1e3, flops / us / 1e6, bytes_ / 1e6))
print("total conv us", tot)
others = [r for r in step if r not in convs]
for r in others:
    print("%9.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80]))
