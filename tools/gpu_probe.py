"""Quick per-stage timing probe on the GPU box (not part of the test-suite)."""
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import numpy as np
import torch

from oracle import box_oracle as O
from ssds import _native as N
from ssds.modeling.layers import box
from ssds.modeling.layers.decoder import Decoder

print(N.device_info())
B = int(os.environ.get("B", 64))
A, C = 6, 80
maps, strides = [32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512]
dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[os.environ.get("DT", "bf16")]
torch.manual_seed(0)
conf = [torch.sigmoid(torch.randn(B, A * C, m, m, device="cuda") * 1.5 - 4.6).to(dt) for m in maps]
loc = [(torch.randn(B, A * 4, m, m, device="cuda") * 0.5).to(dt) for m in maps]
anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828]))) for s in strides)
dec = Decoder(0.01, 0.6, 100, 300, True, True)


def timeit(fn, n=50, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


bytes_alg = sum(c.numel() * c.element_size() + l.numel() * l.element_size() for c, l in zip(conf, loc)) + B * (2 * 24 * 1800 + 2400)
for tpu in os.environ.get("TPUS", "0,8,16,32,64").split(","):
    os.environ["SSDK_TILES_PER_UNIT"] = tpu
    t = timeit(lambda: dec(loc, conf, anchors))
    print("tpu=%s decode+nms %.1f us  -> %.2f TB/s algorithmic, %.0f img/s" % (tpu, t, bytes_alg / t / 1e6, B / t * 1e6))
os.environ["SSDK_TILES_PER_UNIT"] = "0"
t0 = timeit(lambda: box.decode(conf[0], loc[0], 16, 0.01, 300, anchors[16], True))
print("decode level0 only %.1f us (%.2f TB/s)" % (t0, conf[0].numel() * conf[0].element_size() / t0 / 1e6))
(_, _, _), mid = box.decode_nms(loc, conf, anchors, 0.01, 300, True, 0.6, 100, True, return_mid=True)
t1 = timeit(lambda: box.nms(mid[0], mid[1], mid[2], 0.6, 100, True))
print("nms only %.1f us" % t1)
# copy roofline probe
x = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
y = torch.empty_like(x)
tc = timeit(lambda: y.copy_(x), n=20)
print("copy 256MiB: %.1f us -> %.2f TB/s (r+w)" % (tc, 2 * x.numel() * 4 / tc / 1e6))
