"""Times the three GEMMs of a 1x1 convolution (forward, input gradient, weight gradient) on the NCHW shapes of the
SSD-MobileNetV2@512 training step, batch 64, bf16: MIOpen through torch's conv2d against the strided-batched library
GEMMs of ssds/modeling/layers/pointwise.py.  Usage: python tools/pw_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
import torch.nn.functional as F

SHAPES = [(16, 96, 256), (96, 24, 128), (24, 144, 128), (144, 24, 128), (144, 32, 64), (32, 192, 64), (192, 32, 64), (192, 64, 32),
          (64, 384, 32), (384, 64, 32), (384, 96, 32), (96, 576, 32), (576, 96, 32), (576, 160, 16), (160, 960, 16),
          (960, 160, 16), (960, 320, 16), (320, 1280, 16), (1280, 256, 16), (512, 128, 8)]
B = 64


def t(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%-18s | %8s %8s %8s | %8s %8s %8s %8s" % ("cin>cout @hw", "conv fwd", "dgrad", "wgrad", "mm fwd", "mm dgrad", "bmm wgr", "mm-k wgr"))
tot = [0.0] * 7
for cin, cout, hw in SHAPES:
    x = torch.randn(B, cin, hw, hw, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device="cuda", dtype=torch.bfloat16) * 0.1
    gy = torch.randn(B, cout, hw, hw, device="cuda", dtype=torch.bfloat16)
    x3, w2, gy3 = x.view(B, cin, -1), w.view(cout, cin), gy.view(B, cout, -1)
    r = [t(lambda: F.conv2d(x, w)),
         t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False))),
         t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))),
         t(lambda: torch.matmul(w2, x3)),
         t(lambda: torch.matmul(w2.t(), gy3)),
         t(lambda: torch.bmm(gy3, x3.transpose(1, 2), out_dtype=torch.float32).sum(0)),
         t(lambda: torch.matmul(gy3.transpose(0, 1).reshape(cout, -1), x3.transpose(0, 1).reshape(cin, -1).t()))]
    tot = [a + b for a, b in zip(tot, r)]
    print("%4d>%-4d @%-3d     | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f" % ((cin, cout, hw) + tuple(r)), flush=True)
print("%-18s | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f" % (("total us",) + tuple(tot)))
