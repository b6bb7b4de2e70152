"""Times the three depthwise 3x3 training passes (forward, input gradient, weight gradient) through the C-ABI on the
NCHW shapes of the SSD-MobileNetV2@300 training step, batch 64, bf16, and prints them beside their HBM time (the
tensors each pass has to read and write once, at 8 TB/s).  SSDK_DW_PLANE=0 selects the tiled kernels of
ssdk_dwtrain.hip, the default the whole-row kernels of ssdk_dwplane.hip.  Usage: python tools/dw_probe.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
from ssds import _native as N

# (channels, plane side, stride) of the depthwise layers of MobileNetV2 at 300 x 300 input
LAYERS = [(32, 150, 1), (96, 150, 2), (144, 75, 1), (144, 75, 2), (192, 38, 1), (192, 38, 1), (192, 38, 2), (384, 19, 1),
          (384, 19, 1), (384, 19, 1), (384, 19, 1), (576, 19, 1), (576, 19, 1), (576, 19, 2), (960, 10, 1), (960, 10, 1),
          (960, 10, 1)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def t(fn, reps=10):
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


dev = torch.device("cuda:0")
stream = N.stream_ptr(dev)
print("%-16s | %8s %8s %8s | %8s %8s %8s  (us; hbm = bytes / 8 TB/s)" % ("c @side /s", "fwd", "dgrad", "wgrad", "hbm fwd", "hbm dgr", "hbm wgr"))
tot = [0.0] * 6
seen = {}
for c, h, s in LAYERS:
    if (c, h, s) not in seen:
        ho = (h + 2 - 3) // s + 1
        x = torch.randn(B, c, h, h, device=dev, dtype=torch.bfloat16)
        w = torch.randn(c, 1, 3, 3, device=dev, dtype=torch.bfloat16)
        y = torch.empty(B, c, ho, ho, device=dev, dtype=torch.bfloat16)
        gy = torch.randn_like(y)
        gx = torch.empty_like(x)
        gw = torch.empty(c, 9, device=dev, dtype=torch.float32)
        need = int(N.lib.ssdk_dwconv_bwd_weight_workspace_bytes(B, c, h, h, s))
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        dt = N.dtype_code(x)
        r = [t(lambda: N.check(N.lib.ssdk_dwconv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, c, h, h, s, dt, stream), "fwd")),
             t(lambda: N.check(N.lib.ssdk_dwconv_bwd_data(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, c, h, h, s, dt, stream), "dgrad")),
             t(lambda: N.check(N.lib.ssdk_dwconv_bwd_weight(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), need, B, c, h, h, s, dt, stream), "wgrad"))]
        bx, by = x.numel() * 2, y.numel() * 2
        r += [(bx + by) / 8e6] * 3
        seen[(c, h, s)] = r
    r = seen[(c, h, s)]
    print("%-16s | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f" % (("%d @%d /%d" % (c, h, s),) + tuple(r)))
    tot = [a + b for a, b in zip(tot, r)]
print("%-16s | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f" % (("one step",) + tuple(tot)))
