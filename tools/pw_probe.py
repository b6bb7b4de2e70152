"""Times the three passes of a 1x1 convolution (forward, input gradient, weight gradient) on the NCHW shapes of the
SSD-MobileNetV2@512 training step, batch 64, bf16 (reference: the pointwise convolutions behind mobilenet.py:56, 78 and
basic_layers.py:40-57 in the step of pipeline_anchor_apex.py:103-130): the hand-written kernels of csrc/ssdk_pwtrain.hip
(ssdk_pw_forward / ssdk_pw_wgrad) against the strided-batched library GEMMs rounds 2-5 used (torch.matmul / torch.bmm), with
the fraction of the 8 TB/s HBM roof each pass reaches on its algorithmic bytes (in + out, 2 bytes per element).
Usage: python tools/pw_probe.py [count per shape = the number of such layers in the network]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch

from ssds import _native as N

# (cin, cout, map side, layers of that shape in SSD-MobileNetV2@512)
SHAPES = [(32, 16, 256, 1), (16, 96, 256, 1), (96, 24, 128, 1), (24, 144, 128, 2), (144, 24, 128, 1), (144, 32, 64, 1),
          (32, 192, 64, 3), (192, 32, 64, 2), (192, 64, 32, 1), (64, 384, 32, 4), (384, 64, 32, 3), (384, 96, 32, 1),
          (96, 576, 32, 3), (576, 96, 32, 2), (576, 160, 16, 1), (160, 960, 16, 3), (960, 160, 16, 2), (960, 320, 16, 1),
          (320, 256, 16, 1), (512, 128, 8, 1), (256, 128, 4, 1), (256, 64, 2, 1)]
B = 64
PEAK = 8.0e12


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


sp = N.stream_ptr(torch.device("cuda", 0))
print("%-16s %2s | %27s | %27s | %s" % ("cin>cout @hw", "n", "ssdk  fwd   dgrad   wgrad (us)", "library fwd dgrad wgrad (us)", "ssdk frac of 8 TB/s  fwd dgrad wgrad"))
tot = [0.0] * 6
ideal = 0.0
for cin, cout, side, cnt in SHAPES:
    hw = side * side
    x = torch.randn(B, cin, side, side, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(cout, cin, device="cuda") * 0.1).to(torch.bfloat16)
    wt = w.t().contiguous()
    gy = torch.randn(B, cout, side, side, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(gy)
    gx = torch.empty_like(x)
    need = int(N.lib.ssdk_pw_wgrad_workspace_bytes(B, cout, cin, hw))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    gw = torch.empty(cout, cin, device="cuda")
    x3, gy3 = x.view(B, cin, -1), gy.view(B, cout, -1)
    sneed = int(N.lib.ssdk_pw_stats_workspace_bytes(B, cin, cout, hw))
    sws = torch.empty(sneed + 16, dtype=torch.uint8, device="cuda")
    sums = torch.empty(cout, 2, device="cuda")
    t_stats = t(lambda: N.check(N.lib.ssdk_pw_forward_stats(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), sums.data_ptr(),
                                                            (sws.data_ptr() + 15) & ~15, sneed, B, cin, cout, hw, 1, sp), "fs"))
    r = [t(lambda: N.check(N.lib.ssdk_pw_forward(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, cin, cout, hw, 1, sp), "f")),
         t(lambda: N.check(N.lib.ssdk_pw_forward(gy.data_ptr(), wt.data_ptr(), None, gx.data_ptr(), B, cout, cin, hw, 1, sp), "d")),
         t(lambda: N.check(N.lib.ssdk_pw_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), ws.data_ptr(), need, B, cout, cin, hw, 1, sp), "w")),
         t(lambda: torch.matmul(w, x3)),
         t(lambda: torch.matmul(wt, gy3)),
         t(lambda: torch.bmm(gy3, x3.transpose(1, 2), out_dtype=torch.float32).sum(0))]
    byt = 2.0 * B * hw * (cin + cout)
    fr = [byt / (v * 1e-6) / PEAK for v in r[:3]]
    tot = [a + b * cnt for a, b in zip(tot, r)]
    ideal += cnt * byt / PEAK * 1e6
    tot_stats = globals().get("tot_stats", 0.0) + t_stats * cnt
    print("%4d>%-4d @%-3d   %2d | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f | %5.2f %5.2f %5.2f | fwd + BN statistics %8.1f" % (
        (cin, cout, side, cnt) + tuple(r) + tuple(fr) + (t_stats,)), flush=True)
print("forward with the BatchNorm statistics (ssdk_pw_forward_stats), layers counted: %.1f us" % tot_stats)
print("%-19s | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f |  (us per step, layers counted; HBM time of one pass at 8 TB/s: %.1f us)"
      % (("total",) + tuple(tot) + (ideal,)))
