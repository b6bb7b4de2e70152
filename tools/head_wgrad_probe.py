"""Weight gradient of the SSD head pairs (SSD-MobileNetV2@512, batch 64, bf16; reference ssd.py:100-103 in the step of
pipeline_anchor_apex.py:103-130): the framework's convolution backward (MIOpen, weight + bias only, loc and conf separately)
against im2col + ssdk_pw_wgrad on the concatenated 504 channels (csrc/ssdk_pwtrain.hip).  Usage: python tools/head_wgrad_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch

from ssds import _native as N
from ssds.modeling.layers import pointwise as PW


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


B, dt = 64, torch.bfloat16
tot = [0.0, 0.0, 0.0]
print("%-14s | %10s | %10s %10s | max rel err of the native weight gradient" % ("level", "MIOpen us", "im2col us", "wgrad us"))
for cin, side in ((96, 32), (320, 16), (512, 8), (256, 4), (256, 2), (128, 1)):
    x = torch.randn(B, cin, side, side, device="cuda").to(dt)
    gl = torch.randn(B, 24, side, side, device="cuda").to(dt)
    gc = torch.randn(B, 480, side, side, device="cuda").to(dt)
    wl = torch.randn(24, cin, 3, 3, device="cuda").to(dt)
    wc = torch.randn(480, cin, 3, 3, device="cuda").to(dt)

    def lib():
        out = []
        for g, w in ((gl, wl), (gc, wc)):
            out.append(torch.ops.aten.convolution_backward(g, x, w, [int(w.shape[0])], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True]))
        return out

    sp = N.stream_ptr(x.device)
    hw = side * side
    kp = (cin * 9 + 7) // 8 * 8
    col = torch.empty((B, kp, hw), device="cuda", dtype=dt)
    g = torch.cat([gl, gc], 1).contiguous()
    need = int(N.lib.ssdk_pw_wgrad_workspace_bytes(B, 504, kp, hw))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    gw32 = torch.empty((504, kp), device="cuda", dtype=torch.float32)
    f_col = lambda: N.check(N.lib.ssdk_im2col3x3(x.data_ptr(), col.data_ptr(), B, cin, side, side, 1, 1, sp), "im2col")
    f_wg = lambda: N.check(N.lib.ssdk_pw_wgrad(g.data_ptr(), col.data_ptr(), gw32.data_ptr(), ws.data_ptr(), need, B, 504, kp, hw, 1, sp), "wgrad")
    a, b, c = t(lib), t(f_col), t(f_wg)
    ref = lib()
    want = torch.cat([ref[0][1].float(), ref[1][1].float()], 0)
    got = gw32[:, : cin * 9].reshape(504, cin, 3, 3)
    err = float((got - want).abs().max()) / float(want.abs().max())
    tot = [tot[0] + a, tot[1] + b, tot[2] + c]
    print("%4d ch @%-3d   | %10.1f | %10.1f %10.1f | %.3g" % (cin, side, a, b, c, err), flush=True)
print("total          | %10.1f | %10.1f %10.1f" % tuple(tot))
