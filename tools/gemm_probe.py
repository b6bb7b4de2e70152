"""Times the MFMA convolution kernel on the dense layers that matter (SSD heads of SSD-MobileNetV2@512 B=64,
FPN tower of FPN-ResNet50@640 B=32) and checks each against torch conv2d.  Usage: python tools/gemm_probe.py [names]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
import torch.nn as nn
import torch.nn.functional as F

from ssds.modeling.layers import fused_conv as FC

# name: (B, Cin, H, W, Cout, k, stride, mode)   mode: head = NCHW split loc|conf + sigmoid, relu = NHWC BN-less ReLU
SHAPES = {
    "head_L0": (64, 96, 32, 32, 504, 3, 1, "head"),
    "head_L1": (64, 320, 16, 16, 504, 3, 1, "head"),
    "head_L2": (64, 512, 8, 8, 504, 3, 1, "head"),
    "head_L3": (64, 256, 4, 4, 504, 3, 1, "head"),
    "head_L4": (64, 256, 2, 2, 504, 3, 1, "head"),
    "head_L5": (64, 128, 1, 1, 504, 3, 1, "head"),
    "tower_P3": (32, 256, 80, 80, 256, 3, 1, "relu"),
    "tower_P4": (32, 256, 40, 40, 256, 3, 1, "relu"),
    "tower_P5": (32, 256, 20, 20, 256, 3, 1, "relu"),
    "tower_cls": (32, 256, 80, 80, 720, 3, 1, "relu"),
    "pw_320_1280": (64, 320, 16, 16, 1280, 1, 1, "relu"),
    "extras_1x1": (64, 320, 16, 16, 256, 1, 1, "relu"),
    "extras_3x3s2": (64, 256, 16, 16, 512, 3, 2, "relu"),
    "extras_3x3s2_b8": (8, 256, 16, 16, 512, 3, 2, "relu"),   # 16 workgroups: the same k-loop with an L2-resident operand set
    "extras_3x3s2_b2": (2, 256, 16, 16, 512, 3, 2, "relu"),   # 4 workgroups
}


def run(name, reps=20, check=True):
    B, Cin, H, W, Cout, k, s, mode = SHAPES[name]
    torch.manual_seed(0)
    conv = nn.Conv2d(Cin, Cout, k, s, k // 2, bias=True).cuda()
    with torch.no_grad():
        conv.weight.mul_(0.5)
    x = torch.randn(B, Cin, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if os.environ.get("SSDK_PROBE_ZERO") == "1":  # zero operands: what the schedule gives when the power budget does not bind
        x.zero_()
        with torch.no_grad():
            conv.weight.zero_()
    pack = FC.ConvPack(conv, None, "none", torch.bfloat16)
    pack.w, pack.bias = pack.w.cuda(), pack.bias.cuda()
    if mode == "head":
        f = lambda: FC.conv_native(x, pack, act="none", nchw_out=True, split=24, act2="sigmoid")
    else:
        f = lambda: FC.conv_native(x, pack, act="relu")
    out = f()
    if check:
        ref = F.conv2d(x.float(), conv.weight.to(torch.bfloat16).float(), conv.bias.float(), s, k // 2)
        if mode == "head":
            got = torch.cat([out[0].float(), out[1].float()], 1)
            ref = torch.cat([ref[:, :24], torch.sigmoid(ref[:, 24:])], 1)
        else:
            got, ref = out.float(), torch.relu(ref)
        err = (got - ref).abs().max().item()
        scale = ref.abs().max().item()
    else:
        err = scale = float("nan")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    print("%-14s M=%7d N=%4d K=%5d  %8.1f us  %7.1f TF/s  (%4.1f%% of 2.5 PF)  max|err|=%.3g (ref max %.3g)" % (
        name, B * Ho * Wo, Cout, Cin * k * k, us, flops / us / 1e6, flops / us / 1e6 / 25.0, err, scale), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(SHAPES)
    for n in names:
        run(n)
