"""What the box gives a dense bf16 GEMM on the shapes of the FPN-ResNet50@640 1x1 layers (batch 32) and of the 256 -> 256 tower
convolution, as a yardstick for csrc/ssdk_conv*.hip: torch.mm (hipBLASLt / rocBLAS, no epilogue, no residual) on random and on
ZERO operands -- the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS give-back), so the zero-operand rate is the
schedule's ceiling and the random-operand rate the power envelope's.  Usage: python tools/gemm_ceiling_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch


def t(fn, reps=20):
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


SHAPES = [("256>1024 @40x40", 51200, 256, 1024), ("1024>256 @40x40", 51200, 1024, 256), ("512>2048 @20x20", 12800, 512, 2048),
          ("2048>512 @20x20", 12800, 2048, 512), ("128>512 @80x80", 204800, 128, 512), ("512>128 @80x80", 204800, 512, 128),
          ("64>256 @160x160", 819200, 64, 256), ("256>64 @160x160", 819200, 256, 64), ("512>256 @80x80", 204800, 512, 256),
          ("1024>512 @40x40", 51200, 1024, 512), ("tower as a GEMM (K = 2304)", 204800, 2304, 256), ("8192^3", 8192, 8192, 8192)]
print("%-28s %9s %9s | %9s %9s | %s" % ("layer (M x K x N)", "random us", "TF/s", "zeros us", "TF/s", "HBM time of in + out at 8 TB/s (us)"))
for name, m, k, n in SHAPES:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us_r = t(lambda: torch.mm(a, b.t(), out=out))
    a0, b0 = torch.zeros_like(a), torch.zeros_like(b)
    us_z = t(lambda: torch.mm(a0, b0.t(), out=out))
    fl = 2.0 * m * k * n
    print("%-28s %9.1f %9.1f | %9.1f %9.1f | %7.1f" % (name, us_r, fl / us_r / 1e6, us_z, fl / us_z / 1e6, 2.0 * m * (k + n) / 8e12 * 1e6), flush=True)
