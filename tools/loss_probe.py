"""Per-step time of the loss body of ModelWithLossBasic (reference pipeline_anchor_apex.py:43-72) on SSD-MobileNetV2@512-shaped
heads at batch 64: fused kernels (ssdk_match_loss / ssdk_match_multibox_loss) against the unfused torch ops, forward + backward
w.r.t. the heads.

    python tools/loss_probe.py [focal|multibox] [batch]"""
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch

from ssds.core import criterion
from ssds.modeling.layers import box
from ssds.pipeline.pipeline_anchor_ddp import ModelWithLossBasic


class Heads(torch.nn.Module):
    def __init__(self, loc, conf):
        super().__init__()
        self.loc = torch.nn.ParameterList([torch.nn.Parameter(t) for t in loc])
        self.conf = torch.nn.ParameterList([torch.nn.Parameter(t) for t in conf])

    def forward(self, images):
        return list(self.loc), list(self.conf)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "multibox"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    C, A = 81, 6
    sizes = OrderedDict([(16, 32), (32, 16), (64, 8), (128, 4), (256, 2), (512, 1)])
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in sizes)
    torch.manual_seed(0)
    loc = [torch.randn(B, A * 4, h, h, device="cuda", dtype=torch.bfloat16) * 0.2 for h in sizes.values()]
    conf = [torch.randn(B, A * C, h, h, device="cuda", dtype=torch.bfloat16) - 3 for h in sizes.values()]
    g = torch.Generator().manual_seed(1)
    G = 12
    xy = torch.rand(B, G, 2, generator=g) * 380
    wh = 24 + torch.rand(B, G, 2, generator=g) * 120
    lab = torch.randint(0, C, (B, G, 1), generator=g).float()
    targets = torch.cat([xy, wh, lab], -1).cuda()
    crit = criterion.MultiBoxLoss(3) if kind == "multibox" else criterion.FocalLoss()
    out = {}
    for fused in ("1", "0"):
        os.environ["SSDK_FUSED_LOSS"] = fused
        m = ModelWithLossBasic(Heads([t.clone() for t in loc], [t.clone() for t in conf]), crit, criterion.SmoothL1Loss(), C,
                               [0.5, 0.4], 0)

        def step():
            for p in m.parameters():
                p.grad = None
            c, l, _, _ = m(None, targets, anchors)
            (c + l).backward()
            return c, l

        for _ in range(3):
            c, l = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            c, l = step()
        torch.cuda.synchronize()
        out[fused] = ((time.perf_counter() - t0) / n * 1e3, float(c), float(l))
    print("%s, batch %d, 6 levels, %d classes: fused %.2f ms (cls %.4f loc %.4f) | unfused torch %.2f ms (cls %.4f loc %.4f)" % (
        kind, B, C, out["1"][0], out["1"][1], out["1"][2], out["0"][0], out["0"][1], out["0"][2]))


if __name__ == "__main__":
    main()
