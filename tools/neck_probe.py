"""Backbone (PyTorch-ROCm) vs neck+towers (recorded plan) time of an FPN / BiFPN config + per-op table of the plan.
Usage: python tools/neck_probe.py <cfg.yml> <batch>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
from ssds import _native as N
from ssds.core import config
from ssds.modeling import model_builder

cfg = config.cfg_from_file(sys.argv[1])
B = int(sys.argv[2])
torch.manual_seed(0)
model = model_builder.create_model(cfg.MODEL).eval().cuda().to(torch.bfloat16)
H, W = cfg.MODEL.IMAGE_SIZE
x = torch.rand(B, 3, H, W, device="cuda").to(torch.bfloat16)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    feats = model.backbone(x)
    print("features:", [tuple(f.shape) for f in feats])
    t_bb = timeit(lambda: model.backbone(x))
    t_all = timeit(lambda: model(x))
    print("backbone %.2f ms, full forward %.2f ms (neck+towers %.2f ms)" % (t_bb, t_all, t_all - t_bb))
    plan = list(model._neck_plans.values())[0]
    plan.ctx.set_op_profiling(True)  # per-op hipEvents live in the plan's own context
    model(x)
    torch.cuda.synchronize()
    t = plan.ctx.op_timings()
    plan.ctx.set_op_profiling(False)
    tot = {}
    for row, (kern, ms) in zip(plan.layer_table(), t):
        key = (row["name"], kern.replace("_kernel", ""))
        a = tot.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += row["flops"]
        a[3] += row["bytes"]
    for (name, kern), (cnt, ms, fl, by) in tot.items():
        print("%-34s %-14s x%-3d %8.1f us  %7.1f TF/s %7.0f GB/s" % (name, kern, cnt, ms * 1e3, fl / ms / 1e9, by / ms / 1e6))
    print("plan total %.2f ms, %d ops" % (sum(ms for _, ms in t), len(t)))
