"""Which framework element-wise kernels are left in the training step, and who calls them: torch.profiler over two steps of
tools/bench_train.py's setup, the fill / copy / add / reduce ops grouped by the innermost ssds / tools source line.
Usage: python tools/train_glue_probe.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
from torch.profiler import ProfilerActivity, profile

from ssds.core import config
from ssds.modeling import model_builder
from ssds.pipeline.pipeline_anchor_ddp import train_step
from ssds.utils.train_ddp import Solver, SyntheticDetectionLoader

cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
cfg.TRAIN.BATCH_SIZE = 64
cfg.EXP_DIR = "/tmp/ssdk_bench_train"
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
solver = Solver(cfg, 0, dev)
mwl = solver.wrap()
mwl.train()
inner = mwl.module.model if hasattr(mwl, "module") else mwl.model
anchors = model_builder.create_anchors(cfg.MODEL, inner, cfg.MODEL.IMAGE_SIZE)
images, targets = SyntheticDetectionLoader(64, cfg.MODEL.IMAGE_SIZE, cfg.MODEL.NUM_CLASSES, 1, dev, seed=1234).batch()
for _ in range(5):
    train_step(mwl, images, targets, anchors, solver.optimizer)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(2):
        train_step(mwl, images, targets, anchors, solver.optimizer)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::cat", "aten::clone",
        "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros", "aten::zeros_like")
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in ("aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::cat"):
        continue
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if not dt:
        continue
    where = "?"
    for fr in ev.stack or []:
        if ("ssds" in fr or "tools/" in fr) and "profiler" not in fr:
            where = fr.split("ssds.pytorch_amd/")[-1][:110]
            break
    else:
        # no python frame (the autograd engine's thread): the enclosing autograd node and the operand shapes say who it is
        par, chain = ev.cpu_parent, []
        while par is not None and len(chain) < 3:
            chain.append(par.name[:40])
            par = par.cpu_parent
        where = " < ".join(chain) + "  " + str(ev.input_shapes)[:70]
    k = (ev.name, where)
    acc[k][0] += 1
    acc[k][1] += dt
for (name, where), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.1f us x%-4d %-12s %s" % (us / 2, n // 2, name, where))
