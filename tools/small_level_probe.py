"""How far are the SMALL pyramid levels (1x1 / 2x2 maps: a few hundred values per image) of the recorded plan from the fp32
module, compared with PyTorch-ROCm executing the same module in the same 16-bit dtype -- as a DISTRIBUTION over inputs, not as
one draw (tests/test_gpu_nets.py small-level rule; reference ssd.py:42-74).

    python tools/small_level_probe.py [--cfg ssd_mobilenetv2_512.yml] [--batch 8] [--seeds 6] [--dtype bfloat16]

Per level and head: Pearson r of (plan, fp32) and (floor, fp32) per seed, their mean / min over the seeds, and the POOLED r
over all seeds' values (the statistic tests/test_gpu_nets.py::test_small_levels_pooled_over_inputs asserts).  Writes
gpurun_out/small_level_probe_<cfg>_<dtype>.txt."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch

from test_gpu_bench_sizes import _seeded_model
from test_gpu_nets import floor_runs, pooled_small_level_stats

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="ssd_mobilenetv2_512.yml")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--seeds", type=int, default=6)
ap.add_argument("--dtype", default="bfloat16")
args = ap.parse_args()

tdt = getattr(torch, args.dtype)
cpu_model, cfg = _seeded_model(args.cfg)
h, w = cfg.MODEL.IMAGE_SIZE
xs, wants = [], []
for s in range(args.seeds):
    g = torch.Generator().manual_seed(1000 + s)
    x = torch.rand((args.batch, 3, h, w), generator=g)
    with torch.no_grad():
        wl, wc = cpu_model(x)
    xs.append(x)
    wants.append({"loc": [t.clone() for t in wl], "conf": [t.clone() for t in wc]})
model = cpu_model.cuda().to(tdt)
plans, floors = [], []
for x in xs:
    xd = x.cuda().to(tdt)
    with torch.no_grad():
        loc, conf = model(xd)
    plans.append({"loc": [t.float().cpu() for t in loc], "conf": [t.float().cpu() for t in conf]})
    f = floor_runs(model, xd, runs=1)[0]
    floors.append({"loc": [t.float().cpu() for t in f["loc"]], "conf": [t.float().cpu() for t in f["conf"]]})
rows = pooled_small_level_stats(plans, floors, wants, small=1 << 62)  # every level, small or not
lines = ["%s %s batch %d x %d seeds" % (args.cfg, args.dtype, args.batch, args.seeds),
         "%-8s %7s | %s | %s | %s" % ("tensor", "values", "plan r: per-seed mean / min, pooled", "floor r: mean / min, pooled",
                                      "pooled median error plan / floor (centred rms units)")]
for r in rows:
    lines.append("%-8s %7d | %.4f / %.4f, %.4f | %.4f / %.4f, %.4f | %.4f / %.4f" % (
        r["tensor"], r["values"], r["plan_r_mean"], r["plan_r_min"], r["plan_r_pooled"], r["floor_r_mean"], r["floor_r_min"],
        r["floor_r_pooled"], r["plan_median_pooled"], r["floor_median_pooled"]))
text = "\n".join(lines)
print(text)
out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "small_level_probe_%s_%s.txt" % (os.path.splitext(args.cfg)[0], args.dtype)), "w") as fh:
        fh.write(text + "\n")
