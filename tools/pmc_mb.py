"""Runs the fused blocks of SSD-MobileNetV2@512 (B=64) a few times each -- target for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
from ssds.core import config
from ssds.modeling import model_builder
cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
torch.manual_seed(0)
model = model_builder.create_model(cfg.MODEL).eval().cuda().to(torch.bfloat16)
x = torch.rand(int(os.environ.get("B", 64)), 3, 512, 512, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", 3))):
        model(x)
torch.cuda.synchronize()
