import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/ssds.pytorch_amd"]
import torch
from ssds.core import config
from ssds.modeling import model_builder
cfg = config.cfg_from_file("/root/repo/experiments/cfgs/ssd_mobilenetv2_512.yml")
torch.manual_seed(0)
m = model_builder.create_model(cfg.MODEL).eval().cuda().to(torch.bfloat16)
x = torch.rand(64, 3, 512, 512, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    m(x); torch.cuda.synchronize()
    os.environ["X"] = "1"
    m(x); torch.cuda.synchronize()
