"""Owner test of the kernel-family switches (docs/SWITCHES.md): every `SSDK_*=0` that takes a kernel family out of the
SSD-MobileNetV2@512 plan must leave the heads where they were -- another kernel computes the same layer.  The switches are read
once per process (static), so each one runs in its own subprocess: the parent records the default heads, a child rebuilds the
same seeded model with ONE switch set and compares (reference forward: ssd.py:42-74 over mobilenet.py:180-192)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switch -> (value, kernel-name fragment that must DISAPPEAR from the plan's op list with it; None: the same kernels, another path)
SWITCHES = {
    "SSDK_MB_FLOW": ("0", "mbflow"), "SSDK_MBK": ("0", "mbk"), "SSDK_MB_SPLIT": ("0", "mbsplit"), "SSDK_FLOW_PAIR": ("0", None),
    "SSDK_STEM_DWORD": ("0", None), "SSDK_CONV3X3_SHORT": ("0", "conv3x3_short"), "SSDK_CONV3X3_HALO": ("0", "conv3x3_halo"),
    "SSDK_CONV_SMALLMAP": ("0", "conv_smallmap"), "SSDK_CONV_SMALLMAP_GROUP": ("0", "conv_smallmap_group"),
    "SSDK_CONV_SMALLMAP_KW": ("1", None), "SSDK_S3_WIDE": ("0", None), "SSDK_WFRAG": ("0", None), "SSDK_XPAIR": ("0", "xpair"),
    "SSDK_FUSED_BLOCK": ("0", "mb"), "SSDK_HEAD_BALANCE": ("0", "conv_smallmap_group"), "SSDK_SPLITK": ("0", None),
    "SSDK_CONV_WAVE": ("0", None), "SSDK_HALO_PERSIST": ("0", None),
}
CHILD = r'''
import os, sys
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from test_gpu_bench_sizes import _seeded_model
cpu_model, cfg = _seeded_model("ssd_mobilenetv2_512.yml")
g = torch.Generator().manual_seed(99)
x = torch.rand((64, 3, 512, 512), generator=g).cuda().to(torch.bfloat16)
model = cpu_model.cuda().to(torch.bfloat16)
with torch.no_grad():
    loc, conf = model(x)
plan = model._plan(x)
plan.ctx.set_op_profiling(True)
with torch.no_grad():
    model(x)
torch.cuda.synchronize()
names = [k for k, _ in plan.ctx.op_timings()]
torch.save({"loc": [t.float().cpu() for t in loc], "conf": [t.float().cpu() for t in conf], "kernels": names}, OUT)
'''


def _run(env, out):
    code = "ROOT = %r\nOUT = %r\n" % (ROOT, out) + CHILD
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (env, r.stdout[-1500:], r.stderr[-1500:])


@pytest.fixture(scope="module")
def default_heads(tmp_path_factory):
    import torch

    out = str(tmp_path_factory.mktemp("switches") / "default.pt")
    _run({}, out)
    return torch.load(out)


@pytest.mark.parametrize("switch", sorted(SWITCHES))
def test_a_switched_off_kernel_family_leaves_the_heads_where_they_were(switch, default_heads, tmp_path):
    import torch

    value, gone = SWITCHES[switch]
    out = str(tmp_path / "alt.pt")
    _run({switch: value}, out)
    alt = torch.load(out)
    if gone is not None:
        had = [k for k in default_heads["kernels"] if gone in k]
        assert had, "%s is not part of the default plan: the switch has nothing to take out here" % gone
        still = [k for k in alt["kernels"] if gone in k and (gone != "mb" or k.startswith("mb"))]
        assert len(still) < len(had), (switch, still)
    for tag in ("loc", "conf"):
        for i, (a, b) in enumerate(zip(alt[tag], default_heads[tag])):
            if tag == "conf":  # compare logits: a sigmoid output near 0.01 hides its logit
                a, b = (torch.log(t.clamp(1e-7, 1 - 1e-7)) - torch.log1p(-t.clamp(1e-7, 1 - 1e-7)) for t in (a, b))
            if b.numel() < 4096:
                continue  # (a 1x1 / 2x2 level is one draw of the noise: tests/test_gpu_nets.py judges those pooled)
            ac, bc = a - a.mean(), b - b.mean()
            rms = float(bc.pow(2).mean().sqrt())
            r = float((ac * bc).mean()) / max(float(ac.pow(2).mean().sqrt()) * rms, 1e-12)
            # two bf16 executions of the same 60-layer network through DIFFERENT kernels are two draws of its rounding noise
            # (0.1 - 0.45 rms from fp32 each, tests/test_gpu_nets.py): they correlate like either does with fp32; a wrong layer
            # decorrelates everything behind it (r ~ 0)
            assert r >= 0.6 and float((a - b).abs().median()) <= 0.7 * rms, (switch, tag, i, r, float((a - b).abs().median()), rms)
