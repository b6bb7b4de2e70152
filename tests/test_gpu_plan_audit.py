"""Every op of the recorded plans of BASELINE configs 2 / 3 / 5, at the bench's batch sizes and dtypes, against fp32 of
that layer on the op's own input (tests/planaudit.py).  This is the forward check that discriminates in bf16: the end-to-end
comparisons of test_gpu_nets.py / test_gpu_bench_sizes.py are bounded by the noise floor of a deep random-weight network
executed in 16 bits (0.5 - 0.9 RMS at the deep levels, for PyTorch-ROCm as much as for the plan), under which a dead or
all-zero head can hide; per op, every kernel must sit at its own rounding level (~1e-3 of the layer's RMS in bf16) and a
layer whose fp32 output is > 95 % zeros fails the test instead of passing it.

Reference: ssd.py:42-74, fpn.py:10-18, 58-101, bifpn.py:30-63, 104-142, mobilenet.py:180-192, resnet.py:41-56,
regnet.py:270-282 through this repository's modules, whose wiring the reference-class fixtures pin (test_nets_golden.py)."""
import os

import pytest

import planaudit
from test_gpu_bench_sizes import _seeded_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg_name,batch,dtype", [
    ("ssd_mobilenetv2_512.yml", 64, "bfloat16"),   # BASELINE config 2
    ("fpn_resnet50_640.yml", 32, "bfloat16"),      # BASELINE config 3
    ("bifpn_regnetx008_896.yml", 16, "float16"),   # BASELINE config 5
])
def test_every_op_of_the_bench_plan_sits_at_its_rounding_level(cfg_name, batch, dtype):
    import torch

    tdt = getattr(torch, dtype)
    cpu_model, cfg = _seeded_model(cfg_name)
    h, w = cfg.MODEL.IMAGE_SIZE
    g = torch.Generator().manual_seed(99)
    x = torch.rand((batch, 3, h, w), generator=g)
    model = cpu_model.cuda().to(tdt)
    xd = x.cuda().to(tdt)
    with torch.no_grad():
        loc, conf = model(xd)  # records the plan at this shape
    torch.cuda.synchronize()
    plan = model._plan(xd) if hasattr(model, "_plan") else next(iter(model._neck_plans.values()))
    assert not isinstance(plan, str), plan
    audit = planaudit.PlanAudit(plan, [xd])
    rows = audit.run()
    text = planaudit.format_rows(rows)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "plan_audit_%s_%s.txt" % (os.path.splitext(cfg_name)[0], dtype)), "w") as f:
            f.write(text + "\n")
    assert len(rows) == len(plan.layers), "every op of the plan is audited"
    bad = planaudit.failures(rows, tdt)
    assert not bad, "\n".join(bad)
    # the op-by-op execution above and the plan's own single call (grouped small-map launches, side-stream chains) must agree
    # on the heads: the audit has then covered the kernels the timed forward runs
    one_by_one = [t.clone() for t in audit.outs[0] + audit.outs[1]]
    with torch.no_grad():
        loc2, conf2 = model(xd)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(one_by_one, tuple(loc2) + tuple(conf2))):
        d = float((a.float() - b.float()).abs().max())
        assert torch.equal(a, b), "head tensor %d: op-by-op and whole-plan execution differ by %.3g" % (i, d)
    # the class heads are alive (a dead tower would make every comparison of this file vacuous)
    for i, c in enumerate(conf2):
        p = c.float().clamp(1e-7, 1 - 1e-7)
        lg = torch.log(p) - torch.log1p(-p)
        assert float(lg.std()) > 0.3, "conf level %d: logit std %.3f" % (i, float(lg.std()))


def test_the_audit_itself_can_fail():
    """A zeroed or wrongly wired op is flagged: the statistics of a zero output (median 0.67, max > 3 of the RMS) and of an
    output with two channels swapped are far outside every bar; > 95 % zeros in the reference is flagged as a dead layer."""
    import torch

    torch.manual_seed(5)
    want = torch.randn(2, 32, 16, 16, device="cuda")
    ok = (want.to(torch.bfloat16)).float()
    med, p999, mx, zeros = planaudit.stats(ok, want)
    row = dict(index=0, name="x", kernel="k", kind="default", median=med, p999=p999, max=mx, zeros=zeros)
    assert not planaudit.failures([row], torch.bfloat16), row
    for name, bad in (("zeros", torch.zeros_like(want)), ("swapped", want[:, torch.tensor([1, 0] + list(range(2, 32)))]),
                      ("noise", torch.randn_like(want))):
        med, p999, mx, zeros = planaudit.stats(bad, want)
        row = dict(index=0, name=name, kernel="k", kind="default", median=med, p999=p999, max=mx, zeros=zeros)
        assert planaudit.failures([row], torch.bfloat16), (name, row)
    dead = torch.relu(want - 3.0)
    med, p999, mx, zeros = planaudit.stats(dead.to(torch.bfloat16).float(), dead)
    assert zeros > 0.95
    row = dict(index=0, name="dead", kernel="k", kind="default", median=med, p999=p999, max=mx, zeros=zeros)
    assert any("dead layer" in s for s in planaudit.failures([row], torch.bfloat16))
