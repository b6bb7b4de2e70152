"""Layer-by-layer audit of a recorded plan: every op of the plan is launched on its own and its output is compared with
the SAME op computed in fp32 by PyTorch-ROCm on the op's ACTUAL input (the plan's own activation, so the errors of earlier
layers do not count): what is left is the rounding of that one kernel.  Then the plan's final heads, PyTorch-ROCm's
16-bit execution of the module (the "floor" of tests/test_gpu_nets.py) and the fp32 module are compared end to end, the
class heads both as probabilities and as logits.

    python tools/plan_trace.py <cfg.yml> <batch> <bfloat16|float16> [seed]

Written to answer round 3's open question (FPN-ResNet50@640, level-0 class head: p99.9 of |err| / rms 7.1 for the plan
against 3.6 for PyTorch-ROCm): which kernel, if any, produces a heavier tail than its rounding explains."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import torch.nn.functional as F

from ssds import _native as N
from ssds.modeling.layers import fused_conv as FC


def stats(got, want):
    """(median, p99.9, max) of |got - want| / rms(want)."""
    g, w = got.float(), want.float()
    rms = max(float(w.pow(2).mean().sqrt()), 1e-12)
    e = ((g - w).abs() / rms).flatten()
    if e.numel() > 20_000_000:  # kthvalue on a sample is enough for a report
        e = e[torch.randint(0, e.numel(), (20_000_000,), device=e.device)]
    k = max(int(e.numel() * 0.999), 1)
    return float(e.median()), float(e.kthvalue(k).values), float(e.max())


def main():
    cfg_name, batch, dtype = sys.argv[1], int(sys.argv[2]), getattr(torch, sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 321
    from test_gpu_bench_sizes import _seeded_model

    torch.backends.cudnn.allow_tf32 = False
    cpu_model, cfg = _seeded_model(os.path.basename(cfg_name), seed)
    h, w = cfg.MODEL.IMAGE_SIZE
    g = torch.Generator().manual_seed(99)
    x = torch.rand((batch, 3, h, w), generator=g)
    ref32 = cpu_model.cuda()
    with torch.no_grad():
        wl, wc = ref32(x.cuda())  # fp32 module on the device (MIOpen fp32)
    wl, wc = [t.float().cpu() for t in wl], [t.float().cpu() for t in wc]
    model = ref32.to(dtype)
    xd = x.cuda().to(dtype)
    with torch.no_grad():
        loc, conf = model(xd)
    plan = model._plan(xd) if hasattr(model, "_plan") else next(iter(model._neck_plans.values()))
    assert not isinstance(plan, str), plan
    import planaudit

    audit = planaudit.PlanAudit(plan, [xd])
    rows = audit.run()  # every op on its own against fp32 of that op on its own input (tests/planaudit.py)
    outs = audit.outs
    print(planaudit.format_rows(rows))
    bad = planaudit.failures(rows, dtype)
    print("\n%d ops, %d outside their bar%s" % (len(rows), len(bad), (":\n  " + "\n  ".join(bad)) if bad else ""))
    torch.cuda.synchronize()

    # ---- end to end: plan vs fp32, PyTorch-ROCm 16-bit vs fp32; class heads also as logits -------------------------------
    os.environ["SSDK_FUSED_CONV"] = "0"
    with torch.no_grad():
        tl, tc = model(xd)
    del os.environ["SSDK_FUSED_CONV"]

    def logit(p):
        p = p.float().clamp(1e-7, 1 - 1e-7)
        return torch.log(p) - torch.log1p(-p)

    print("\nend to end, |err| / rms(fp32 reference): median / p99.9 / max   (plan | PyTorch-ROCm %s)" % sys.argv[3])
    for i in range(len(wl)):
        print("loc%d           plan %8.4f %8.4f %8.4f | torch %8.4f %8.4f %8.4f" % ((i,) + stats(outs[0][i].cpu(), wl[i]) + stats(tl[i].cpu(), wl[i])))
    for i in range(len(wc)):
        print("conf%d (prob)   plan %8.4f %8.4f %8.4f | torch %8.4f %8.4f %8.4f" % ((i,) + stats(outs[1][i].cpu(), wc[i]) + stats(tc[i].cpu(), wc[i])))
        print("conf%d (logit)  plan %8.4f %8.4f %8.4f | torch %8.4f %8.4f %8.4f" % (
            (i,) + stats(logit(outs[1][i].cpu()), logit(wc[i])) + stats(logit(tc[i].cpu()), logit(wc[i]))))
        # where the big probability errors sit: elements whose reference probability is above 0.1
        hot = wc[i] > 0.1
        if int(hot.sum()) > 0:
            ep = (outs[1][i].cpu().float() - wc[i]).abs()[hot]
            et = (tc[i].cpu().float() - wc[i]).abs()[hot]
            print("    %d elements with p > 0.1: mean |err| plan %.5f torch %.5f; share of the plan's 0.1%% largest errors among them: %.2f"
                  % (int(hot.sum()), float(ep.mean()), float(et.mean()),
                     float(((outs[1][i].cpu().float() - wc[i]).abs() >= (outs[1][i].cpu().float() - wc[i]).abs().flatten().kthvalue(
                         max(int(wc[i].numel() * 0.999), 1)).values)[hot].float().sum() / max(1.0, wc[i].numel() * 0.001))))


if __name__ == "__main__":
    main()
