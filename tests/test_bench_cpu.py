"""bench.py's forward check must be able to fail (Weak 2 / Next 1c): `judge_heads` is fed the statistics of
the driver's own bf16 line (loc floors at 0.5 - 0.8 of the signal) with a good plan, then with zeros, a constant, shuffled
values and noise of the right size in one head at a time.  CPU only, no kernels."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _logit(p):
    p = p.float().clamp(1e-7, 1 - 1e-7)
    return torch.log(p) - torch.log1p(-p)


def _case(floor_sigma):
    g = torch.Generator().manual_seed(17)
    ref_loc = [torch.randn(4, 24, 8, 8, generator=g) * 0.8, torch.randn(4, 24, 4, 4, generator=g) * 0.7]
    ref_conf = [torch.sigmoid(torch.randn(4, 480, 8, 8, generator=g) * 0.6 - 4), torch.sigmoid(torch.randn(4, 480, 4, 4, generator=g) * 0.5 - 4)]

    def noisy_loc(t):
        return t + floor_sigma * float(t.std()) * torch.randn(t.shape, generator=g)

    def noisy_conf(t):
        lg = _logit(t)
        return torch.sigmoid(lg + floor_sigma * float(lg.std()) * torch.randn(t.shape, generator=g))

    floor = ([noisy_loc(t) for t in ref_loc], [noisy_conf(t) for t in ref_conf])
    plan = ([noisy_loc(t) for t in ref_loc], [noisy_conf(t) for t in ref_conf])
    return g, ref_loc, ref_conf, floor, plan


@pytest.mark.parametrize("floor_sigma", [0.08, 0.5, 0.8])
def test_forward_check_passes_a_plan_at_the_floor_and_fails_dead_heads(floor_sigma):
    g, ref_loc, ref_conf, floor, plan = _case(floor_sigma)
    rows, worst, corr_ok = bench.judge_heads(plan[0], plan[1], floor[0], floor[1], ref_loc, ref_conf)
    assert worst <= 1.0 and corr_ok, rows
    prior = float(torch.sigmoid(torch.tensor(-4.0)))

    def perm(t):
        return t.flatten()[torch.randperm(t.numel(), generator=g)].view_as(t)

    for name, mk_loc, mk_conf in (
            ("zeros", torch.zeros_like, lambda t: torch.full_like(t, prior)),
            ("shuffled", perm, perm),
            ("noise", lambda t: torch.randn(t.shape, generator=g) * float(t.std()),
             lambda t: torch.sigmoid(torch.randn(t.shape, generator=g) * float(_logit(t).std()) - 4))):
        for head in ("loc", "conf"):
            for level in (0, 1):
                bl = [mk_loc(t) if (head == "loc" and i == level) else p for i, (t, p) in enumerate(zip(ref_loc, plan[0]))]
                bc = [mk_conf(t) if (head == "conf" and i == level) else p for i, (t, p) in enumerate(zip(ref_conf, plan[1]))]
                rows, worst, corr_ok = bench.judge_heads(bl, bc, floor[0], floor[1], ref_loc, ref_conf)
                assert not (worst <= 1.0 and corr_ok), (name, head, level, floor_sigma, rows)


def test_round4_rule_alone_admitted_a_zero_head():
    """Why the correlation rule exists: at the bf16 floors of the driver's line (0.5 - 0.8) the error rule alone passes zeros."""
    g, ref_loc, ref_conf, floor, plan = _case(0.8)
    zl = [torch.zeros_like(ref_loc[0]), plan[0][1]]
    rows, worst, corr_ok = bench.judge_heads(zl, plan[1], floor[0], floor[1], ref_loc, ref_conf)
    assert worst <= 1.0 and not corr_ok, rows
