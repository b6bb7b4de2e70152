"""The noise-floor comparison of tests/test_gpu_nets.py must be able to FAIL (Weak 1-3).  CPU only: the rule
is exercised on synthetic tensors with the statistics of a deep pyramid level."""
import pytest

from test_gpu_nets import _check_against_floor, _logit


def test_a_zero_or_shuffled_head_fails_the_floor_check():
    """The floor check must be able to fail (Weak 2): a head that writes zeros, a constant, the right values
    in the wrong order, or noise of the right size is rejected in bf16 although its median error stays under the cap; the
    floor itself and a bf16 rounding of the reference pass.  (No kernel runs here: synthetic tensors of a deep level's
    statistics -- floor at 0.5 sigma like PyTorch-ROCm on MobileNetV2's level 3.)"""
    import torch

    g = torch.Generator().manual_seed(3)
    want = {"loc": [torch.randn(4, 24, 8, 8, generator=g)], "conf": [torch.sigmoid(torch.randn(4, 480, 8, 8, generator=g) * 0.6 - 4)]}

    def noisy(t, sigma, logit):
        if logit:
            return torch.sigmoid(_logit(t) + sigma * 0.6 * torch.randn(t.shape, generator=g))
        return t + sigma * torch.randn(t.shape, generator=g)

    floor = {"loc": [noisy(want["loc"][0], 0.5, False)], "conf": [noisy(want["conf"][0], 0.5, True)]}
    good = {"loc": [noisy(want["loc"][0], 0.5, False)], "conf": [noisy(want["conf"][0], 0.5, True)]}
    _check_against_floor(good, [floor], want, "synthetic good", "bfloat16")
    prior = torch.sigmoid(torch.tensor(-4.0))
    for name, loc, conf in (
            ("zeros", torch.zeros_like(want["loc"][0]), torch.full_like(want["conf"][0], float(prior))),
            ("shuffled", want["loc"][0].flatten()[torch.randperm(want["loc"][0].numel(), generator=g)].view_as(want["loc"][0]),
             want["conf"][0].flatten()[torch.randperm(want["conf"][0].numel(), generator=g)].view_as(want["conf"][0])),
            ("noise", torch.randn(want["loc"][0].shape, generator=g), noisy(torch.full_like(want["conf"][0], float(prior)), 1.0, True))):
        for tag in ("loc", "conf"):
            bad = {"loc": [loc if tag == "loc" else good["loc"][0]], "conf": [conf if tag == "conf" else good["conf"][0]]}
            with pytest.raises(AssertionError):
                _check_against_floor(bad, [floor], want, "synthetic %s %s" % (name, tag), "bfloat16")


def test_the_pooled_small_level_rule_can_fail_and_is_at_full_strength():
    """check_small_levels_pooled (round 6: the 1x1 / 2x2 levels judged over many input draws instead of a weaker rule on one
    draw): a plan as noisy as the floor passes; a plan with twice the floor's noise level (correlation 0.45 instead of 0.7 --
    what round 5's "half the floor's r" rule accepted), zeros and a shuffled level fail."""
    import torch
    from test_gpu_nets import check_small_levels_pooled

    g = torch.Generator().manual_seed(5)
    draws = 25
    wants = [{"loc": [torch.randn(2, 24, 1, 1, generator=g)], "conf": [torch.sigmoid(torch.randn(2, 30, 1, 1, generator=g) * 0.6 - 4)]}
             for _ in range(draws)]

    def noisy(sigma):
        return [{"loc": [w["loc"][0] + sigma * torch.randn(w["loc"][0].shape, generator=g)],
                 "conf": [torch.sigmoid(_logit(w["conf"][0]) + sigma * 0.6 * torch.randn(w["conf"][0].shape, generator=g))]}
                for w in wants]

    floors = noisy(1.0)  # r ~ 0.7, like PyTorch-ROCm's bf16 execution on the last SSD level
    rows = check_small_levels_pooled(noisy(1.0), floors, wants, "synthetic small", "bfloat16")
    assert len(rows) == 2 and all(r["values"] == draws * w for r, w in zip(rows, (48, 60)))
    zeros = [{"loc": [torch.zeros_like(w["loc"][0])], "conf": [torch.full_like(w["conf"][0], 0.018)]} for w in wants]
    shuffled = [{"loc": [w["loc"][0].flip(1)], "conf": [w["conf"][0].flip(1)]} for w in wants]
    for name, bad in (("twice the noise", noisy(2.0)), ("zeros", zeros), ("shuffled", shuffled)):
        with pytest.raises(AssertionError):
            check_small_levels_pooled(bad, floors, wants, "synthetic small " + name, "bfloat16")


def test_both_dtypes_stay_paired_on_the_network_cases():
    """The fixture and bench-size comparisons run every case in bfloat16 AND float16 (the fp16 run is the one that discriminates
    on the deep levels: its floor sits at r >= 0.95): nobody drops one of the two silently."""
    import test_gpu_bench_sizes
    import test_gpu_nets

    marks = [m for m in test_gpu_nets.test_plan_matches_reference_module.pytestmark if m.name == "parametrize"]
    dtypes = [m.args[1] for m in marks if m.args[0] == "dtype"]
    assert dtypes == [["bfloat16", "float16"]], dtypes
    rows = [m.args[1] for m in test_gpu_bench_sizes.test_forward_at_bench_size_against_the_fp32_module.pytestmark
            if m.name == "parametrize"][0]
    for cfg in ("ssd_mobilenetv2_512.yml", "fpn_resnet50_640.yml"):
        assert {r[2] for r in rows if r[0] == cfg} == {"bfloat16", "float16"}, cfg
