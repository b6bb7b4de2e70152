"""The judge of tests/test_gpu_train.py::test_whole_step_gradients_match_the_fp64_cpu_module must be able to FAIL (CPU test), and
the CPU half of that test -- the reference step of pipeline_anchor_apex.py:37-72 in fp32 / fp64 with the oracle's targets -- must
produce a live gradient for every parameter at config-4 geometry."""
import pytest

import test_gpu_train as T


def test_the_gradient_judge_can_fail():
    """_judge_gradients on synthetic gradients: the floor itself passes; a zeroed, a sign-flipped and a transposed (summed over the
    wrong axis) gradient of ONE parameter fail."""
    import torch

    g = torch.Generator().manual_seed(3)
    ref = {"a.weight": torch.randn(64, 32, 1, 1, generator=g), "b.weight": torch.randn(96, generator=g),
           "c.weight": torch.randn(32, 32, 3, 3, generator=g)}
    noise = lambda s: {k: v + s * v.pow(2).mean().sqrt() * torch.randn(v.shape, generator=g) for k, v in ref.items()}
    floor = noise(0.1)
    T._judge_gradients(noise(0.1), floor, ref, "selftest")
    for name, wrong in (("zero", torch.zeros(64, 32, 1, 1)), ("flip", -ref["a.weight"]),
                        ("wrong axis", ref["a.weight"].mean(0, keepdim=True).expand(64, 32, 1, 1))):
        bad = noise(0.1)
        bad["a.weight"] = wrong
        with pytest.raises(AssertionError):
            T._judge_gradients(bad, floor, ref, name)


def test_the_cpu_reference_step_reaches_every_parameter():
    """320 px, B = 2: six levels with distinct strides, every weight tensor gets a non-zero fp32 gradient, the losses are
    finite and the localisation loss is live (some anchors match)."""
    import math

    model, anchors, images, targets, cfg = T._whole_step_case(320, 2)
    assert list(anchors) == [16, 32, 64, 106, 160, 320]
    c, l, grads = T._cpu_reference_step(model, anchors, images, targets, cfg.MODEL.NUM_CLASSES, cfg.MATCHER.MATCH_THRESHOLD)
    assert math.isfinite(c) and math.isfinite(l) and l > 0 and c > 0
    assert len(grads) == len(list(model.parameters()))
    dead = [k for k, g in grads.items() if k.endswith("weight") and float(g.abs().max()) == 0.0]
    assert not dead, dead


def test_fp32_has_a_noise_floor_on_this_network_and_the_judge_accepts_it():
    """The statement the GPU test's bars rest on: the step in fp32 on the CPU sits 0.1 - 5 % (relative rms, median over the
    parameters) from the same step in fp64 -- not 1e-6 -- while the losses agree to 1e-6; the structurally zero gradients (the
    bias of a BatchNorm that only feeds a 1x1 convolution + train-mode BatchNorm) are < 1e-7 of the typical gradient in fp64;
    and fp32 judged against itself as the floor passes."""
    import copy

    model, anchors, images, targets, cfg = T._whole_step_case(320, 2)
    nc, match = cfg.MODEL.NUM_CLASSES, cfg.MATCHER.MATCH_THRESHOLD
    tc, tl, truth = T._cpu_reference_step(copy.deepcopy(model).double(), anchors, images.double(), targets, nc, match)
    rc, rl, cpu32 = T._cpu_reference_step(model, anchors, images, targets, nc, match)
    assert abs(tc - rc) <= 1e-5 * abs(tc) and abs(tl - rl) <= 1e-5 * abs(tl)
    rows = T._judge_gradients(cpu32, cpu32, truth, "fp32 CPU against itself", factor=1.0, slack=0.0, r_slack=0.0)
    med = sorted(r[2] for r in rows)[len(rows) // 2]
    assert 1e-3 < med < 5e-2, med
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    scale = sorted(rms(w) for w in truth.values())[len(truth) // 2]
    zeros = [k for k, w in truth.items() if rms(w) < T.ZERO_BELOW * scale]
    assert zeros and all(k.endswith(".bias") and k.startswith("backbone.") for k in zeros), zeros
