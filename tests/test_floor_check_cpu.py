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
