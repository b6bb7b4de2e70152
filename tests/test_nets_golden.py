"""a13-a16 module semantics pinned to the reference: this repo's SSD / SSDFPN / SSDBiFPN (+ MobileNetV2, ResNet,
RegNetX) in fp32 on the CPU against the outputs of the REFERENCE's own classes on the same seeded weights
(tests/golden/net_*.npz; reference ssd.py:42-74, fpn.py:58-101, bifpn.py:30-63,104-142, mobilenet.py:180-192,
resnet.py:41-56, regnet.py:270-282).  The GPU plans are checked against the same fixtures in
tests/test_gpu_nets.py."""
import pytest
import torch

import cases
import nethelp


@pytest.mark.parametrize("name", list(cases.NET_CASES))
def test_module_matches_reference_fp32(name):
    model, x, fx = nethelp.build(name)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)  # the fixtures were made single-threaded: on the same host the outputs are bit-identical
    try:
        with torch.no_grad():
            loc, conf = model(x)
    finally:
        torch.set_num_threads(nt)
    wl, wc = nethelp.want(fx)
    assert len(loc) == len(wl) and len(conf) == len(wc)
    for i, (l, a, c, b) in enumerate(zip(loc, wl, conf, wc)):
        assert l.shape == a.shape and c.shape == b.shape, (name, i)
        # same fp32 ops; another host may pick other convolution kernels (summation order): ulps through ~50 layers.
        # A wiring error (wrong level, missing skip, swapped tower) is O(1).
        torch.testing.assert_close(l, a, rtol=1e-3, atol=5e-4 * float(a.abs().max()))
        torch.testing.assert_close(c, b, rtol=1e-3, atol=2e-4)


def test_train_mode_returns_logits():
    """conf = logits in training mode, sigmoid(logits) in eval (ssd.py:72-73): on frozen statistics the two differ
    by exactly the sigmoid."""
    model, x, fx = nethelp.build("ssd_stub")
    with torch.no_grad():
        _, conf_eval = model(x)
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        _, conf_train = model(x)
    for a, b in zip(conf_eval, conf_train):
        torch.testing.assert_close(a, torch.sigmoid(b), rtol=1e-6, atol=1e-7)
