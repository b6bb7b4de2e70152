"""Shared by the CPU and GPU tests of the whole-detector fixtures (tests/golden/net_*.npz, written by
make_golden.gen_nets from the REFERENCE's SSD / SSDFPN / SSDBiFPN classes): builds this repo's model for a case,
checks its ``state_dict`` schema against the reference's and loads the seeded weights + stored BatchNorm statistics."""
import os
import re

import numpy as np
import torch

import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# parameters of the reference backbones that the detector never reads (classifier tails: mobilenet.py:91-99,
# torchvision ResNet.fc, regnet.py AnyHead) and this repo does not instantiate
UNUSED_TAILS = ("backbone.head_conv.", "backbone.classifier.", "backbone.fc.", "backbone.head.")


class StubBackbone(torch.nn.Module):
    def __init__(self, feats):
        super().__init__()
        self.feats = feats

    def initialize(self):
        pass

    def forward(self, x):
        return [f.to(device=x.device, dtype=x.dtype).clone() for f in self.feats]


_FINAL_CONF = re.compile(r"^conf\.\d+\.(weight|bias)$")


def untrained_score_prior(state):
    """An untrained-looking score distribution (logits ~ N(-4, ~1), a few confident peaks) applied to the FINAL class
    convolutions ONLY -- ``conf.<i>.weight|bias``: the six leaf convolutions of SSD (ssd.py:100-103), the last convolution
    of the shared tower of SSDFPN / SSDBiFPN (fpn.py:10-18: ``conf.4``).  Round 4 matched ``conf.*bias``, which in the
    tower also names the BatchNorm betas ``conf.<i>.1.bias`` of the four ConvBNReLU layers: beta ~ -4 put every tower
    activation behind the ReLU's zero, and the bench-size FPN / BiFPN case compared a constant.  In place; returns the
    number of tensors touched."""
    n = 0
    for k in state:
        m = _FINAL_CONF.match(k)
        if not m:
            continue
        n += 1
        if m.group(1) == "weight":
            state[k] = state[k] * np.float32(0.6)
        else:
            state[k] = (state[k] * 3 - 4.0).astype(np.float32)
    return n


def load_fixture(name):
    return np.load(os.path.join(GOLD, "net_%s.npz" % name))


def build(name, fx=None):
    """-> (model in eval mode on the CPU in fp32 with the case's weights, image tensor, fixture)."""
    from ssds.modeling import nets, ssds

    fx = fx if fx is not None else load_fixture(name)
    seed, head, net, fl, A, C, _ = cases.NET_CASES[name]
    cls = getattr(ssds, head)
    nets_outputs, extras, hd = cls.add_extras(feature_layer=[list(f) if isinstance(f, list) else f for f in fl],
                                              mbox=[A] * len(fl[0]), num_classes=C)
    if net == "stub":
        backbone = StubBackbone([torch.from_numpy(f) for f in cases.stub_features(name)])
    else:
        backbone = getattr(nets, net)(outputs=nets_outputs)
    model = cls(backbone=backbone, extras=extras, head=hd, num_classes=C)
    # --- the state_dict schema is the reference's (checkpoint compatibility, SURVEY f-4) ---
    ref_spec = [(str(k), tuple(int(s) for s in str(sh).split(",")) if str(sh) else ())
                for k, sh in zip(fx["keys"], fx["shapes"])]
    ref = dict(ref_spec)
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    missing = [k for k in ref if k not in mine and not k.startswith(UNUSED_TAILS)]
    extra = [k for k in mine if k not in ref]
    assert not missing, "reference parameters this model lacks: %s" % missing[:8]
    assert not extra, "parameters the reference does not have: %s" % extra[:8]
    bad = [(k, mine[k], ref[k]) for k in mine if mine[k] != ref[k]]
    assert not bad, "shape mismatch: %s" % bad[:4]
    # order too: Sequential indices / ModuleList positions decide which weight goes where
    assert [k for k, _ in ref_spec if k in mine] == list(mine), "state_dict key order differs from the reference"
    state = cases.seeded_state(ref_spec, seed)
    for k in list(state):
        if "bn/" + k in fx:
            state[k] = fx["bn/" + k]
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items() if k in mine})
    return model.eval(), torch.from_numpy(cases.net_image(name)), fx


def want(fx):
    n = len([k for k in fx.files if k.startswith("loc")])
    return [torch.from_numpy(fx["loc%d" % i]) for i in range(n)], [torch.from_numpy(fx["conf%d" % i]) for i in range(n)]
