"""The network forward and the decode stage at the sizes bench.py runs (BASELINE configs 2, 3, 5), against the fp32 CPU
module / the numpy oracle on sampled images.

Why this file exists: the reference-class fixtures of test_gpu_nets.py top out at 160 px inputs, and the kernels the
planner picks depend on the size -- the register-flow block kernel (`mbflow_kernel`) is only auto-selected from 2 048 work
items on, i.e. at bench size; the halo-tile head kernel only from 96 tiles on.  bench.py's own `verified` compares the
GPU with the oracle on the GPU's heads; this file pins the heads themselves (ssd.py:42-74, fpn.py:58-101,
bifpn.py:104-142 at B = 64 / 32 / 16)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import cases
import nethelp
from oracle import box_oracle as O
from test_gpu_box import BOX_ATOL
from test_gpu_nets import SMALL, _check_against_floor, check_small_levels_pooled, floor_runs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _seeded_model(cfg_name, seed=321):
    """create_model(cfg) with seeded weights (tests/golden/cases.seeded_state), an untrained-looking score distribution
    and BatchNorm statistics calibrated on two random batches -> (fp32 CPU module in eval mode, cfg)."""
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder

    config.reset_cfg()
    cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", cfg_name))
    torch.manual_seed(seed)
    model = model_builder.create_model(cfg.MODEL)
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    state = cases.seeded_state(spec, seed)
    touched = nethelp.untrained_score_prior(state)  # the FINAL class convolutions only (not the towers' BatchNorm betas)
    assert touched == (2 * len(model.conf) if isinstance(model.conf, torch.nn.ModuleList) else 2), touched
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = None
    model.train()
    g = torch.Generator().manual_seed(7)
    h, w = cfg.MODEL.IMAGE_SIZE
    with torch.no_grad():
        for _ in range(2):
            model(torch.rand((2, 3, h, w), generator=g))
    return model.eval(), cfg


@pytest.mark.parametrize("cfg_name,batch,dtype,expect", [
    ("ssd_mobilenetv2_512.yml", 64, "bfloat16", ("mbflow", "conv3x3_halo", "conv3x3_short", "conv_smallmap", "conv_smallmap_group", "xpair")),
    ("ssd_mobilenetv2_512.yml", 64, "float16", ("mbflow", "conv3x3_halo", "conv3x3_short", "conv_smallmap_group")),
    ("fpn_resnet50_640.yml", 32, "float16", ("stem7", "conv3x3_halo")),
    ("fpn_resnet50_640.yml", 32, "bfloat16", ("stem7", "conv3x3_halo")),  # BASELINE config 3's own dtype
    ("bifpn_regnetx008_896.yml", 16, "float16", ("gconv3x3_g16", "fuse", "conv3x3_halo")),
])
def test_forward_at_bench_size_against_the_fp32_module(cfg_name, batch, dtype, expect, monkeypatch):
    """BASELINE configs 2 / 3 / 5 at their batch sizes: the recorded plan (with the kernels the planner picks AT THIS
    SIZE: asserted by name) against the fp32 CPU forward of the same module on 4 of the images, under the same
    noise-floor-relative bar as the reference-class fixtures (PyTorch-ROCm executing the module in the same dtype)."""
    import torch
    from ssds.modeling.layers import fused_conv as FC

    tdt = getattr(torch, dtype)
    cpu_model, cfg = _seeded_model(cfg_name)
    h, w = cfg.MODEL.IMAGE_SIZE
    g = torch.Generator().manual_seed(99)
    x = torch.rand((batch, 3, h, w), generator=g)
    pick = [0, batch // 3, (2 * batch) // 3, batch - 1]
    with torch.no_grad():
        wl, wc = cpu_model(x[pick])
    ref_state = {k: v.clone() for k, v in cpu_model.state_dict().items()}
    model = cpu_model.cuda().to(tdt)
    xd = x.cuda().to(tdt)
    runs = FC.STATS["plan_runs"]
    with torch.no_grad():
        loc, conf = model(xd)
    torch.cuda.synchronize()
    assert FC.STATS["plan_runs"] == runs + 1, "the forward did not run as one recorded plan"
    plan = model._plan(xd) if hasattr(model, "_plan") else next(iter(model._neck_plans.values()))
    assert not isinstance(plan, str), plan
    plan.ctx.set_op_profiling(True)
    with torch.no_grad():
        loc2, conf2 = model(xd)
    torch.cuda.synchronize()
    names = [k for k, _ in plan.ctx.op_timings()]
    plan.ctx.set_op_profiling(False)
    for kern in expect:
        assert any(kern in n for n in names), "%s was not selected at this size: %s" % (kern, sorted(set(names)))
    for a, b in zip(tuple(loc) + tuple(conf), tuple(loc2) + tuple(conf2)):
        assert torch.equal(a, b), "replay is not deterministic"
    # noise floor: the same module, same dtype, on PyTorch-ROCm (4 images), three executions
    floor = floor_runs(model, xd[pick])
    got = {"loc": [t[pick] for t in loc], "conf": [t[pick] for t in conf]}
    # no dead towers: the class logits of every level vary (round 4 shifted the towers' BatchNorm betas by -4 and compared
    # a constant, nethelp.untrained_score_prior)
    for i, c in enumerate(wc):
        p = c.float().clamp(1e-7, 1.0 - 1e-7)
        assert float((torch.log(p) - torch.log1p(-p)).std()) > 0.3, "conf level %d of the fp32 reference is dead" % i
    # (class heads as logits, see _check_against_floor; tail_factor 3: the floor here is PyTorch-ROCm on 4 images, the plan
    #  ran the whole batch -- MIOpen picks its algorithms per batch size, which moves the floor's own tail by ~1.5x)
    _check_against_floor(got, floor, {"loc": wl, "conf": wc}, "bench size %s B=%d" % (cfg_name, batch), dtype, tail_factor=3.0)
    # small levels (SSD: the 2x2 / 1x1 maps): pooled over the WHOLE batch at full strength (test_gpu_nets.py, comment at SMALL)
    if any(w.numel() < SMALL for w in tuple(wl) + tuple(wc)):
        cpu_ref, _ = _seeded_model(cfg_name)
        with torch.no_grad():
            al, ac = cpu_ref(x)
        fa = floor_runs(model, xd, runs=1)[0]
        cpu = lambda t: t.float().cpu()
        rows = check_small_levels_pooled([{"loc": [cpu(t) for t in loc], "conf": [cpu(t) for t in conf]}],
                                         [{"loc": [cpu(t) for t in fa["loc"]], "conf": [cpu(t) for t in fa["conf"]]}],
                                         [{"loc": list(al), "conf": list(ac)}], "bench size %s B=%d" % (cfg_name, batch), dtype,
                                         small=SMALL * batch // len(pick))  # (the levels that are small on the picked images)
        assert rows
    del ref_state


@pytest.mark.parametrize("kind", ["all_equal", "prior_plus_peaks"])
def test_full_size_ssd512_all_ties_vs_oracle(kind):
    """The bench's own input class at the bench's size: reference init puts every one of the 41.9 M scores of a batch at
    sigmoid(-log 99) = the bf16 value 0.010009766 -- above the threshold and tied with all the others, so the flat index
    decides every top-k (box.py:446 under the (score desc, index asc) contract).  B = 64, six levels, bf16; final and
    per-level outputs against the oracle on three images; `prior_plus_peaks` adds a few hundred distinct confident
    scores per image on top of the tie floor (what a barely-trained head looks like)."""
    import torch
    from ssds.modeling.layers.box import decode_nms

    torch.manual_seed(77)
    B, A, C = 64, 6, 80
    maps, strides = [32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512]
    prior = float(torch.sigmoid(torch.tensor(-np.log(99.0), dtype=torch.float32)).to(torch.bfloat16))
    assert prior > 0.01
    conf = []
    for m in maps:
        c = torch.full((B, A * C, m, m), prior, device="cuda")
        if kind == "prior_plus_peaks":
            hot = torch.rand(c.shape, device="cuda") < 2e-3
            c = torch.where(hot, 0.02 + 0.9 * torch.rand(c.shape, device="cuda"), c)
        conf.append(c.to(torch.bfloat16))
    loc = [(torch.randn(B, A * 4, m, m, device="cuda") * 0.5).to(torch.bfloat16) for m in maps]
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828]))) for s in strides)
    args = (0.01, 300, True, 0.6, 100, True)
    (s, b, c), mid = decode_nms(loc, conf, anchors, *args, return_mid=True)
    (s2, b2, c2), mid2 = decode_nms(loc, conf, anchors, *args, return_mid=True)
    for x_, y_ in zip((s, b, c) + tuple(mid), (s2, b2, c2) + tuple(mid2)):
        assert torch.equal(x_, y_)
    sn, bn, cn = s.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    ms, mb, mc = (t.cpu().numpy() for t in mid)
    oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
    odec = O.Decoder(0.01, 0.6, 100, 300, True, True)
    for img in (0, 31, 63):
        ol = [l[img:img + 1].float().cpu().numpy() for l in loc]
        oc = [x_[img:img + 1].float().cpu().numpy() for x_ in conf]
        wm = odec.decode_levels(ol, oc, oanch)
        np.testing.assert_array_equal(mc[img:img + 1], wm[2])
        np.testing.assert_allclose(mb[img:img + 1], wm[1], atol=BOX_ATOL, rtol=0)
        np.testing.assert_allclose(ms[img:img + 1], wm[0], atol=1e-4, rtol=1e-4)
        w = odec(ol, oc, oanch)
        np.testing.assert_array_equal(cn[img:img + 1], w[2])
        np.testing.assert_allclose(bn[img:img + 1], w[1], atol=BOX_ATOL, rtol=0)
        np.testing.assert_allclose(sn[img:img + 1], w[0], atol=1e-4, rtol=1e-4)
