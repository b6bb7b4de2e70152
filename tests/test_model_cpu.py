"""Host-side plugin surface (no GPU): config schema behaviour, the model_builder factories for the four
BASELINE geometries (shapes, strides, anchor counts of SURVEY.md section 8), checkpoint round trip."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "experiments", "cfgs")


@pytest.fixture(autouse=True)
def fresh_cfg():
    from ssds.core import config

    config.reset_cfg()
    yield
    config.reset_cfg()


def test_config_defaults_and_errors(tmp_path):
    from ssds.core import config

    c = config.cfg
    assert c.POST_PROCESS.SCORE_THRESHOLD == 0.01 and c.POST_PROCESS.MAX_DETECTIONS_PER_LEVEL == 300
    assert c.MATCHER.MATCH_THRESHOLD == [0.5, 0.4] and c.POST_PROCESS.USE_DIOU is True
    bad = tmp_path / "bad.yml"
    bad.write_text("MODEL:\n  NOT_A_KEY: 1\n")
    with pytest.raises(KeyError, match="Non-existent config key: MODEL.NOT_A_KEY"):
        config.cfg_from_file(str(bad))
    bad.write_text("MODEL:\n  NUM_CLASSES: 'eighty'\n")
    with pytest.raises(ValueError, match="Type mismatch"):
        config.cfg_from_file(str(bad))
    ok = tmp_path / "ok.yml"
    ok.write_text("MODEL:\n  NUM_CLASSES: 7\n  IMAGE_SIZE: (320, 320)\nTRAIN:\n  MAX_EPOCHS: 9\n"
                  "  LR_SCHEDULER:\n    WARM_UP_EPOCHS: 2\n")
    c = config.cfg_from_file(str(ok))
    assert c.POST_PROCESS.NUM_CLASSES == 7 and c.MATCHER.NUM_CLASSES == 7  # derived fields
    assert c.MODEL.IMAGE_SIZE == [320, 320] and c.DATASET.IMAGE_SIZE == [320, 320]
    assert c.TRAIN.LR_SCHEDULER.MAX_EPOCHS == 7


GEOM = [
    ("ssd_mobilenetv2_300", [15, 30, 60, 100, 150, 300], [19, 10, 5, 3, 2, 1], 6, 3000),
    ("ssd_mobilenetv2_512", [16, 32, 64, 128, 256, 512], [32, 16, 8, 4, 2, 1], 6, 8190),
]


@pytest.mark.parametrize("name,strides,maps,A,total", GEOM)
def test_ssd_builder_geometry(name, strides, maps, A, total):
    from ssds.core import config
    from ssds.modeling import model_builder

    cfg = config.cfg_from_file(os.path.join(CFG, name + ".yml"))
    model = model_builder.create_model(cfg.MODEL)
    assert model.training
    anchors = model_builder.create_anchors(cfg.MODEL, model, cfg.MODEL.IMAGE_SIZE)
    assert model.training, "create_anchors must restore the mode"
    assert list(anchors.keys()) == strides
    assert all(tuple(a.shape) == (A, 4) and a.dtype == torch.float32 and not a.is_cuda for a in anchors.values())
    x = torch.rand(2, 3, *cfg.MODEL.IMAGE_SIZE)
    loc, conf = model(x)  # training mode: logits
    assert [c.shape[-1] for c in conf] == maps
    assert all(l.shape[1] == A * 4 and c.shape[1] == A * 80 for l, c in zip(loc, conf))
    assert sum(A * m * m for m in maps) == total
    model.eval()
    with torch.no_grad():
        _, conf_e = model(x)
    assert all(float(c.min()) >= 0 and float(c.max()) <= 1 for c in conf_e)  # sigmoid in eval (ssd.py:72)
    # conf prior: bias = -log(99) -> an untrained head scores ~0.01 (ssdsbase.py:15-19)
    assert abs(float(conf_e[0].mean()) - 0.01) < 5e-3
    dec = model_builder.create_decoder(cfg.POST_PROCESS)
    assert (dec.conf_threshold, dec.nms_threshold, dec.top_n, dec.top_n_per_level, dec.rescore, dec.use_diou) == (
        0.01, 0.6, 100, 300, True, True)
    names = list(model.state_dict().keys())
    assert "backbone.conv1.0.weight" in names and "extras.0.3.weight" in names
    assert "loc.5.bias" in names and "conf.0.weight" in names


def test_fpn_and_bifpn_build_small():
    from ssds.modeling import nets, ssds

    for cls, net, outs, depth in ((ssds.SSDFPN, "ResNet18", [3, 4, 5], [128, 256, 512]),
                                  (ssds.SSDBiFPN, "RegNetX002", [2, 3, 4], [56, 152, 368])):
        fl = [outs + ["Conv:S", "Conv:S"], depth + [depth[-1], 256]]
        nets_outputs, extras, head = cls.add_extras(fl, [9] * 5, 4)
        model = cls(getattr(nets, net)(outputs=nets_outputs), extras, head, 4).eval()
        with torch.no_grad():
            loc, conf = model(torch.rand(1, 3, 128, 128))
        assert [c.shape[-1] for c in conf] == [16, 8, 4, 2, 1]
        assert all(l.shape[1] == 36 and c.shape[1] == 36 for l, c in zip(loc, conf))
    with pytest.raises(ValueError, match="number of box"):
        ssds.SSDFPN.add_extras([[3, 4], [128, 256]], [9, 6], 4)


def test_backbone_registry_and_parser():
    from ssds.modeling import nets
    from ssds.modeling.layers.layers_parser import parse_feature_layer

    for n in ("MobileNetV1", "MobileNetV2", "ResNet50", "RegNetX008"):
        assert hasattr(nets, n)
    assert nets.RegNetX008(outputs=[2, 3, 4]).stage_widths == [64, 128, 288, 672]
    with pytest.raises(AssertionError, match="Undefined layer"):
        parse_feature_layer("Bogus", 8, 8)


def test_checkpoint_round_trip(tmp_path):
    from ssds.core import checkpoint
    from ssds.modeling import nets, ssds

    def make():
        o, e, h = ssds.SSD.add_extras([[5, 7, "Conv:S"], [96, 320, 64]], [2, 2, 2], 3)
        return ssds.SSD(nets.MobileNetV2(outputs=o), e, h, 3)

    a, b = make(), make()
    path = checkpoint.save_checkpoints(a, str(tmp_path), "ssd_test", 3)
    assert checkpoint.find_previous_checkpoint(str(tmp_path)) == ([3], [path])
    # a reference checkpoint also carries the unused classifier tail and a DataParallel prefix
    sd = {"module." + k: v for k, v in torch.load(path).items()}
    sd["module.backbone.classifier.1.weight"] = torch.zeros(10, 10)
    torch.save(sd, path)
    assert checkpoint.resume_checkpoint(b, path, "") is b
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    assert checkpoint.resume_checkpoint(b, str(tmp_path / "missing.pth"), "") is False


def test_export_sidecar_has_the_reference_layout(tmp_path):
    """reference export.py:94-106: keys, value layout (flattened [A*4] anchors per level, level order)."""
    import json
    from collections import OrderedDict

    import torch
    from ssds.core import config
    from ssds.utils import export as E
    from oracle import box_oracle as O

    config.reset_cfg()
    cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_300.yml"))
    strides = [15, 30, 60, 100, 150, 300]
    anchors = OrderedDict((s, torch.from_numpy(O.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828]))) for s in strides)
    model = torch.nn.Linear(2, 2)
    jpath, ppath = E.save_export(model, cfg, anchors, str(tmp_path / "net"))
    p = json.load(open(jpath))
    assert list(p) == ["image_size", "score", "iou", "max_detects", "max_detects_per_level", "rescore", "use_diou",
                       "NHWC", "anchors"]
    assert p["image_size"] == [300, 300] and p["score"] == 0.01 and p["iou"] == 0.6 and p["max_detects"] == 100
    assert len(p["anchors"]) == 6 and all(len(a) == 24 for a in p["anchors"])
    assert p["anchors"][0] == anchors[15].reshape(-1).tolist()
    assert set(torch.load(ppath)) == {"weight", "bias"}
    dec, back = E.decoder_from_params(p)
    assert list(back) == strides and all(torch.equal(back[s], anchors[s]) for s in strides)
    assert dec.top_n == 100 and dec.top_n_per_level == 300 and dec.use_diou and dec.rescore


@pytest.mark.parametrize("name", ["focal", "focal_g15", "multibox", "smoothl1", "iou", "giou", "diou", "ciou"])
def test_criteria_match_reference_outputs_and_gradients(name):
    """core/criterion.py against the reference's criteria on seeded inputs (tests/golden/losses.npz: element-wise
    outputs and autograd gradients produced by the reference's own modules; MultiBoxLoss on one image, the only
    batch size the reference's implementation accepts)."""
    import os

    import numpy as np
    import torch

    import cases
    from ssds.core import criterion

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))
    d = cases.loss_inputs()
    crit = {"focal": lambda: criterion.FocalLoss(0.25, 2), "focal_g15": lambda: criterion.FocalLoss(0.4, 1.5),
            "multibox": lambda: criterion.MultiBoxLoss(3), "smoothl1": criterion.SmoothL1Loss,
            "iou": lambda: criterion.IOULoss("iou"), "giou": criterion.GIOULoss, "diou": criterion.DIOULoss,
            "ciou": criterion.CIOULoss}[name]()
    if name in ("focal", "focal_g15", "multibox"):
        sl = slice(0, 1) if name == "multibox" else slice(None)
        x = torch.from_numpy(d["logits"][sl]).requires_grad_(True)
        y = crit(x, torch.from_numpy(d["target"][sl]), torch.from_numpy(d["depth"][sl]))
    else:
        x = torch.from_numpy(d["pred"]).requires_grad_(True)
        y = crit(x, torch.from_numpy(d["tgt"]))
    y.sum().backward()
    assert y.shape == g[name + "/out"].shape
    np.testing.assert_allclose(y.detach().numpy(), g[name + "/out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(x.grad.numpy(), g[name + "/grad"], rtol=1e-5, atol=1e-6)
    if name == "multibox":  # whole batch: per-image mining (what the reference's expression means for B > 1)
        full = crit(torch.from_numpy(d["logits"]), torch.from_numpy(d["target"]), torch.from_numpy(d["depth"]))
        np.testing.assert_allclose(full[0:1].numpy(), g[name + "/out"], rtol=1e-6, atol=1e-7)
        assert float(full[2].sum()) == 0.0  # no positives -> no mined negatives either


def test_plan_arena_pins_buffers_read_by_side_lane_heads():
    """Recording only (no kernel runs): a buffer read by a small head -- which the executor runs on its side stream,
    concurrently with the main chain -- must never be handed out again by the plan's arena, while the input of a big
    (in-line) head is recycled as usual."""
    import torch.nn as nn

    from ssds.modeling.layers import fused_conv as FC

    dt = torch.bfloat16

    def pack(cin, cout, k=3):
        return FC.ConvPack(nn.Conv2d(cin, cout, k, 1, k // 2), None, "relu", dt)

    for n, expect_lane in ((2, 1), (64, 0)):  # 2 x 16 x 16 = 512 pixels (side lane) / 64 x 16 x 16 = 16384 (in line)
        plan = FC.ConvPlan(torch.device("cpu"), dt, (n, 32, 16, 16))
        x = plan.add_input((n, 32, 16, 16))
        a = plan.conv(x, pack(32, 32))
        plan.head(a, FC.pack_heads(nn.Conv2d(32, 8, 3, 1, 1), nn.Conv2d(32, 12, 3, 1, 1), dt), split=8, act="none",
                  act2="sigmoid")
        assert plan.layers[-1]["lane"] == expect_lane
        b = plan.conv(a, pack(32, 32))
        plan.release(a)                 # the planner is done with `a` ...
        c = plan.conv(b, pack(32, 32))  # ... so the next output may take its place -- unless a side-lane op reads it
        if expect_lane:
            assert a[0] in plan.pinned and c[0] != a[0]
        else:
            assert a[0] not in plan.pinned and c[0] == a[0]


def test_loading_through_a_parent_drops_every_folded_weight_cache():
    """a parent's load_state_dict() reaches children only through _load_from_state_dict, so the caches of
    folded weights (plans, head packs, per-block packs) must be dropped there too -- otherwise the fallback path keeps
    running the old weights after resume_checkpoint / an EMA swap."""
    import torch
    from ssds.modeling import ssds as S

    class Wrapper(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.model = m

    class Stub(torch.nn.Module):
        def initialize(self):
            pass

        def forward(self, x):
            return [x]

    for cls, fl in ((S.SSD, [[0, "Conv:S"], [8, 16]]), (S.SSDFPN, [[0, "Conv:S"], [8, 8]]),
                    (S.SSDBiFPN, [[0, 1, 2, "Conv:S"], [8, 8, 8, 8]])):
        outs, extras, head = cls.add_extras(fl, [2] * len(fl[0]), 3)
        model = cls(Stub(), extras, head, 3).eval()
        marks = []
        for m in model.modules():
            if hasattr(m, "_packs"):  # FusedSequentialMixin
                object.__setattr__(m, "_ssdk_packs", ("stale",))
                marks.append(m)
            if "_final_pack" in m.__dict__ or type(m).__name__ == "SharedHead":
                m.__dict__["_final_pack"] = ("stale",)
        assert marks, cls
        model.__dict__["_plans"] = {"k": "stale"}
        model.__dict__["_neck_plans"] = {"k": "stale"}
        model.__dict__["_head_packs"] = ("stale",)
        w = Wrapper(model)
        w.load_state_dict(w.state_dict())
        assert all(m.__dict__.get("_ssdk_packs") is None for m in marks)
        assert all(m.__dict__.get("_final_pack") is None for m in model.modules())
        if cls is S.SSD:
            assert model.__dict__["_plans"] == {} and model.__dict__["_head_packs"] is None
        else:
            assert model.__dict__["_neck_plans"] == {}


def test_checkpoint_written_by_the_reference_loads_here():
    """SURVEY f-4: tests/golden/ckpt_ref/ was written by the REFERENCE's own ``save_checkpoints``
    (core/checkpoint.py:18-35) from its own SSD class (make_golden.gen_checkpoint; the other direction -- a checkpoint
    written here resumed by the reference's ``resume_checkpoint`` -- is asserted there, where the reference is
    importable).  ``find_previous_checkpoint`` parses its index, ``resume_checkpoint`` loads it into this repo's SSD
    (skipping the backbone's unused classifier tail) and the eval forward reproduces the reference model's outputs."""
    import numpy as np
    import torch
    from ssds.core import checkpoint
    from ssds.modeling import ssds as S

    class StubNet(torch.nn.Module):  # the backbone of the fixture, without the unused tail
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 16, 3, 2, 1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(16)
            self.layer1 = torch.nn.Sequential(torch.nn.Conv2d(16, 24, 3, 2, 1, bias=False), torch.nn.BatchNorm2d(24),
                                              torch.nn.ReLU())

        def initialize(self):
            pass

        def forward(self, x):
            a = torch.relu(self.bn1(self.conv1(x)))
            return [a, self.layer1(a)]

    gdir = os.path.join(ROOT, "tests", "golden")
    epochs, paths = checkpoint.find_previous_checkpoint(os.path.join(gdir, "ckpt_ref"))
    assert epochs == [7, 9] and paths[1] == "tests/golden/ckpt_ref/ssd_stubnet_ref_epoch_9.pth"
    fx = np.load(os.path.join(gdir, "checkpoint_ref.npz"))
    torch.manual_seed(5)
    _, extras, head = S.SSD.add_extras([[0, 1, "Conv:S"], [16, 24, 32]], [2, 2, 2], 3)
    model = S.SSD(StubNet(), extras, head, 3)
    ref_keys = [str(k) for k in fx["keys"]]
    mine = list(model.state_dict())
    assert [k for k in ref_keys if not k.startswith("backbone.classifier.")] == mine
    assert checkpoint.resume_checkpoint(model, os.path.join(ROOT, paths[1]), "") is model
    model.eval()
    with torch.no_grad():
        loc, conf = model(torch.from_numpy(fx["x"]))
    for i, (l, c) in enumerate(zip(loc, conf)):
        torch.testing.assert_close(l, torch.from_numpy(fx["loc%d" % i]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(c, torch.from_numpy(fx["conf%d" % i]), rtol=1e-5, atol=1e-6)
    # scope-filtered resume (cfg.TRAIN.RESUME_SCOPE, checkpoint.py:111-119): only the heads
    torch.manual_seed(6)
    _, extras, head = S.SSD.add_extras([[0, 1, "Conv:S"], [16, 24, 32]], [2, 2, 2], 3)
    part = S.SSD(StubNet(), extras, head, 3)
    before = {k: v.clone() for k, v in part.state_dict().items()}
    checkpoint.resume_checkpoint(part, os.path.join(ROOT, paths[0]), "loc,conf")
    for k, v in part.state_dict().items():
        same_as_ckpt = torch.equal(v, model.state_dict()[k])
        assert same_as_ckpt if k.startswith(("loc.", "conf.")) else torch.equal(v, before[k]), k


def test_pointwise_gemm_switch_keeps_state_dict_and_cpu_behaviour():
    """use_pointwise_gemm re-classes dense 1x1 / stride-1 convolutions only; on CPU they are nn.Conv2d."""
    import torch
    import torch.nn as nn
    from ssds.modeling.layers.pointwise import PointwiseConv2d, use_pointwise_gemm

    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(8, 16, 1, bias=False), nn.BatchNorm2d(16), nn.Conv2d(16, 16, 3, padding=1, groups=16),
                        nn.Conv2d(16, 8, 1, stride=2), nn.Conv2d(8, 4, 1))
    keys = list(net.state_dict().keys())
    x = torch.randn(2, 8, 6, 6)
    y0 = net(x)
    use_pointwise_gemm(net)
    assert [type(m) is PointwiseConv2d for m in net] == [True, False, False, False, True]
    assert list(net.state_dict().keys()) == keys
    assert torch.equal(net(x), y0)


def test_fusing_bn_activations_changes_nothing_on_cpu():
    """fuse_bn_activations marks BN -> ReLU6 pairs inside Sequentials only; on CPU (no kernels) the network computes
    what it computed before, in train and eval mode, and keeps its state_dict."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers.batchnorm import FastBatchNorm2d, fuse_bn_activations, use_fast_batchnorm

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL)
    keys = list(model.state_dict().keys())
    x = torch.randn(2, 3, 128, 128)
    model.eval()
    with torch.no_grad():
        ref = [t.clone() for t in model(x)[0]]
    use_fast_batchnorm(model)
    n = fuse_bn_activations(model)
    assert n >= 35, n  # 17 inverted-residual blocks with two Conv-BN-ReLU6 each + stem + last conv + extras
    assert list(model.state_dict().keys()) == keys
    fused = [m for m in model.modules() if type(m) is FastBatchNorm2d and m._ssdk_act]
    assert len(fused) == n
    with torch.no_grad():
        out = model(x)[0]
    assert all(torch.equal(a, b) for a, b in zip(ref, out))
    model.train()
    y = model(x)
    (y[0][0].float().sum() + y[1][0].float().sum()).backward()  # plain autograd path still differentiates


def test_ssd_mobilenetv2_plan_recording_fuses_extras_and_keeps_head_order():
    """Recording only (CPU, no kernel runs): the recorded plan of SSD-MobileNetV2@512 has one fused block per
    inverted-residual block, the three small extra layers as single `xpair` ops, and its head outputs in LEVEL order
    although the heads of the extras' levels are recorded behind the extras chain, next to each other (main lane: the
    executor launches the three small-map heads that follow the 8x8 one as one kernel)."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import planner

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL).eval().to(torch.bfloat16)
    plan = planner.build_ssd_plan(model, torch.zeros(2, 3, 512, 512, dtype=torch.bfloat16))
    kinds = [L.get("kind") or ("head" if L.get("nchw") else "conv") for L in plan.layers]
    assert kinds.count("mb") == 17 and kinds.count("xpair") == 3 and kinds.count("head") == 6 and kinds.count("conv") == 2
    assert kinds[-4:] == ["head"] * 4 and all(L["lane"] == 0 for L in plan.layers[-4:])  # the extras' heads: last ops, main lane
    assert [(h[4], h[5]) for h in plan.heads] == [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    assert [h[0] for h in plan.heads[2:]] == list(range(len(plan.layers) - 4, len(plan.layers)))   # ... in level order
    xp = [L for L in plan.layers if L.get("kind") == "xpair"]
    assert [(L["h"], L["pack"].cin, L["pack"].cout, L["pack2"].cout) for L in xp] == [(8, 512, 128, 256), (4, 256, 128, 256),
                                                                                      (2, 256, 64, 128)]


def test_blocks_whose_folded_weights_leave_the_fp16_range_are_not_fused():
    """The block kernel's internal tensors are fp16 (include/ssdk.h, ssdk_mbconv): a block whose folded depthwise weights
    could push an intermediate past 65504 -- a BatchNorm with a tiny running variance and a large gain -- is recorded as three
    layer launches instead; the other blocks stay fused.  Recording only (CPU)."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import planner

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL).eval()
    blk = model.backbone.layer3[1]                      # an ordinary residual block
    dw_bn = [m for m in blk.conv.modules() if isinstance(m, torch.nn.BatchNorm2d)][1]
    dw_bn.running_var.fill_(1e-12)                      # sigma = sqrt(eps): gamma / sigma ~ 3e6,
    dw_bn.weight.data.fill_(1e4)                        # 6 * sum|w'| is far outside fp16
    model = model.to(torch.bfloat16)
    x = torch.zeros(2, 3, 512, 512, dtype=torch.bfloat16)
    plan = planner.build_ssd_plan(model, x)
    kinds = [L.get("kind") or ("head" if L.get("nchw") else "conv") for L in plan.layers]
    assert kinds.count("mb") == 16 and kinds.count("conv") == 2 + 3   # that block: expand, depthwise, project
    dw_bn.running_var.fill_(float("nan"))               # a broken checkpoint: not fused either (and no exception here)
    plan = planner.build_ssd_plan(model, x)
    assert [L.get("kind") for L in plan.layers].count("mb") == 16


@pytest.mark.parametrize("px", [300, 320, 416, 448, 512])
def test_xpair_ops_are_only_recorded_on_maps_the_c_entry_accepts(px):
    """ssdk_xpair() takes maps of <= 16 pixels or exactly 64 (csrc/ssdk_xpair.hip); a 7x7 map (416 / 448 px inputs) rounds
    up to four fragments as well but is NOT an instance: it has to be recorded as two conv ops.  Recording only (CPU)."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import planner

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL).eval().to(torch.bfloat16)
    plan = planner.build_ssd_plan(model, torch.zeros(1, 3, px, px, dtype=torch.bfloat16))
    xp = [L for L in plan.layers if L.get("kind") == "xpair"]
    for L in xp:
        assert L["h"] * L["w"] <= 16 or L["h"] * L["w"] == 64, (px, L["h"], L["w"])
    if px == 512:
        assert len(xp) == 3
    if px in (416, 448):  # 26 -> 13 -> 7 -> 4 -> 2 -> 1 / 28 -> 14 -> 7 -> ...: the 7x7 extra is two launches
        assert all(L["h"] != 7 for L in xp)


def test_training_switches_of_round_6_pair_what_they_should_and_change_nothing_on_cpu():
    """fuse_bn_into_depthwise / use_head_pairs (round 6; ssds/utils/train_ddp.py calls both) only MARK modules: the sixteen expand
    BatchNorms of SSD-MobileNetV2 get a (non-registered) reference to the depthwise convolution behind them, the SSD head gets its
    pair flag; no parameter, buffer or state_dict key appears, and on CPU tensors -- where no kernel exists -- training and eval
    forwards are the unmarked model's, bit for bit (headconv.supported refuses CPU tensors, a deferred BatchNorm needs a HIP
    tensor)."""
    import copy
    import os
    import torch
    import torch.nn as nn
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import headconv as HC
    from ssds.modeling.layers.batchnorm import FastBatchNorm2d, fuse_bn_activations, fuse_bn_into_depthwise, use_fast_batchnorm
    from ssds.modeling.layers.dwconv import DepthwiseConv2d

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL)
    use_fast_batchnorm(model)
    fuse_bn_activations(model)
    plain = copy.deepcopy(model)
    keys = list(model.state_dict().keys())
    n_modules = len(list(model.modules()))
    os.environ.pop("SSDK_BN_DEFER", None)
    os.environ.pop("SSDK_HEAD_PAIR", None)
    assert fuse_bn_into_depthwise(model) == 16  # every inverted-residual block with an expansion (the first block has none)
    assert HC.use_head_pairs(model) == 1 and model.__dict__.get("_ssdk_head_pair") is True
    from ssds.modeling.layers.pointwise import StemConv3x3s2, use_native_stem

    assert use_native_stem(model) == 1 and use_native_stem(model) == 0  # the image-side 3x3 / stride-2 convolution, once
    assert sum(1 for m in model.modules() if type(m) is StemConv3x3s2) == 1
    assert list(model.state_dict().keys()) == keys and len(list(model.modules())) == n_modules
    linked = [m for m in model.modules() if type(m) is FastBatchNorm2d and "_ssdk_defer_to" in m.__dict__]
    assert len(linked) == 16 and all(type(m.__dict__["_ssdk_defer_to"]) is DepthwiseConv2d and m._ssdk_act == 1 for m in linked)
    for bn in linked:  # the BatchNorm's channel count is the depthwise convolution's: the pair really is (expand BN -> depthwise)
        assert bn.num_features == bn.__dict__["_ssdk_defer_to"].in_channels
    os.environ["SSDK_BN_DEFER"] = "0"
    os.environ["SSDK_HEAD_PAIR"] = "0"
    try:
        other = copy.deepcopy(plain)
        assert fuse_bn_into_depthwise(other) == 0 and HC.use_head_pairs(other) == 0
    finally:
        del os.environ["SSDK_BN_DEFER"], os.environ["SSDK_HEAD_PAIR"]
    x = torch.randn(2, 3, 128, 128)
    for m in (model, plain):
        m.train()
    torch.manual_seed(1)
    a = model(x)
    torch.manual_seed(1)
    b = plain(x)
    assert all(torch.equal(u, v) for u, v in zip(a[0] + a[1], b[0] + b[1]))
    for m in (model, plain):
        m.eval()
    with torch.no_grad():
        a, b = model(x), plain(x)
    assert all(torch.equal(u, v) for u, v in zip(a[0] + a[1], b[0] + b[1]))
    # the head pair's admission test
    loc, conf = model.loc[0], model.conf[0]
    f = torch.randn(2, loc.in_channels, 8, 8)
    assert not HC.supported(f, loc, conf)  # CPU tensor
    assert not HC.supported(f, nn.Conv2d(loc.in_channels, 24, 1), conf)


def test_a_model_that_went_through_the_training_solver_still_records_its_eval_plan():
    """fuse_bn_activations swaps activation classes to FusedAway* subclasses; the eval planner has to recognise them (it
    used to match activations by exact type and silently fell back to torch for the whole network)."""
    import os
    import torch
    from ssds.core import config
    from ssds.modeling import model_builder
    from ssds.modeling.layers import planner
    from ssds.modeling.layers.batchnorm import fuse_bn_activations, use_fast_batchnorm

    cfg = config.cfg_from_file(os.path.join(os.path.dirname(__file__), "..", "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    torch.manual_seed(0)
    model = model_builder.create_model(cfg.MODEL)
    use_fast_batchnorm(model)
    assert fuse_bn_activations(model) >= 35
    model = model.eval().to(torch.bfloat16)
    plan = planner.build_ssd_plan(model, torch.zeros(2, 3, 512, 512, dtype=torch.bfloat16))
    kinds = [L.get("kind") or ("head" if L.get("nchw") else "conv") for L in plan.layers]
    assert kinds.count("mb") == 17 and kinds.count("xpair") == 3 and kinds.count("head") == 6
    acts = [L["pack"].act for L in plan.layers if L.get("kind") == "xpair"]
    assert acts == ["relu"] * 3 or all(a in ("relu", "relu6") for a in acts)


def test_fragment_major_weight_image_layout_and_size():
    """ConvPack.frag() (pure layout, runs on CPU): block (g, ks), lane (fg, fr) holds w[16 g + fr][32 ks + 8 fg .. + 7] of the
    KRSC matrix, rows past Cout are zero, and the byte count is what the C-ABI's ssdk_weight_frag_bytes promises
    (include/ssdk.h); layers no kernel reads it for get none."""
    import torch
    import torch.nn as nn
    from ssds import _native as N
    from ssds.modeling.layers import fused_conv as FC

    torch.manual_seed(3)
    conv = nn.Conv2d(96, 504, 3, 1, 1, bias=True)
    pack = FC.ConvPack(conv, None, "none", torch.bfloat16)
    img = pack.frag()
    k = 9 * 96
    assert img.shape == (32, k // 32, 4, 16, 8) and img.dtype == torch.bfloat16 and img.is_contiguous()
    assert img.numel() * 2 == N.lib.ssdk_weight_frag_bytes(504, k) == 32 * 16 * k * 2
    w2d = pack.w.reshape(504, k)
    for g, ks, fg, fr in ((0, 0, 0, 0), (7, 13, 3, 9), (31, 26, 2, 7)):
        assert torch.equal(img[g, ks, fg, fr], w2d[16 * g + fr, 32 * ks + 8 * fg:32 * ks + 8 * fg + 8])
    assert float(img[31, :, :, 8:].float().abs().max()) == 0.0          # rows 504 .. 511
    assert pack.frag() is img                                          # built once per weight tensor
    pack.w = pack.w.clone()
    assert pack.frag() is not img                                      # ... and again for a new one
    assert N.lib.ssdk_weight_frag_bytes(504, 100) == 0                 # K must be a multiple of 32
    dw = FC.ConvPack(nn.Conv2d(32, 32, 3, 1, 1, groups=32, bias=False), nn.BatchNorm2d(32), "relu6", torch.bfloat16)
    assert dw.frag() is None                                           # depthwise: no MFMA operand stream
    odd = FC.ConvPack(nn.Conv2d(24, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), "relu", torch.bfloat16)
    assert odd.frag() is None                                          # Cin % 32 != 0


def test_batch_fold_of_the_small_map_layers_is_a_pure_permutation():
    """pointwise._fold / _unfold (round 6): [B, C, P] <-> [1, C, B * P] with pixel index b * P + p -- the layout
    ssdk_im2col3x3_folded writes and ssdk_col2im3x3_folded reads, so that the 1x1 kernels see ONE image of B * P pixels where a
    layer has only 64 / 16 / 4 / 1 pixels per image (the SSD extras).  A round trip is the identity, column b * P + p of the
    folded tensor is pixel p of image b, and a GEMM on the folded operand equals the per-image GEMMs."""
    import torch
    from ssds.modeling.layers import pointwise as PW

    torch.manual_seed(0)
    for b, c, p_ in ((4, 5, 16), (3, 2, 1), (2, 7, 100)):
        t = torch.randn(b, c, p_)
        f = PW._fold(t)
        assert f.shape == (1, c, b * p_) and f.is_contiguous()
        assert torch.equal(PW._unfold(f, b), t)
        for bi in range(b):
            assert torch.equal(f[0, :, bi * p_:(bi + 1) * p_], t[bi])
        w = torch.randn(6, c)
        assert torch.allclose(PW._unfold(torch.matmul(w, f), b), torch.matmul(w, t), atol=1e-6)
    assert PW.FOLD_BELOW == 128  # (the 1x1 kernels' pixel group)
