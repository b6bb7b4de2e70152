#!/usr/bin/env python
"""CPU baseline as BASELINE.md section 3 defines it: the REFERENCE's own code timed on this host.

    python tools/cpu_reference_baseline.py [--runs 10] [--warmup 3] [--out profiles/r02_cpu_reference.json]

Runs only in the build container (needs /root/reference; the GPU box has no reference, so bench.py *cites* the JSON
this writes next to its own on-box port timing).  Two measurements on SSD-MobileNetV2@512-shaped synthetic inputs,
full batch of 64, fp32, threads in {1, all}:

  decoder   ssds.modeling.layers.decoder.Decoder.__call__ (decoder.py:25-49: box.decode per level + box.nms) on
            SURVEY 8d's microbench heads  conf = sigmoid(N(-4.6, 1.5^2)), loc = N(0, 0.5^2)
  model     the reference's SSD class (ssd.py:42-74) around its MobileNetV2 backbone (nets/mobilenet.py on the
            torchvision shim of tests/golden/tv_shim.py), eval forward of torch.rand images, reference init

images/sec of the hot path = 64 / (model + decoder seconds).
"""
import argparse
import json
import os
import platform
import sys
import time
import warnings
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--model-runs", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference.json"))
    args = ap.parse_args()

    import importlib
    import types

    import numpy as np
    import torch

    import tv_shim

    tv_shim.install()
    from ssds.modeling.layers import box as rbox
    from ssds.modeling.layers.decoder import Decoder as RDecoder
    from ssds.modeling import ssds as rssds

    pkg = types.ModuleType("ssds.modeling.nets")
    pkg.__path__ = ["/root/reference/ssds/modeling/nets"]
    sys.modules["ssds.modeling.nets"] = pkg
    rmob = importlib.import_module("ssds.modeling.nets.mobilenet")

    B, A, C = args.batch, 6, 80
    maps, strides = [32, 16, 8, 4, 2, 1], [16, 32, 64, 128, 256, 512]
    g = torch.Generator().manual_seed(1234)
    conf = [torch.sigmoid(torch.randn(B, A * C, m, m, generator=g) * 1.5 - 4.6) for m in maps]
    loc = [torch.randn(B, A * 4, m, m, generator=g) * 0.5 for m in maps]
    anchors = OrderedDict((s, rbox.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in strides)
    dec = RDecoder(0.01, 0.6, 100, 300, True, True)

    fl = [[5, 7, "Conv:S", "Conv:S", "Conv:S", "Conv:S"], [96, 320, 512, 256, 256, 128]]
    nets_outputs, extras, head = rssds.SSD.add_extras(feature_layer=fl, mbox=[A] * 6, num_classes=C)
    backbone = rmob.MobileNetV2(outputs=nets_outputs)
    backbone.url = None
    torch.manual_seed(1234)
    model = rssds.SSD(backbone=backbone, extras=extras, head=head, num_classes=C).eval()
    x = torch.rand(B, 3, 512, 512, generator=g)

    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ncores = os.cpu_count()
    result = {
        "what": "reference's own CPU path (ShuangXieIrene/ssds.pytorch v1.5 imported unmodified), SSD-MobileNetV2@512 "
                "shapes, batch %d, fp32" % B,
        "host": {"cpu": cpu_model, "logical_cores": ncores, "machine": platform.machine(),
                 "torch": torch.__version__},
        "inputs": "decoder: conf = sigmoid(N(-4.6, 1.5^2)), loc = N(0, 0.5^2) (SURVEY 8d microbench heads); model: "
                  "torch.rand images, reference init",
        "protocol": "%d warm-ups + %d timed runs of Decoder.__call__ per thread setting (time.perf_counter); "
                    "model: 1 warm-up + %d runs" % (args.warmup, args.runs, args.model_runs),
        "threads": {},
    }
    for nt in (1, ncores):
        torch.set_num_threads(nt)
        for _ in range(args.warmup):
            dec(loc, conf, anchors)
        ts = []
        for _ in range(args.runs):
            t0 = time.perf_counter()
            out = dec(loc, conf, anchors)
            ts.append(time.perf_counter() - t0)
        kept = int((out[0] > 0).sum())
        with torch.no_grad():
            model(x[:8])
            tm = []
            for _ in range(args.model_runs):
                t0 = time.perf_counter()
                model(x)
                tm.append(time.perf_counter() - t0)
        d, m = float(np.median(ts)), float(np.median(tm))
        result["threads"][str(nt)] = {
            "decoder_s_per_batch": {"median": d, "min": float(min(ts)), "max": float(max(ts)), "runs": ts},
            "decoder_images_per_s": B / d,
            "model_forward_s_per_batch": {"median": m, "runs": tm},
            "hot_path_images_per_s": B / (d + m),
            "detections_kept": kept,
        }
        print(nt, "threads: decoder %.2f s/batch (%.1f img/s), model %.2f s/batch, hot path %.2f img/s" % (
            d, B / d, m, B / (d + m)), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(result, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
