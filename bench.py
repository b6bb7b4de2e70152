#!/usr/bin/env python
"""bench.py -- images/sec of the detection hot path (fwd + decode + NMS) of SSD-MobileNetV2@512 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
backbone + extras + multibox heads forward (bf16), decode of every level and NMS.  BASELINE.json's metric
is quoted on `configs[1]` (SSD + MobileNetV2 @512x512 bf16, batch 64, one MI355X); with N > 1 every rank
runs the same per-GPU batch (weak scaling, replicas only: inference has no collective -- SURVEY.md 8e).

Serving-loop pipelining (default, --tail-stream 0 turns it off): the latency-bound end of the decode stage
(level_kernel + nms_kernel, 64-384 workgroups) is enqueued on its own HIP stream and runs under the NEXT step's
forward pass; the HBM-bound scan_kernel stays in line on the main stream, so its live event timing is
un-overlapped.  All work of the K steps completes inside the timed region (device-wide synchronize on both sides).

Prints ONE JSON line (rank 0).  Besides the driver's contract fields it carries
  roofline      HBM roofline of the dominant hand-written kernel (scan_kernel: the one pass over the conf
                tensors), from hipEvents recorded live inside the timed region (ssdk_set_profiling ring)
  stages        per-stage milliseconds (forward / decode+NMS kernels) for orientation
  cpu_baseline  the CPU path (torch fp32 forward of the same module + the numpy oracle's Decoder) timed on
                this host on a bounded sample of the same workload (N=1 runs only)
"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ssds.pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md), no sparsity


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--cfg", default=os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
    ap.add_argument("--cpu-sample", type=int, default=4, help="images for the CPU baseline (0 = skip)")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the step as one captured hipGraph")
    ap.add_argument("--tail-stream", type=int, default=1,
                    help="1: level/NMS kernels of the decode stage on their own stream (overlap the next step's forward)")
    ap.add_argument("--layers", type=int, default=0, help="1: add the per-layer table (us, TFLOP/s, GB/s) to the JSON")
    ap.add_argument("--channels-last", type=int, default=int(os.environ.get("SSDK_CHANNELS_LAST", "0")))
    return ap.parse_args()


def scan_traffic_bytes():
    """HBM read bytes per scan_kernel launch measured by the separate `rocprofv3 --pmc FETCH_SIZE` pass of the same
    command (profiles/r01_pmc_fetch_size_*.csv: raw KB x 1024 x 2, the gfx950 correction of MI355X_MICROARCH.md), or
    None when no such profile is committed.  PMC collection cannot run inside the timed region."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_size_*.csv")))
    if not files:
        return None
    for row in csv.DictReader(open(files[-1])):
        if "scan_kernel" in row["kernel"]:
            return int(float(row["bytes_per_dispatch_x2_gfx950_correction"]))
    return None


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from ssds import _native as N
    from ssds.core import config
    from ssds.modeling import model_builder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
    n_gpus = world

    cfg = config.cfg_from_file(args.cfg)
    torch.manual_seed(1234)  # same random-init weights on every rank (reference init, conf bias -log 99)
    model = model_builder.create_model(cfg.MODEL).eval()
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if (rank == 0 and args.cpu_sample) else None
    model = model.to(dev, torch.bfloat16)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    anchors = model_builder.create_anchors(cfg.MODEL, model, cfg.MODEL.IMAGE_SIZE)
    decoder = model_builder.create_decoder(cfg.POST_PROCESS)
    if args.tail_stream and not args.graph:
        decoder.enable_tail_stream()
    H, W = cfg.MODEL.IMAGE_SIZE
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.rand((B, 3, H, W), device=dev, generator=g).to(torch.bfloat16)  # synthetic, resident in HBM
    if args.channels_last:
        x = x.contiguous(memory_format=torch.channels_last)

    @torch.no_grad()
    def step():
        loc, conf = model(x)
        return decoder(loc, conf, anchors)

    if args.graph:
        from ssds.utils.graph import GraphedInference

        graphed = GraphedInference(model, decoder, anchors, x)
        eager_step = step

        def step():  # noqa: F811 -- the timed step is one hipGraphLaunch (the input is the static, resident batch)
            return graphed(graphed.static_x)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    N.set_profiling(not args.graph)  # event ring: recorded inside the timed region (not capturable: eager only)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel times recorded live in the timed region -------------------------------------------
    nprof = min(args.steps, 256)
    if args.graph:  # per-kernel events cannot be recorded inside a captured graph: a few eager steps afterwards
        N.set_profiling(True)
        nprof = 5
        for _ in range(nprof):
            eager_step()
        torch.cuda.synchronize(dev)
    tim = np.array([N.timings_ms(i) for i in range(nprof)], dtype=np.float64)  # [steps, (scan, level, nms)]
    scan_ms, level_ms, nms_ms = tim.mean(0)
    N.set_profiling(False)

    @torch.no_grad()
    def time_fn(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    if args.tail_stream and not args.graph:
        decoder.disable_tail_stream()  # the stage times below are measured un-overlapped on one stream
    with torch.no_grad():
        loc, conf = model(x)
    fwd_ms = time_fn(lambda: model(x), max(3, min(10, args.steps)))
    dec_ms = time_fn(lambda: decoder(loc, conf, anchors), max(3, min(20, args.steps)))

    # ---- per-layer table (separate, untimed pass: one hipEvent per op of the recorded plan) ------------------
    layers, heads, body = None, None, None
    plan = model._plan(x) if hasattr(model, "_plan") else None
    if plan is not None and not isinstance(plan, str):
        N.lib.ssdk_set_op_profiling(1)
        acc = None
        with torch.no_grad():
            for _ in range(5):
                model(x)
                torch.cuda.synchronize(dev)
                t = N.op_timings()
                acc = [a + b[1] for a, b in zip(acc, t)] if acc else [b[1] for b in t]
        N.lib.ssdk_set_op_profiling(0)
        names = [k for k, _ in t]
        layers = []
        for row, kern, ms5 in zip(plan.layer_table(), names, acc):
            ms = ms5 / 5.0
            layers.append({"layer": row["name"], "kernel": kern.replace("_kernel", ""), "us": round(ms * 1e3, 1),
                           "TFLOPs": round(row["flops"] / (ms * 1e-3) / 1e12, 1),
                           "GBps": round(row["bytes"] / (ms * 1e-3) / 1e9, 0), "kind": row["kind"]})
        hl = [(r, l) for r, l in zip(plan.layer_table(), layers) if r["kind"] == "head"]
        h_flops = sum(r["flops"] for r, _ in hl)
        h_ms = sum(l["us"] for _, l in hl) * 1e-3
        bl = [(r, l) for r, l in zip(plan.layer_table(), layers) if r["kind"] in ("mbconv", "conv", "dw", "gconv", "stem", "pool", "fuse")]
        b_ms = sum(l["us"] for _, l in bl) * 1e-3
        b_bytes, b_flops = sum(r["bytes"] for r, _ in bl), sum(r["flops"] for r, _ in bl)
        body = {"ms": round(b_ms, 4), "algorithmic_bytes": b_bytes, "flops": b_flops,
                "achieved_GBps": round(b_bytes / (b_ms * 1e-3) / 1e9, 1), "hbm_frac": round(b_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "achieved_TFLOPs": round(b_flops / (b_ms * 1e-3) / 1e12, 1),
                "note": "backbone + extras (every op of the plan that is not a head): the part of the step that dominates "
                        "by time; bytes = each op's input + output + weights once"}
        heads = {"flops": h_flops, "ms": round(h_ms, 4), "achieved_TFLOPs": round(h_flops / (h_ms * 1e-3) / 1e12, 1),
                 "peak_TFLOPs": MFMA_PEAK_TFLOPS, "frac": round(h_flops / (h_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                 "note": "loc|conf 3x3 head convs of all levels (one fused GEMM per level), bf16 MFMA dense peak"}

    conf_bytes = sum(c.numel() * c.element_size() for c in conf)  # the scan kernel reads conf exactly once
    loc_bytes = sum(l.numel() * l.element_size() for l in loc)
    K, D, L = decoder.top_n_per_level, decoder.top_n, len(conf)
    stage_bytes = conf_bytes + loc_bytes + B * (2 * 24 * L * K + 24 * D)  # SURVEY.md 8d (1.465 MB/img)
    scan_gbs = conf_bytes / (scan_ms * 1e-3) / 1e9
    roofline = {
        "kernel": "ssdk::scan_kernel<bf16> (threshold + exact top-k over the conf tensors, one pass)",
        "bound": "hbm",
        "achieved": round(scan_gbs, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(scan_gbs / HBM_PEAK_GBS, 4),
        "traffic": scan_traffic_bytes(),  # HBM bytes per launch from the separate PMC pass (FETCH_SIZE x2), profiles/
        "algorithmic_bytes_per_launch": int(conf_bytes),
        "avg_launch_ms": round(float(scan_ms), 5),
        "decode_nms_stage": {
            "algorithmic_bytes": int(stage_bytes),
            "kernels_ms": {"scan": round(float(scan_ms), 5), "level": round(float(level_ms), 5),
                           "nms": round(float(nms_ms), 5)},
            "achieved_GBps": round(stage_bytes / ((scan_ms + level_ms + nms_ms) * 1e-3) / 1e9, 1),
            "frac": round(stage_bytes / ((scan_ms + level_ms + nms_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
    }

    result = OrderedDict()
    name = "%s-%s@%d" % (cfg.MODEL.SSDS.upper().replace("SSD", "SSD", 1), cfg.MODEL.NETS, H)
    is_headline = os.path.basename(args.cfg) == "ssd_mobilenetv2_512.yml"
    result["metric"] = "images/sec (fwd+decode+NMS) " + ("SSD-MobileNetV2@512" if is_headline else name)
    result["value"] = round(n_gpus * B * args.steps / elapsed, 2)
    result["unit"] = "images/sec"
    result["n_gpus"] = n_gpus
    result["steps"] = args.steps
    result["warmup"] = args.warmup
    result["ms_per_step"] = round(elapsed / args.steps * 1e3, 4)
    result["higher_is_better"] = True
    result["scaling"] = "weak"
    result["vs_baseline"] = None  # the reference publishes no numbers
    result["dtype"] = "bf16"
    result["data"] = "synthetic"
    result["config"] = {
        "workload": ("SSD+MobileNetV2" if is_headline else name) + " @%dx%d bf16, batch %d per GPU: backbone+neck+heads "
                    "forward, decode (thr .01, 300/level, rescore) + DIoU-NMS (.6, 100 dets); random-init weights "
                    "(reference init), torch.rand images resident in HBM" % (H, W, B),
        "cfg": os.path.relpath(args.cfg, ROOT),
        "batch_per_gpu": B,
        "global_batch": B * n_gpus,
        "image_size": [H, W],
        "parallelism": "replicas x%d (no collective)" % n_gpus,
        "fused_head_conv": os.environ.get("SSDK_FUSED_CONV", "1") != "0",
        "hipgraph": bool(args.graph),
        "decode_tail_stream": bool(args.tail_stream and not args.graph),
        "channels_last": bool(args.channels_last),
    }
    result["roofline"] = roofline
    result["stages"] = {"forward_ms": round(fwd_ms, 4), "decode_nms_ms": round(dec_ms, 4)}
    if heads is not None:
        result["roofline"]["head_convs_mfma"] = heads
        result["roofline"]["backbone_by_time"] = body
    if layers is not None and args.layers:
        result["layers"] = layers

    # ---- CPU baseline (rank 0, N = 1): torch fp32 forward + numpy oracle decoder, bounded sample ----------
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import box_oracle as O  # checker / baseline only

        S = args.cpu_sample
        config.reset_cfg()
        cfg2 = config.cfg_from_file(args.cfg)
        cpu_model = model_builder.create_model(cfg2.MODEL).eval()
        cpu_model.load_state_dict(cpu_state)
        xs = x[:S].float().cpu()
        oanch = OrderedDict((k, v.numpy()) for k, v in anchors.items())
        odec = O.Decoder(decoder.conf_threshold, decoder.nms_threshold, decoder.top_n, decoder.top_n_per_level,
                         decoder.rescore, decoder.use_diou)
        with torch.no_grad():
            cpu_model(xs[:1])  # warm
            t0 = time.perf_counter()
            cl, cc = cpu_model(xs)
            odec([t.numpy() for t in cl], [t.numpy() for t in cc], oanch)
            cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(S / cpu_s, 3),
            "unit": "images/sec",
            "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": "%d of the %d images of one batch: torch fp32 CPU forward of the same module (%d threads) "
                      "+ numpy oracle decode+NMS (1 thread), %.1f s" % (S, B, torch.get_num_threads(), cpu_s),
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
