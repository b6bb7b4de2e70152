"""Training-step throughput (BASELINE config 4: SSD+MobileNetV2@512 DDP step, synthetic COCO-shaped targets).
    python tools/bench_train.py --steps 10            (1 GPU)
    python tools/bench_train.py --gpus N              (re-executes itself under torch.distributed.run with N ranks)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --gpus N
Prints one JSON line on rank 0 (images/sec of the full step: fwd, target assignment, loss, bwd, all-reduce,
optimizer)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
import torch.distributed as dist

from ssds.core import config
from ssds.dataset.synthetic import SyntheticDetectionLoader
from ssds.modeling import model_builder
from ssds.pipeline.pipeline_anchor_ddp import train_step
from ssds.utils.train_ddp import Solver

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--channels-last", type=int, default=0)
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--graph", type=int, default=0, help="1: the whole step as one captured hipGraph (single GPU)")
ap.add_argument("--size", type=int, default=0, help="square input size instead of the config's (300: planes that are not a multiple of 8)")
ap.add_argument("--cpu", type=int, default=0,
                help="(tests/test_ddp_cpu.py) 1: ONLY the launcher / rank / barrier / MAX-time / rank-0-print logic of this "
                     "script on CPU under gloo, with a stub step (sleep + one real all-reduce); no model, never a measurement "
                     "(the training step itself has no CPU path: its kernels need a HIP device)")
args = ap.parse_args()
if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: become N ranks on this node
    import socket

    _s = socket.socket()
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
    _s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                              "--master-addr", "127.0.0.1", "--master-port", str(_port), os.path.abspath(__file__)] + sys.argv[1:])
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
lr = int(os.environ.get("LOCAL_RANK", "0"))
if args.cpu:
    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
else:
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
cfg = config.cfg_from_file(os.path.join(ROOT, "experiments", "cfgs", "ssd_mobilenetv2_512.yml"))
cfg.TRAIN.BATCH_SIZE = args.batch
if args.size:
    cfg.MODEL.IMAGE_SIZE = [args.size, args.size]
cfg.EXP_DIR = "/tmp/ssdk_bench_train"
torch.manual_seed(1234)
if args.cpu:
    _grad = torch.ones(1 << 16)

    def train_step(*_a):  # noqa: F811 -- stub: rank r "computes" for 2 (r + 1) ms, then one real (gloo) all-reduce
        time.sleep(0.002 * (rank + 1))
        if world > 1:
            dist.all_reduce(_grad)
        return 0.0, 0.0, False

    mwl = images = targets = anchors = None
    solver = argparse.Namespace(optimizer=None)
else:
    solver = Solver(cfg, lr, dev)
    if args.channels_last:
        solver.model.to(memory_format=torch.channels_last)
    mwl = solver.wrap()
    mwl.train()
    inner = mwl.module.model if hasattr(mwl, "module") else mwl.model
    anchors = model_builder.create_anchors(cfg.MODEL, inner, cfg.MODEL.IMAGE_SIZE)
    mwl.train()
    loader = SyntheticDetectionLoader(args.batch, cfg.MODEL.IMAGE_SIZE, cfg.MODEL.NUM_CLASSES, 1, dev, seed=1234 + rank)
    images, targets = loader.batch()
    if args.channels_last:
        images = images.contiguous(memory_format=torch.channels_last)


def sync():
    if world > 1:
        dist.barrier(device_ids=None if args.cpu else [lr])
    if not args.cpu:
        torch.cuda.synchronize()


if args.graph and not args.cpu and world == 1:
    from ssds.pipeline.pipeline_anchor_ddp import GraphedTrainStep

    _graphed = GraphedTrainStep(mwl, images, targets, anchors, solver.optimizer)

    def train_step(_m, im, tg, _a, _o):  # noqa: F811 -- one hipGraphLaunch per step
        return _graphed(im, tg)

for _ in range(args.warmup):
    train_step(mwl, images, targets, anchors, solver.optimizer)
sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    c, l, sk = train_step(mwl, images, targets, anchors, solver.optimizer)
sync()
el = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t)
if rank == 0:
    print(json.dumps({"metric": "images/sec (DDP training step) SSD-MobileNetV2@%d" % cfg.MODEL.IMAGE_SIZE[0], "value": round(world * args.batch * args.steps / el, 1),
                      "n_gpus": world, "ms_per_step": round(el / args.steps * 1e3, 2), "batch_per_gpu": args.batch,
                      "cls_loss": float(c), "loc_loss": float(l), "dtype": "bf16 autocast", "hipgraph": bool(args.graph),
                      "data": "synthetic" if not args.cpu else "stub (CPU / gloo run of the rank logic)"}))
if world > 1:
    dist.destroy_process_group()
