"""Data-parallel training entry point -- the reference's ``ssds/utils/train_ddp.py`` (Solver :31-191,
main :193-220) on torch DDP over RCCL/xGMI instead of Apex DDP/AMP/SyncBN over NCCL:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m ssds.utils.train_ddp -cfg experiments/cfgs/ssd_mobilenetv2_512.yml --steps 100

One process per GPU; ``nccl`` backend == RCCL on ROCm.  bf16 autocast replaces AMP O1 / static loss scale
128 (bf16 has fp32's exponent range: no loss scaling).  BatchNorm stays local (per-GPU batch 64; set
``--sync-bn`` for torch SyncBatchNorm).  Gradients: bucketed all-reduce (mean) overlapped with backward;
16 MB buckets keep every xGMI ring message bandwidth-bound for the 44 MB of SSD-MobileNetV2 gradients.
Data: synthetic COCO-shaped batches (ssds/dataset/synthetic.py); the reference's DALI loaders are out of
scope."""
import argparse
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from ssds.core import checkpoint, config, criterion, optimizer
from ssds.dataset.synthetic import SyntheticDetectionLoader
from ssds.modeling import model_builder
from ssds.pipeline.pipeline_anchor_ddp import ModelWithLossBasic, train_anchor_based_epoch


class Solver(object):
    """Same life cycle as the reference Solver (train_ddp.py:31-191)."""

    def __init__(self, cfg, local_rank, device, steps_per_epoch=100, sync_bn=False, render=False):
        self.cfg, self.local_rank, self.device = cfg, local_rank, device
        self.steps_per_epoch = steps_per_epoch
        if local_rank == 0:
            print("===> Building model")
        self.model = model_builder.create_model(cfg.MODEL)
        self.load_model()
        if sync_bn:
            self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.model)
        elif os.environ.get("SSDK_FAST_BN", "1") != "0":
            from ssds.modeling.layers.batchnorm import use_fast_batchnorm

            use_fast_batchnorm(self.model)  # training BN on the ssdk kernels (local statistics, like the default)
            if os.environ.get("SSDK_FUSE_BN_ACT", "1") != "0":
                from ssds.modeling.layers.batchnorm import fuse_bn_activations

                fuse_bn_activations(self.model)  # Conv-BN-ReLU6: the clamp and its gradient mask ride on the BN passes
                from ssds.modeling.layers.batchnorm import fuse_bn_into_depthwise

                fuse_bn_into_depthwise(self.model)  # expand BN (+ ReLU6) applied by the depthwise kernels on load: no apply pass
        if os.environ.get("SSDK_PW_GEMM", "1") != "0":
            from ssds.modeling.layers.pointwise import use_pointwise_gemm

            use_pointwise_gemm(self.model)  # 1x1 convolutions on the NCHW tensors (16 bit: csrc/ssdk_pwtrain.hip; fp32: library GEMMs)
            if os.environ.get("SSDK_FAST_BN", "1") != "0" and not sync_bn:
                from ssds.modeling.layers.pointwise import fuse_conv_bn_statistics

                fuse_conv_bn_statistics(self.model)  # the 1x1 kernels hand their BatchNorm the batch statistics
        from ssds.modeling.layers.pointwise import use_native_stem

        use_native_stem(self.model)  # the image-side 3x3 / stride-2 convolution: forward + weight gradient on csrc/ssdk_stemtrain.hip (A/B tools/run/r06_s42.sh: 17.0 vs 17.5 ms per step)
        from ssds.modeling.layers.headconv import use_head_pairs

        use_head_pairs(self.model)  # SSD heads: forward of each level's loc | conf pair on the inference kernels (SSDK_HEAD_PAIR=0: MIOpen)
        conv3 = os.environ.get("SSDK_CONV3_NATIVE", "2")
        from ssds.modeling.layers import headconv

        headconv.WGRAD_MIN_PIXELS = 64 if conv3 == "0" else 0
        if conv3 == "2" and hasattr(self.model, "extras"):
            # (default) NO library convolution in the step: the extras' 3x3 / stride-2 layers as im2col + ssdk_pw_* + col2im with the
            # batch folded into the GEMM's pixel dimension (64 / 16 / 4 / 1 pixels per image: pointwise.FOLD_BELOW), and the weight
            # gradients of the small head levels on ssdk_pw_wgrad the same way.  tools/run/r06_s53.sh: 16.7 vs 17.0 ms per step
            # against the library (SSDK_CONV3_NATIVE=0); before the fold the same path was 0.6 ms SLOWER than the library
            from ssds.modeling.layers.pointwise import use_native_conv3x3

            use_native_conv3x3(self.model.extras)
        if conv3 == "1":
            # EVERY 3x3 layer (the heads too) as im2col + the 1x1 kernels.  Correct (tests/test_gpu_train.py) and slow: 23.2 vs
            # 20.8 ms per step when it was measured (round 6, session 4: the streaming 1x1 kernels are the wrong shape for
            # K = 864 ... 4608 at 480 output channels; the heads run on the inference kernels instead: headconv.py)
            from ssds.modeling.layers.pointwise import use_native_conv3x3

            use_native_conv3x3(self.model)
        self.model.to(self.device)
        if render and local_rank == 0:
            print("Model architectures:\n{}\n".format(self.model))
        if local_rank == 0:
            print("Trainable scope: {}".format(cfg.TRAIN.TRAINABLE_SCOPE))
        params = optimizer.trainable_param(self.model, cfg.TRAIN.TRAINABLE_SCOPE)
        self.optimizer = optimizer.configure_optimizer(params, cfg.TRAIN.OPTIMIZER)
        self.lr_scheduler = optimizer.configure_lr_scheduler(self.optimizer, cfg.TRAIN.LR_SCHEDULER)
        self.max_epochs = cfg.TRAIN.MAX_EPOCHS
        self.cls_criterion = getattr(criterion, cfg.MATCHER.CLASSIFY_LOSS)(
            alpha=cfg.MATCHER.FOCAL_ALPHA, gamma=cfg.MATCHER.FOCAL_GAMMA, negpos_ratio=cfg.MATCHER.NEGPOS_RATIO)
        self.loc_criterion = getattr(criterion, cfg.MATCHER.LOCATE_LOSS)()

    def wrap(self):
        mwl = ModelWithLossBasic(self.model, self.cls_criterion, self.loc_criterion, self.cfg.MODEL.NUM_CLASSES,
                                 self.cfg.MATCHER.MATCH_THRESHOLD, self.cfg.MATCHER.CENTER_SAMPLING_RADIUS)
        if dist.is_initialized() and dist.get_world_size() > 1:
            ids = [self.device.index] if self.device.type == "cuda" else None
            mwl = DDP(mwl, device_ids=ids, bucket_cap_mb=16, gradient_as_bucket_view=True)
        return mwl

    def train_model(self, epochs=None):
        mwl = self.wrap()
        if self.local_rank == 0:
            print("===> Loading data (synthetic)")
        rank = dist.get_rank() if dist.is_initialized() else 0
        loader = SyntheticDetectionLoader(self.cfg.TRAIN.BATCH_SIZE, self.cfg.MODEL.IMAGE_SIZE,
                                          self.cfg.MODEL.NUM_CLASSES, self.steps_per_epoch, self.device,
                                          seed=1234 + rank)
        last = self.start_epoch + (epochs if epochs is not None else self.max_epochs)
        for epoch in range(self.start_epoch + 1, min(last, self.max_epochs) + 1):
            if self.local_rank == 0:
                sys.stdout.write("\rEpoch {epoch:d}/{max_epochs:d}:\n".format(epoch=epoch, max_epochs=self.max_epochs))
            inner = mwl.module.model if hasattr(mwl, "module") else mwl.model
            anchors = model_builder.create_anchors(self.cfg.MODEL, inner, self.cfg.MODEL.IMAGE_SIZE)
            train_anchor_based_epoch(mwl, loader, self.optimizer, anchors, epoch, self.device, self.local_rank)
            if epoch % self.cfg.TRAIN.CHECKPOINTS_EPOCHS == 0 and self.local_rank == 0 and rank == 0:
                checkpoint.save_checkpoints(inner, self.cfg.EXP_DIR, self.cfg.CHECKPOINTS_PREFIX, epoch)
            self.lr_scheduler.step()

    def load_model(self):
        previous = checkpoint.find_previous_checkpoint(self.cfg.EXP_DIR)
        if previous:
            self.start_epoch = previous[0][-1]
            checkpoint.resume_checkpoint(self.model, previous[1][-1], self.cfg.TRAIN.RESUME_SCOPE)
        else:
            self.start_epoch = 0
            if self.cfg.RESUME_CHECKPOINT:
                checkpoint.resume_checkpoint(self.model, self.cfg.RESUME_CHECKPOINT, self.cfg.TRAIN.RESUME_SCOPE)


def main(argv=None):
    parser = argparse.ArgumentParser(description="Train a ssds.pytorch network (data parallel, RCCL)")
    parser.add_argument("-cfg", "--config", dest="config_file", required=True, help="the address of config file")
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    parser.add_argument("--steps", type=int, default=100, help="synthetic steps per epoch")
    parser.add_argument("--epochs", type=int, default=1)
    parser.add_argument("--sync-bn", action="store_true")
    parser.add_argument("-r", "--render", action="store_true")
    args = parser.parse_args(argv)
    cfg = config.cfg_from_file(args.config_file)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if torch.cuda.is_available():
        torch.cuda.set_device(args.local_rank)
        device = torch.device("cuda", args.local_rank)
        backend = "nccl"  # RCCL on ROCm
    else:
        device, backend = torch.device("cpu"), "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, init_method="env://")
    solver = Solver(cfg, args.local_rank, device, steps_per_epoch=args.steps, sync_bn=args.sync_bn,
                    render=args.render)
    solver.train_model(epochs=args.epochs)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
