# round 6, session 55: the driver's multi-GPU launch line with one rank (RCCL process group, barrier, MAX reduce) on a 1-GPU box
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-260
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 tools/bench_train.py --gpus 1 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
