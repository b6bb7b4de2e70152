# training-step kernels: parity tests + step A/B (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3
SSDK_FUSE_BN_ACT=0 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
SSDK_FUSE_BN_ACT=1 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
