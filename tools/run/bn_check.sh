# training kernels: parity, then the 300 px step (planes that are not a multiple of 8) with the per-plane BatchNorm
# kernels and with the default choice (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_train.py -q --tb=line 2>&1 | tail -3
SSDK_BN_FLAT=0 timeout 200 python tools/bench_train.py --size 300 --steps 6 --warmup 3 2>&1 | tail -1
timeout 200 python tools/bench_train.py --size 300 --steps 6 --warmup 3 2>&1 | tail -1
