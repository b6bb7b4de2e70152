# pointwise GEMM convs: parity test + training step A/B (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_train.py -q -x -k "pointwise or model_with_loss" 2>&1 | tail -3
SSDK_PW_GEMM=0 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
SSDK_PW_GEMM=1 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
