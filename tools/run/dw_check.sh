# depthwise training kernels: parity, then per-layer times and the step A/B against the tiled kernels (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_train.py -q -k depthwise --tb=line 2>&1 | tail -8
SSDK_DW_PLANE=1 timeout 200 python tools/dw_probe.py 2>&1 | tail -19
SSDK_DW_PLANE=0 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
SSDK_DW_PLANE=1 timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1
