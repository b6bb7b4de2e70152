# round 6, session 52: head pair test with the small levels' weight gradients on the kernels (the SSDK_CONV3_NATIVE=2 configuration)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair_conv" 2>&1 | grep -E "passed|failed|Error|assert|rel err" | tail -6
SSDK_CONV3_NATIVE=2 timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130
