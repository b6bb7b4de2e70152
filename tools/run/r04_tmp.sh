cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_conv.py tests/test_gpu_bench_sizes.py -x -q -m gpu 2>&1 | tail -8
