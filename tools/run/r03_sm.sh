cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-200 | head
bash tools/run/r03_cfgs.sh
