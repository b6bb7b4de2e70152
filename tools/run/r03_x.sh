cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "short_k" -x 2>&1 | grep -E "^E  |assert|Error" | cut -c1-300 | head -20
