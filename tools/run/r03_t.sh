cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "large_tile or short_k or split_heads" 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -30
timeout 100 python tools/gemm_probe.py head_L0 head_L1 tower_P3 tower_P4 2>&1 | grep -E "head_|tower_" | cut -c1-100
echo "== halo"; SSDK_CONV3X3_SHORT=2 timeout 100 python tools/gemm_probe.py head_L1 tower_P3 tower_P4 2>&1 | grep -E "head_|tower_" | cut -c1-100
