cd $GRAFT_REPO_ROOT
for d in 0 8 1; do echo "== DBG=$d"; SSDK_S3_DBG=$d timeout 100 python tools/gemm_probe.py head_L0 2>&1 | grep head_L0 | cut -c1-60; done
