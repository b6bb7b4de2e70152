# depthwise training kernels: parity, then the per-layer times at three unit budgets (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_train.py -q -k depthwise --tb=line 2>&1 | tail -1
for u in 1024 512 768 1024; do echo "units $u"; SSDK_DW_UNITS=$u timeout 200 python tools/dw_probe.py 2>&1 | tail -18; done
