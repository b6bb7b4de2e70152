# Candidates written after the round-2 GPU budget was spent (run through gpurun, one call):
#   SSDK_BN_FLAT=3  second flat BatchNorm apply pass (uniform coefficients per workgroup) -- parity first, then the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
SSDK_BN_FLAT=3 timeout 300 python -m pytest tests/test_gpu_train.py -q -k batchnorm --tb=line 2>&1 | tail -3
for v in 2 3 2 3; do
  echo "SSDK_BN_FLAT=$v"
  SSDK_BN_FLAT=$v timeout 200 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1 | cut -c1-150
done
for v in 2 3; do
  echo "SSDK_BN_FLAT=$v @300"
  SSDK_BN_FLAT=$v timeout 200 python tools/bench_train.py --size 300 --steps 6 --warmup 3 2>/dev/null | tail -1 | cut -c1-150
done
