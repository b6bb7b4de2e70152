# round 6, session 47: tests/test_gpu_train.py as a whole (the 512 px whole-step case failed once inside the full suite and passes alone)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s47; rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_train.py -q 2>&1 | grep -v "^$" > $OUT/t.log; grep -E "passed|failed" $OUT/t.log | tail -3; grep -n "plan rel\|pooled\|structurally\|AssertionError" $OUT/t.log | head -20
