# round 6, session 51: no materialised zero gradients for the statistics outputs, depthwise fp32 master weights cast outside autograd
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s51; rm -rf $OUT; mkdir -p $OUT
( timeout 2400 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 ) > $OUT/t.log 2>&1; cat $OUT/t.log
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | cut -c60-130; }
run; run; run --graph 1
timeout 600 python tools/train_glue_probe.py 2>/dev/null | tail -25
