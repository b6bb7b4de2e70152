# round 6, session 41: stem tests again; kernel split of the all-native step (where the extras' +0.55 ms goes)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s41; rm -rf $OUT; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "stem_conv or whole_step" 2>&1 | grep -E "passed|failed|Error|assert|rel err|outside|dweight" | tail -8 ) > $OUT/t.log 2>&1; cat $OUT/t.log
cd /tmp && export TMPDIR=/tmp
for mode in mixed native; do
  if [ $mode = native ]; then export SSDK_CONV3_NATIVE=2 SSDK_HEAD_WGRAD_MIN=0; fi
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$mode -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/log_$mode.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr_$mode/*/*kernel_trace.csv | head -1) 88 400 > $OUT/split_$mode.txt
  rm -rf $OUT/tr_$mode
done
