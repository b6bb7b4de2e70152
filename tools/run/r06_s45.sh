# round 6, session 45: stem kernels with the next tile's loads requested before the current one is multiplied -- parity, times
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s45; rm -rf $OUT; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "stem_conv" 2>&1 | grep -E "passed|failed|Error|assert|rel err|outside|dweight" | tail -8 ) > $OUT/t.log 2>&1; cat $OUT/t.log
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130; }
run; run
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 88 400 > $OUT/split.txt
rm -rf $OUT/tr; grep stem_ $OUT/split.txt
