# round 6, session 31: the focal-loss class loop of match_kernel with eight logits requested up front -- parity green, match_kernel 1 846 vs 1 831 us per four steps: no gain (the loop is VALU-bound: expf, log1pf, a division per element), not kept
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s31; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_box.py -q -x -k "loss or match or model_with_loss or training_module" 2>&1 | grep -E "passed|failed|Error" | tail -4 ) > $OUT/t.log 2>&1; cat $OUT/t.log
for v in 1 2; do
  timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_$v.json 2> $OUT/train_$v.err
  tail -1 $OUT/train_$v.json | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/split.txt
rm -rf $OUT/tr
grep -E "match_kernel|kernel time" $OUT/split.txt
