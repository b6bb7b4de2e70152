# round 6, session 28: the head pairs' output-gradient buffer by ssdk_concat_nchw_to_nhwc -- parity, timing
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s28; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair or whole_step" 2>&1 | tail -4 ) > $OUT/t_train.log 2>&1; tail -4 $OUT/t_train.log
for v in 1 2; do
  timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_$v.json 2> $OUT/train_$v.err
  tail -1 $OUT/train_$v.json | cut -c1-200
done
