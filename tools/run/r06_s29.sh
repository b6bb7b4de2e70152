# round 6, session 29: weight gradients of the large head levels by im2col + ssdk_pw_wgrad (SSDK_HEAD_PAIR_WGRAD) -- parity, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s29; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair or whole_step" 2>&1 | grep -E "passed|failed|Error" | tail -4 ) > $OUT/t_train.log 2>&1; cat $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_HEAD_PAIR_WGRAD=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_hw$v.json 2> $OUT/train_hw$v.err
  tail -1 $OUT/train_hw$v.json | cut -c1-200
done
