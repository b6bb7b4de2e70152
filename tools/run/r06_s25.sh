# round 6, session 25: input gradient of the head pairs on the inference kernels (SSDK_HEAD_PAIR_DGRAD) -- parity, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s25; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair or whole_step" 2>&1 | tail -6 ) > $OUT/t_train.log 2>&1; tail -6 $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_HEAD_PAIR_DGRAD=$v  # (the switch existed for this A/B only) timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_hd$v.json 2> $OUT/train_hd$v.err
  tail -1 $OUT/train_hd$v.json | cut -c1-200
done
