# round 6, session 27: the deferred BatchNorm's backward sums from the depthwise input-gradient kernel (SSDK_BN_BWD_SUMS) -- parity, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s27; mkdir -p $OUT
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x -k "deferred or whole_step or batchnorm or depthwise or dwconv" 2>&1 | tail -8 ) > $OUT/t_train.log 2>&1; tail -8 $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_BN_BWD_SUMS=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_bs$v.json 2> $OUT/train_bs$v.err
  tail -1 $OUT/train_bs$v.json | cut -c1-200
done
