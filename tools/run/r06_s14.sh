# round 6, session 14: the deferred BatchNorm (fuse_bn_into_depthwise: the expand block's BN + ReLU6 applied by the depthwise
# kernels on load) -- parity tests, then A/B of the training step
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s14; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "deferred or statistics or whole_step or batchnorm or graphed" 2>&1 | tail -15 ) > $OUT/t_train.log 2>&1; tail -15 $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_BN_DEFER=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_defer$v.json 2> $OUT/train_defer$v.err
  tail -1 $OUT/train_defer$v.json | cut -c1-400
done
