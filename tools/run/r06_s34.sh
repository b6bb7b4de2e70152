# round 6, session 34: the BatchNorm finalize arithmetic in the reduction's last workgroup per channel (SSDK_BN_FOLD_FIN) -- parity, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s34; mkdir -p $OUT
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x -k "batchnorm or deferred or statistics or whole_step or graphed" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 ) > $OUT/t.log 2>&1; cat $OUT/t.log
for v in 1 0 1 0; do
  SSDK_BN_FOLD_FIN=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c1-160
done
