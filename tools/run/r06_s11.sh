cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s11; mkdir -p $OUT
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -k "statistics" 2>&1 | grep -E "^E  |passed|failed|FAILED" | head -30 ) > $OUT/t_stats.log 2>&1; cat $OUT/t_stats.log
