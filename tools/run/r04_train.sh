cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04train; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "graphed or skipped_on" 2>&1 | tail -25 ) > $OUT/pytest_train.log 2>&1
tail -12 $OUT/pytest_train.log
