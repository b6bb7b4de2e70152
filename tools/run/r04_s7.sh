# round 4 session 7: the new loss / box tests first, then the whole GPU suite
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s7; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_box.py -x -q -m gpu -k "multibox or golden or fused_match" 2>&1 | tail -25 ) > $OUT/new.log 2>&1; tail -25 $OUT/new.log
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > $OUT/all.log 2>&1; tail -25 $OUT/all.log
