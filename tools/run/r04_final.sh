# round-end validation: every GPU test, smoke (the three configurations' bench lines: tools/run/profile_r04.sh)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04final; mkdir -p $OUT
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > $OUT/all.log 2>&1; tail -40 $OUT/all.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
