# round-end validation: every GPU test, smoke, the three configs' bench lines
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03final; mkdir -p $OUT
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
tail -2 $OUT/pytest.log; tail -1 $OUT/smoke.log
bash tools/run/r03_cfgs.sh
