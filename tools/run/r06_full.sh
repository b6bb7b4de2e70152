# the driver's round-end checks, as the driver runs them: every GPU test, then smoke(), then the default bench line
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/full; mkdir -p $OUT
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $OUT/t_all.log 2>&1; tail -15 $OUT/t_all.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
