# Round-end validation on one MI355X box (run through gpurun): GPU test-suite, smoke(), the default bench line with the
# per-layer table, the other two inference configs, rocprofv3 kernel stats and the FETCH_SIZE pass of the same command.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/v7
mkdir -p $OUT
( time timeout 500 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
timeout 200 python bench.py --layers 1 > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 0 --steps 10 --layers 1 > $OUT/bench_fpn.json 2>> $OUT/bench.err
timeout 200 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --cpu-sample 0 --steps 10 --layers 1 > $OUT/bench_bifpn.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $OUT/pytest.log; tail -2 $OUT/smoke.log; head -c 300 $OUT/bench.json; echo; head -c 200 $OUT/bench_fpn.json; echo; head -c 200 $OUT/bench_bifpn.json; echo; cat $OUT/bench.err | tail -3
