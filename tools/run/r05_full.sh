# round 5: the whole GPU suite + smoke() on one box
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/full; mkdir -p $OUT
( timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 ) > $OUT/t_gpu.log 2>&1; tail -6 $OUT/t_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
cp gpurun_out/net_report.txt $OUT/ 2>/dev/null; true
