cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -25 ) > $OUT/t_train.log 2>&1; tail -25 $OUT/t_train.log | cut -c1-220
for v in 1 0 1 0; do
  echo "== SSDK_BN_STATS_FUSED=$v"
  SSDK_BN_STATS_FUSED=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_stats$v.json
done
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --size 300 2>&1 | tail -1 | tee $OUT/train_step_300.json
