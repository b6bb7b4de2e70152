# round 6, session 8: BatchNorm statistics from the 1x1 kernels' epilogues; the switch owner tests; nets tests with the new tail rule
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s8; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -8 ) > $OUT/t_train.log 2>&1; tail -8 $OUT/t_train.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
for v in 1 0; do
  echo "== SSDK_BN_STATS_FUSED=$v"
  SSDK_BN_STATS_FUSED=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_stats$v.json
done
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
( timeout 2400 python -m pytest tests/test_gpu_box.py tests/test_gpu_nets.py tests/test_gpu_switches.py -q -x 2>&1 | tail -8 ) > $OUT/t_rest.log 2>&1; tail -8 $OUT/t_rest.log
