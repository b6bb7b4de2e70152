# round 6, session 5: native SGD, wgrad load mapping + slice rule (per-layer table), whole-step gradients (4 variants), step time
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s5; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -12 ) > $OUT/t_train.log 2>&1; tail -12 $OUT/t_train.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
timeout 600 python tools/pw_probe.py 2>&1 | tail -26 | tee $OUT/pw_probe.txt
for v in 1 0; do
  echo "== SSDK_SGD_NATIVE=$v"
  SSDK_SGD_NATIVE=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_sgd$v.json
done
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --size 300 2>&1 | tail -1 | tee $OUT/train_step_300.json
