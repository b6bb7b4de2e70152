# round 6, session 6: LDS-tiled weight gradient, BatchNorm counters in one launch, both SGD implementations through the step tests
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s6; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -12 ) > $OUT/t_train.log 2>&1; tail -12 $OUT/t_train.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
timeout 600 python tools/pw_probe.py 2>&1 | tail -26 | tee $OUT/pw_probe.txt
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step.json
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
tail -1 $OUT/prof_log.txt
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/train_kernel_split.txt
head -50 $OUT/train_kernel_split.txt
rm -rf $OUT/tr
