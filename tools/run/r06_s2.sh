# round 6, session 2: the native 1x1 training kernels (ssdk_pwtrain.hip) -- parity, whole-step gradients, step time A/B -- and
# the pooled small-level rule on the fixture / bench-size network tests
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt gpurun_out/net_report.txt
( timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "pointwise" 2>&1 | tail -15 ) > $OUT/t_pw.log 2>&1; tail -15 $OUT/t_pw.log
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "whole_step" 2>&1 | tail -15 ) > $OUT/t_grad.log 2>&1; tail -15 $OUT/t_grad.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
for v in 1 0; do
  echo "== SSDK_PW_NATIVE=$v"
  SSDK_PW_NATIVE=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_native$v.json
done
( timeout 2400 python -m pytest tests/test_gpu_nets.py tests/test_gpu_bench_sizes.py -q -x -k "plan_matches_reference or forward_at_bench_size" 2>&1 | tail -15 ) > $OUT/t_nets.log 2>&1; tail -15 $OUT/t_nets.log
grep -A3 "small levels pooled" gpurun_out/net_report.txt | head -60
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
tail -1 $OUT/prof_log.txt
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/train_kernel_split.txt
head -60 $OUT/train_kernel_split.txt
rm -rf $OUT/tr
