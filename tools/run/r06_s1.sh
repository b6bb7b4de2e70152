# round 6, session 1: the new parity tests (whole-step gradients, audit with 4 images + plan-kernel column), the
# small-level distribution probe, and the training step re-measured BEFORE this round's kernel work (r06 baseline)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s1; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "whole_step" 2>&1 | tail -25 ) > $OUT/t_grad.log 2>&1; tail -25 $OUT/t_grad.log
cat gpurun_out/whole_step_gradients.txt 2>/dev/null | grep -E "fp32|pooled|px B" | head -20
( timeout 1200 python -m pytest tests/test_gpu_plan_audit.py -q -x 2>&1 | tail -6 ) > $OUT/t_audit.log 2>&1; tail -6 $OUT/t_audit.log
sed -n 1,40p gpurun_out/plan_audit_ssd_mobilenetv2_512_bfloat16.txt
for dt in bfloat16 float16; do
  timeout 600 python tools/small_level_probe.py --dtype $dt --batch 8 --seeds 6 2>&1 | tail -16
done
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step.json
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
tail -1 $OUT/prof_log.txt
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/train_kernel_split.txt
head -50 $OUT/train_kernel_split.txt
rm -rf $OUT/tr
