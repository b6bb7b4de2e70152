# round 6, session 4: 3x3 training convolutions on the ssdk kernels (im2col + pw kernels + col2im), whole-step gradients, step A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s4; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
python - <<'PY'
import torch
torch.manual_seed(0)
a = torch.randn(4, 512, 25, device="cuda"); b = torch.randn(128, 512, device="cuda")
for name, fn in (("matmul", lambda: torch.matmul(b, a)), ("bmm", lambda: torch.bmm(b.expand(4, -1, -1).contiguous(), a))):
    got = fn(); want = torch.matmul(b.double(), a.double())
    print(name, "fp32 vs fp64 rel rms", float((got.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()),
          "allow_tf32", torch.backends.cuda.matmul.allow_tf32, getattr(torch.backends.cuda.matmul, "fp32_precision", None))
PY
( timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "pointwise or conv3x3" 2>&1 | tail -15 ) > $OUT/t_pw.log 2>&1; tail -15 $OUT/t_pw.log
( timeout 900 python -m pytest tests/test_gpu_train.py -q -k "whole_step" 2>&1 | tail -30 ) > $OUT/t_grad.log 2>&1; tail -30 $OUT/t_grad.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
for v in 1 0; do
  echo "== SSDK_CONV3_NATIVE=$v"
  SSDK_CONV3_NATIVE=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_conv3_$v.json
done
timeout 300 python tools/bench_train.py --steps 20 --warmup 5 --graph 1 2>&1 | tail -1 | tee $OUT/train_step_graph.json
( timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_plan_audit.py tests/test_gpu_train.py -q -x -k "not soak and not whole_step" 2>&1 | tail -6 ) > $OUT/t_conv.log 2>&1; tail -6 $OUT/t_conv.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
tail -1 $OUT/prof_log.txt
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/train_kernel_split.txt
head -45 $OUT/train_kernel_split.txt
rm -rf $OUT/tr
