# round 6, session 3: the fixed-order parallel reduce of the 1x1 weight gradient, per-layer table of the native 1x1 kernels,
# the K split over two wave groups in conv_smallmap (A/B), whole-step gradients with two fp32 floors
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s3; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt gpurun_out/net_report.txt
( timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "pointwise or whole_step" 2>&1 | tail -15 ) > $OUT/t_pw.log 2>&1; tail -15 $OUT/t_pw.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
timeout 600 python tools/pw_probe.py 2>&1 | tail -30 | tee $OUT/pw_probe.txt
for v in 1 0; do
  echo "== SSDK_PW_NATIVE=$v"
  SSDK_PW_NATIVE=$v timeout 300 python tools/bench_train.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/train_step_native$v.json
done
( timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_plan_audit.py -q -x -k "not soak" 2>&1 | tail -6 ) > $OUT/t_conv.log 2>&1; tail -6 $OUT/t_conv.log
for v in 2 1; do
  echo "== SSDK_CONV_SMALLMAP_KW=$v"
  SSDK_CONV_SMALLMAP_KW=$v timeout 400 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_kw$v.json 2> $OUT/bench_kw$v.err
  python - <<PY
import json
d=json.loads(open('$OUT/bench_kw$v.json').read().strip().splitlines()[-1])
print('KW=$v', d['value'], d['ms_per_step'], d.get('verified'), d['roofline'].get('head_convs_mfma'))
for l in d.get('layers') or []:
    if l['kind'] in ('head',) or 'extra' in l['layer']: print('   ', l['layer'], l['kernel'], l['us'])
PY
done
( timeout 1500 python -m pytest tests/test_gpu_bench_sizes.py -q -x -k "forward_at_bench_size" 2>&1 | tail -6 ) > $OUT/t_bs.log 2>&1; tail -6 $OUT/t_bs.log
