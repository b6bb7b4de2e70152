# round 6, session 7: v_permlane16_swap semantics, the paired 16-byte-store epilogue of conv3x3_short (parity + A/B), rest of the training tests
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s7; mkdir -p $OUT
rm -f gpurun_out/whole_step_gradients.txt
./tools/micro/swap16
( timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_plan_audit.py -q -x -k "not soak" 2>&1 | tail -6 ) > $OUT/t_conv.log 2>&1; tail -6 $OUT/t_conv.log
for v in 1 0 1 0; do
  SSDK_S3_WIDE=$v timeout 400 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_wide$v.json 2> $OUT/bench_wide$v.err
  python - <<PY
import json
d=json.loads(open('$OUT/bench_wide$v.json').read().strip().splitlines()[-1])
h=d['roofline'].get('head_convs_mfma')
print('WIDE=$v', d['value'], d['ms_per_step'], d.get('verified'), 'heads frac', h['frac'], h['ms'], [ (l['kernel'], l['us']) for l in d.get('layers') or [] if l['kind']=='head'][:2])
PY
done
( timeout 1800 python -m pytest tests/test_gpu_train.py -q -x -k "multibox_path or eval_epoch or whole_step or counters" 2>&1 | tail -8 ) > $OUT/t_train.log 2>&1; tail -8 $OUT/t_train.log
grep -E "rows, median" gpurun_out/whole_step_gradients.txt
