# round 6, session 13: a 36-deep weight ring in the small-map head kernels (SSDK_CONV_SMALLMAP_RING=2) -- parity + A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s13; mkdir -p $OUT
( SSDK_CONV_SMALLMAP_RING=2 timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_plan_audit.py -q -x -k "not soak" 2>&1 | tail -4 ) > $OUT/t_conv.log 2>&1; tail -4 $OUT/t_conv.log
for v in 1 2 1 2; do
  SSDK_CONV_SMALLMAP_RING=$v timeout 400 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_ring$v.json 2> $OUT/bench_ring$v.err
  python - <<PY
import json
d=json.loads(open('$OUT/bench_ring$v.json').read().strip().splitlines()[-1])
h=d['roofline'].get('head_convs_mfma')
print('RING=$v', d['value'], d['ms_per_step'], d.get('verified'), 'heads frac', h['frac'], h['ms'], [(l['kernel'], l['us']) for l in d.get('layers') or [] if l['kind']=='head'])
PY
done
