# round 5, session 12: mbflow row pairs (block 3) A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "flow or mbconv or block or register" 2>&1 | tail -5 ) > $OUT/t_conv.log 2>&1; tail -3 $OUT/t_conv.log
for v in 0 1; do
  SSDK_FLOW_PAIR=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
    print('PAIR=$v', d['value'], d['ms_per_step'], d['verified'], [ (r['kernel'], r['us']) for r in d['layers'][:4]])
except Exception as e:
    print('PAIR=$v FAILED', e); print(open('$OUT/bench_$v.err').read()[-800:])
PY
done
