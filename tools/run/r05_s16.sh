# round 5, session 16: the stem block's dword image loads (SSDK_STEM_DWORD=0 | 1)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s16; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "stem or flow" 2>&1 | tail -8 ) > $OUT/t_conv.log 2>&1; tail -4 $OUT/t_conv.log
( timeout 900 python -m pytest tests/test_gpu_plan_audit.py tests/test_gpu_bench_sizes.py -q -x -k "ssd or audit" 2>&1 | tail -5 ) > $OUT/t_audit.log 2>&1; tail -3 $OUT/t_audit.log
for v in 0 1 0 1; do
  SSDK_STEM_DWORD=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
    print('DWORD=$v', d['value'], d['ms_per_step'], d['verified'], [ (r['kernel'], r['us']) for r in d['layers'][:3]])
except Exception as e:
    print('DWORD=$v FAILED', e); print(open('$OUT/bench_$v.err').read()[-800:])
PY
done
timeout 400 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -1 $OUT/bench_full.json | cut -c1-200; tail -2 $OUT/bench_full.err
