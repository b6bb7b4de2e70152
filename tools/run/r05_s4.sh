# round 5, session 4: the generalised row-pair block kernel (stride 2, 32-wide maps)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s4; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "row_pair" 2>&1 | tail -25 ) > $OUT/t_mbk.log 2>&1; tail -12 $OUT/t_mbk.log
for v in "SSDK_MBK=2" "SSDK_MBK=1" "SSDK_MBK=6" "SSDK_MBK=34" "SSDK_MBK=26"; do
  tag=$(echo $v | tr '= ' '__')
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --layers 1 --cpu-sample 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1])
    print('$v', d['value'], d['ms_per_step'], d['stages'])
    for r in d['layers']:
        if ('@16x16' in r['layer'] or '@32x32' in r['layer']) and 'mbconv' in r['layer']: print('   %-40s %-30s %7.1f' % (r['layer'], r['kernel'], r['us']))
except Exception as e:
    print('$v', 'FAILED', e)
PY
done
SSDK_MB_DBG=1 timeout 200 python tools/mb_dbg.py 2>&1 | grep "mbk dbg" | tail -10
( timeout 600 python -m pytest tests/test_gpu_plan_audit.py tests/test_gpu_bench_sizes.py -q -k "ssd" 2>&1 | tail -12 ) > $OUT/t_audit.log 2>&1; tail -5 $OUT/t_audit.log
cp gpurun_out/plan_audit_ssd*.txt $OUT/ 2>/dev/null
