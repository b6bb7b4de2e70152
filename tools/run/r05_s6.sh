# round 5, session 6: mbk instances for blocks 5-7 (64-wide maps); A/B per instance
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s6; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "row_pair" 2>&1 | tail -25 ) > $OUT/t_mbk.log 2>&1; tail -6 $OUT/t_mbk.log
# instance bits: 2 = 16x16 s1, 4 = 32->16 s2, 8 / 16 = 64->384->64 | 96, 32 = 96->576->96, 64 = block 7 (64->32 s2), 128 = blocks 5-6 (64x64)
for v in "SSDK_MBK=1" "SSDK_MBK=1 SSDK_MBK_FIRST=1"; do
  tag=$(echo $v | tr '= ' '__')
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --layers 1 --cpu-sample 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1])
    print('$v', d['value'], d['ms_per_step'], d['stages'])
    for r in d['layers']:
        if '@64x64' in r['layer']: print('   %-40s %-30s %7.1f' % (r['layer'], r['kernel'], r['us']))
except Exception as e:
    print('$v', 'FAILED', e)
PY
done
SSDK_MB_DBG=1 timeout 200 python tools/mb_dbg.py 2>&1 | grep "mbk dbg" | head -4
( timeout 600 python -m pytest tests/test_gpu_plan_audit.py tests/test_gpu_bench_sizes.py -q -k "ssd" 2>&1 | tail -12 ) > $OUT/t_audit.log 2>&1; tail -4 $OUT/t_audit.log
