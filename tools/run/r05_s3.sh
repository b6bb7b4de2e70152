# round 5, session 3: ssdk_mbk.hip with the deeper weight prefetch; full GPU suite
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s3; mkdir -p $OUT
rm -f gpurun_out/net_report.txt
( timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "row_pair" 2>&1 | tail -15 ) > $OUT/t_mbk.log 2>&1; tail -3 $OUT/t_mbk.log
for v in "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=2" "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=1" "SSDK_MBK_NW=3 SSDK_MBK_ITEMS=1" "SSDK_MBK_NW=6 SSDK_MBK_ITEMS=1"; do
  tag=$(echo $v | tr '= ' '__')
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --layers 1 --cpu-sample 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1])
    print('$v', d['value'], d['ms_per_step'], d['stages'])
    for r in d['layers']:
        if '@16x16' in r['layer'] and 'mbconv' in r['layer']: print('   %-40s %-30s %7.1f' % (r['layer'], r['kernel'], r['us']))
except Exception as e:
    print('$v', 'FAILED', e)
PY
done
for v in "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=2" "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=1"; do
  env $v SSDK_MB_DBG=1 timeout 200 python tools/mb_dbg.py 2>&1 | grep "mbk dbg" | tail -3
done
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > $OUT/all.log 2>&1; tail -12 $OUT/all.log
cp gpurun_out/net_report.txt $OUT/ 2>/dev/null
cp gpurun_out/plan_audit_*.txt $OUT/ 2>/dev/null
