# round 5, session 2: ssdk_mbk.hip v2 (hazard guards, constants in the image, two items per workgroup)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $OUT
rm -f gpurun_out/net_report.txt
( timeout 200 python tools/mbk_debug.py 2>&1 | grep -v Warn | tail -40 ) > $OUT/dbg.log 2>&1; head -3 $OUT/dbg.log; tail -6 $OUT/dbg.log
( timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "row_pair" 2>&1 | tail -15 ) > $OUT/t_mbk.log 2>&1; tail -4 $OUT/t_mbk.log
for v in "SSDK_MBK=0" "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=2" "SSDK_MBK_NW=4 SSDK_MBK_ITEMS=1" "SSDK_MBK_NW=3 SSDK_MBK_ITEMS=2" "SSDK_MBK_NW=6 SSDK_MBK_ITEMS=1"; do
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
( timeout 900 python -m pytest tests/test_gpu_plan_audit.py tests/test_gpu_bench_sizes.py tests/test_gpu_nets.py -q 2>&1 | tail -30 ) > $OUT/t_nets.log 2>&1; tail -8 $OUT/t_nets.log
cp gpurun_out/net_report.txt $OUT/ 2>/dev/null
cp gpurun_out/plan_audit_*.txt $OUT/ 2>/dev/null
