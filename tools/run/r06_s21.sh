# round 6, session 21: the rest of the GPU suite behind the dispatch expectation fixed after session "full"; the persistent GEMM
# on shorter K (SSDK_GEMMP_MIN_CIN = 128 / 64) A/B on the FPN / BiFPN configurations
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s21; mkdir -p $OUT
( timeout 3000 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_train.py 2>&1 | tail -8 ) > $OUT/t_rest.log 2>&1; tail -4 $OUT/t_rest.log
for v in 256 128 64; do
  SSDK_GEMMP_MIN_CIN=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_c$v.json 2> $OUT/fpn_c$v.err
  SSDK_GEMMP_MIN_CIN=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 > $OUT/bifpn_c$v.json 2> $OUT/bifpn_c$v.err
  python - <<PY
import json
for f in ('fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/%s_c$v.json' % f).read().strip().splitlines()[-1])
        print('MIN_CIN=$v', f, d['value'], d['ms_per_step'], d.get('verified'))
        for l in d.get('layers') or []:
            if ' k1 ' in l['layer'] and (' 64>' in l['layer'] or ' 128>' in l['layer']): print('    %-40s %-12s %7.1f us %7.1f TF %6.0f GB/s' % (l['layer'],l['kernel'],l['us'],l['TFLOPs'],l['GBps']))
    except Exception as e:
        print('MIN_CIN=$v', f, 'failed', e); print(open('$OUT/%s_c$v.err' % f).read()[-1500:])
PY
done
