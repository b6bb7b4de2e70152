# round 6, session 20: persistent 1x1 GEMM, residual fetched behind the last k-step -- parity, stamps, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s20; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "persistent_gemm or pointwise_streaming or dense_conv or large_tile or resnet or fpn" 2>&1 | tail -8 ) > $OUT/t_gemmp.log 2>&1; tail -4 $OUT/t_gemmp.log
SSDK_GP_DBG=3 timeout 600 python bench.py --cpu-sample 0 --steps 2 --warmup 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 2>&1 >/dev/null | grep "gemmp dbg" | head -60 > $OUT/stamps.txt; head -40 $OUT/stamps.txt
for v in 0 1; do
  SSDK_GEMMP=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_g$v.json 2> $OUT/fpn_g$v.err
  SSDK_GEMMP=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 > $OUT/bifpn_g$v.json 2> $OUT/bifpn_g$v.err
  python - <<PY
import json
for f in ('fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/%s_g$v.json' % f).read().strip().splitlines()[-1])
        print('GEMMP=$v', f, d['value'], d['ms_per_step'], d.get('verified'))
        if $v:
            for l in d.get('layers') or []:
                if l['kernel'] in ('conv_gemmp',): print('    %-40s %-12s %7.1f us %7.1f TF %6.0f GB/s' % (l['layer'],l['kernel'],l['us'],l['TFLOPs'],l['GBps']))
    except Exception as e:
        print('GEMMP=$v', f, 'failed', e); print(open('$OUT/%s_g$v.err' % f).read()[-1500:])
PY
done
