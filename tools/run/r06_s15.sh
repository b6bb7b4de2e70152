# round 6, session 15: persistent form of the halo kernel (SSDK_HALO_PERSIST, default 1) -- parity, then A/B on the three configs
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s15; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_conv.py -q -x -k "not soak" 2>&1 | tail -6 ) > $OUT/t_conv.log 2>&1; tail -6 $OUT/t_conv.log
for v in 0 1 0 1; do
  SSDK_HALO_PERSIST=$v timeout 400 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_p$v.json 2> $OUT/bench_p$v.err
  SSDK_HALO_PERSIST=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_p$v.json 2> $OUT/fpn_p$v.err
  SSDK_HALO_PERSIST=$v timeout 600 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 > $OUT/bifpn_p$v.json 2> $OUT/bifpn_p$v.err
  python - <<PY
import json
for f in ('bench','fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/%s_p$v.json' % f).read().strip().splitlines()[-1])
        h=d['roofline'].get('head_convs_mfma') or {}
        halo=sum(l['us'] for l in d.get('layers') or [] if l['kernel']=='conv3x3_halo')
        print('PERSIST=$v', f, d['value'], d['ms_per_step'], d.get('verified'), 'heads frac', h.get('frac'), 'halo us', round(halo,1))
    except Exception as e:
        print('PERSIST=$v', f, 'failed', e)
PY
done
