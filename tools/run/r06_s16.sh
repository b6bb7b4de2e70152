# round 6, session 16: tiles per workgroup of the persistent halo kernel (SSDK_HALO_PERSIST = 0 first form | 1 | 2 | 3 | 9999)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s16; mkdir -p $OUT
for v in 0 1 2 3 9999 0 1; do
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
