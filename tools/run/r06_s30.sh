# round 6, session 30: tiles per workgroup of the persistent 1x1 GEMM under the level lanes (SSDK_GEMMP_TILES = 9999 | 4 | 2 | 1)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s30; mkdir -p $OUT
for v in 9999 4 2 1 9999 2; do
  SSDK_GEMMP_TILES=$v timeout 600 python bench.py --cpu-sample 0 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_t$v.json 2> $OUT/fpn_t$v.err
  SSDK_GEMMP_TILES=$v timeout 600 python bench.py --cpu-sample 0 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 > $OUT/bifpn_t$v.json 2> $OUT/bifpn_t$v.err
  python - <<PY
import json
for f in ('fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/%s_t$v.json' % f).read().strip().splitlines()[-1])
        print('TILES=$v', f, d['value'], d['ms_per_step'], d.get('verified'))
    except Exception as e:
        print('TILES=$v', f, 'failed', e)
PY
done
