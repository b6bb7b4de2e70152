# round 5, session 18: the halo kernel's transposed product for NHWC outputs (SSDK_H3_TR=0 | 1)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s18; mkdir -p $OUT
for v in 0 1; do
  echo "== SSDK_H3_TR=$v"
  SSDK_H3_TR=$v SSDK_H3_DBG=1 SSDK_H3_DBG_WG=300 timeout 200 python tools/gemm_probe.py tower_P3 2>&1 | grep -E "h3 dbg\] setup|TF/s" | cut -c1-160
  SSDK_H3_TR=$v timeout 200 python tools/gemm_probe.py tower_P3 tower_cls tower_P5 2>&1 | grep -E "TF/s" | cut -c1-160
done
( timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_plan_audit.py -q -x -k "not soak" 2>&1 | tail -4 ) > $OUT/t.log 2>&1; tail -3 $OUT/t.log
for v in 0 1; do
  SSDK_H3_TR=$v timeout 400 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 0 > $OUT/fpn_$v.json 2> $OUT/fpn_$v.err
  SSDK_H3_TR=$v timeout 400 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 --cpu-sample 0 > $OUT/bifpn_$v.json 2> $OUT/bifpn_$v.err
  python - <<PY
import json
for t in ('fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/%s_$v.json' % t).read().strip().splitlines()[-1])
        print('TR=$v', t, d['value'], d['ms_per_step'], d['verified'])
    except Exception as e:
        print('TR=$v', t, 'FAILED', e); print(open('$OUT/%s_$v.err' % t).read()[-600:])
PY
done
