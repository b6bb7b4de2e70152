# level lanes A/B: the new test, then FPN-R50@640 and BiFPN@896 with the small levels on the side stream / in line
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04lanes; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "side_stream" 2>&1 | tail -12 ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for tag in lanes inline lanes2 inline2; do
  if [ "${tag#inline}" != "$tag" ]; then export SSDK_LEVEL_LANES=0; else unset SSDK_LEVEL_LANES; fi
  timeout 300 python bench.py --cpu-sample 0 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_$tag.json 2> $OUT/fpn_$tag.err
  timeout 300 python bench.py --cpu-sample 0 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bifpn_$tag.json 2> $OUT/bifpn_$tag.err
  python - <<PY
import json
for f in ["fpn_$tag","bifpn_$tag"]:
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["stages"], d.get("verified"))
    except Exception as e:
        print(f, "FAILED", e)
PY
done
