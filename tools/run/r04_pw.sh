cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04pw; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pointwise_streaming or dense_conv or half_resolution or activation_after or resnet_plan or fpn_bifpn" 2>&1 | tail -12 ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for tag in pw; do
  if [ $tag = nopw ]; then export SSDK_PWFLOW=0; else unset SSDK_PWFLOW; fi
  timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_$tag.json 2> $OUT/fpn_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/fpn_$tag.json") if l.startswith("{")][-1])
    t={}
    for r in d["layers"]:
        k=(r["layer"],r["kernel"]); a=t.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=r["us"]
    pw=sum(v[1] for k,v in t.items() if " k1 " in k[0])
    print("$tag", d["value"], d["ms_per_step"], "1x1 total us %.0f" % pw)
    for k,v in sorted(t.items(), key=lambda kv:-kv[1][1]):
        if " k1 " in k[0]: print("    %-34s %-14s x%d %7.1f" % (k[0],k[1],v[0],v[1]))
except Exception as e:
    print("$tag FAILED", e)
PY
done
unset SSDK_PWFLOW
timeout 300 python bench.py --cpu-sample 0 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bifpn_pw.json 2> $OUT/bifpn_pw.err
SSDK_PWFLOW=0 timeout 300 python bench.py --cpu-sample 0 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bifpn_nopw.json 2> $OUT/bifpn_nopw.err
python - <<PY
import json
for f in ["bifpn_pw","bifpn_nopw"]:
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
