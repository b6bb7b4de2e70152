cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g256
mkdir -p $OUT
for v in 1 0; do
SSDK_GEMM256=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 --steps 10 --warmup 3 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_$v.json 2> $OUT/fpn_$v.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/fpn_$v.json") if l.startswith("{")][-1])
print("GEMM256=$v", d["value"], d["ms_per_step"], d["verified"])
rows=[r for r in d["layers"] if r["kernel"] in ("conv_gemm256","conv_gemm") and r["kind"]=="conv"]
import collections
agg=collections.OrderedDict()
for r in rows:
    a=agg.setdefault(r["layer"],[0,0.0,r["kernel"]]); a[0]+=1; a[1]+=r["us"]
for k,a in sorted(agg.items(), key=lambda x:-x[1][1])[:14]: print("   %-40s %-13s n=%d %7.0f us"%(k,a[2],a[0],a[1]))
PY
done
