cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03cfgs; mkdir -p $OUT
for v in 32 96; do
SSDK_HALO_MIN_COUT=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 --steps 10 --warmup 3 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_loc_$v.json 2> $OUT/fpn.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/fpn_loc_$v.json") if l.startswith("{")][-1])
print("MINCO=$v", d["value"], d["ms_per_step"], d["verified"], [(r["layer"][5:], r["kernel"], r["us"]) for r in d["layers"] if "256>36" in r["layer"]])
PY
done
