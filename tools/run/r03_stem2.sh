cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03cfgs; mkdir -p $OUT
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn_resnet50_640.json 2> $OUT/fpn.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03cfgs/bench_fpn_resnet50_640.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["verified"], [ (r["layer"], r["kernel"], r["us"]) for r in d["layers"] if r["kernel"] in ("stem7","maxpool3x3s2")])
PY
