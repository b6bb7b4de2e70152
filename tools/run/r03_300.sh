cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_300; mkdir -p $OUT
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/ssd_mobilenetv2_300.yml --batch 64 > $OUT/b300.json 2> $OUT/b300.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b300.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["verified"], d["roofline"]["decode_nms_stage"]["realistic_heads_in_line"])
for r in d["layers"]:
    if r["kind"]!="mbconv": print("   %-46s %-20s %7.1f us %7.1f TF/s"%(r["layer"][:46], r["kernel"][:20], r["us"], r["TFLOPs"]))
PY
tail -2 $OUT/b300.err
