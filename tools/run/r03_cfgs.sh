cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03cfgs
mkdir -p $OUT
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn_resnet50_640.json 2> $OUT/fpn.err
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bench_bifpn_regnetx008_896.json 2> $OUT/bifpn.err
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
for f in ["bench_fpn_resnet50_640","bench_bifpn_regnetx008_896","bench"]:
    d=json.loads([l for l in open("$OUT/%s.json"%f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["verified"], r["frac"], r["decode_nms_stage"]["realistic_heads_in_line"]["stage_frac"], r["decode_nms_stage"]["realistic_heads_in_line"]["scan_frac"], r.get("head_convs_mfma",{}).get("frac"))
PY
