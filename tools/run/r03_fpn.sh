cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03fpn
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -20
for v in 1 2; do
SSDK_CONV3X3_SHORT=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_$v.json 2> $OUT/fpn_$v.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/fpn_$v.json") if l.startswith("{")][-1])
print("SHORT=$v", d["value"], d["ms_per_step"], d["verified"], d["roofline"]["head_convs_mfma"]["frac"], d["roofline"]["head_convs_mfma"]["ms"])
PY
done
