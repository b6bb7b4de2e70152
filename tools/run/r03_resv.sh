cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03cfgs; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-200 | head -5
for v in 1 0 1; do
SSDK_GEMM_RESV=$v timeout 300 python bench.py --cpu-sample 0 --steps 20 --warmup 5 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/fpn_resv_$v.json 2> $OUT/fpn.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/fpn_resv_$v.json") if l.startswith("{")][-1])
print("RESV=$v", d["value"], d["ms_per_step"], d["verified"])
PY
done
