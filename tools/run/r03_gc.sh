cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03cfgs; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu -k "grouped or regnet or bifpn" 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-200 | head
for v in 1 0; do
SSDK_GCONV_TILE=$v timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bench_bifpn_$v.json 2> $OUT/bifpn.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_bifpn_$v.json") if l.startswith("{")][-1])
g=[r for r in d["layers"] if r["kernel"].startswith("gconv")]
print("TILE=$v", d["value"], d["ms_per_step"], d["verified"], "gconv total us", round(sum(r["us"] for r in g)), [r["us"] for r in g][:6])
PY
done
