cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "large_tile or short_k or split_heads or small_map" 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -30
timeout 200 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_layers.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_layers.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["verified"], d["roofline"]["head_convs_mfma"]["frac"], d["roofline"]["head_convs_mfma"]["ms"])
for r in d["layers"]:
    if r["kind"]=="head": print(r)
PY
