cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_bench_sizes.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -30
timeout 200 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_layers.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_layers.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["verified"], d["roofline"]["head_convs_mfma"]["frac"], d["roofline"]["head_convs_mfma"]["ms"])
for r in d["layers"]:
    if r["kind"]=="head": print(r)
PY
