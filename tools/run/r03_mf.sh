cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03mf
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "flow or stem or inverted or fused_block" 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -20
timeout 200 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_layers.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_layers.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["verified"])
for r in d["layers"][:4]: print(r)
PY
