cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_bench_sizes.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-250 | head -20
timeout 200 python bench.py --cpu-sample 4 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["verified"], r["frac"])
print(r["decode_nms_stage"]["bench_input_in_line"])
print(r["decode_nms_stage"]["realistic_heads_in_line"])
PY
tail -2 $OUT/bench.err
