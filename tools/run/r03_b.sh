cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b
mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_gpu_bench_sizes.py -q 2>&1 | tail -40 ) > $OUT/sizes.log 2>&1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -30 $OUT/sizes.log
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); print("bench", d["value"], d["ms_per_step"], d.get("verified")); print(d["config"]["verification"])
except Exception as e: print("bench FAILED", e)
PY
tail -40 $OUT/bench.err
