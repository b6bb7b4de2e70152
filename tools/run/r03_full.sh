# full GPU validation: test-suite, smoke, default bench (+ other configs), probe
cd $GRAFT_REPO_ROOT
TAG=${1:-r03full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/probe.log 2>&1
tail -6 $OUT/pytest.log; tail -2 $OUT/smoke.log
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); r=d["roofline"]; st=r["decode_nms_stage"]
    print("bench", d["value"], d["ms_per_step"], d.get("verified"), "scan frac", r["frac"], r["avg_launch_ms"])
    print(" stage bench-input", st["bench_input_in_line"]); print(" stage realistic", st["realistic_heads_in_line"])
    print(" heads", r.get("head_convs_mfma")); print(" body", r.get("backbone_by_time")); print(" cpu", d.get("cpu_baseline", {}).get("value"))
    print(d["config"]["verification"][-420:])
except Exception as e: print("bench FAILED", e)
PY
tail -4 $OUT/bench.err
grep -A1 "SURVEY\|all equal" $OUT/probe.log | cut -c1-200
