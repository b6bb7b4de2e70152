# generic A/B helper of round 4: block tests, then bench.py --layers 1 under each "tag ENV=.." argument
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04ab; mkdir -p $OUT
if [ -n "$PYTEST_K" ]; then
  ( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "$PYTEST_K" 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
fi
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env $envs timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
    print("%-10s" % "$tag", d["value"], d["ms_per_step"], " ".join("%.1f" % r["us"] for r in d["layers"]))
except Exception as e:
    print("$tag FAILED", e)
PY
done
