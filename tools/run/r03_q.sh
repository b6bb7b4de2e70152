cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --cpu-sample 0 --steps 50 --warmup 10 $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/$tag.json") if l.startswith("{")][-1]); print("$tag", d["value"], d["ms_per_step"], d.get("verified"))
except Exception as e: print("$tag failed", e)
PY
}
EXTRA="" run inline A=1
EXTRA="" run side SSDK_SIDE_STREAM=1
EXTRA="--graph 1" run graph A=1
EXTRA="--graph 1" run graph_side SSDK_SIDE_STREAM=1
