cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ts
mkdir -p $OUT
for ts in 0 1 0 1; do
timeout 200 python bench.py --cpu-sample 0 --steps 50 --warmup 10 --tail-stream $ts > $OUT/b$ts.json 2> $OUT/b$ts.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b$ts.json") if l.startswith("{")][-1]); print("tail-stream $ts", d["value"], d["ms_per_step"], d["verified"])
PY
done
