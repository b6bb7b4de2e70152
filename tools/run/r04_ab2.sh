# generic A/B of the headline bench WITHOUT the per-layer table (side stream etc. are off under op profiling)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04ab2; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env $envs timeout 300 python bench.py --cpu-sample 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
    print("%-10s" % "$tag", d["value"], d["ms_per_step"], d["stages"], d.get("verified"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/bench_$tag.err").read()[-300:])
PY
done
