# Round-end validation on one MI355X box (run through gpurun): tools/run/final_validation_r02.sh <tag>
# GPU test-suite, smoke(), the default bench line (+ per-layer table, tail stream on/off), the other inference configs,
# the training step.  Profiles of the default command: tools/run/profile_r02.sh (separate call).
cd $GRAFT_REPO_ROOT
TAG=${1:-v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --layers 1 --cpu-sample 0 > $OUT/bench_layers.json 2>> $OUT/bench.err
timeout 300 python bench.py --tail-stream 0 --cpu-sample 0 > $OUT/bench_inline.json 2>> $OUT/bench.err
timeout 300 python bench.py --graph 1 --cpu-sample 0 > $OUT/bench_graph.json 2>> $OUT/bench.err
timeout 300 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 0 --steps 10 --layers 1 > $OUT/bench_fpn.json 2>> $OUT/bench.err
timeout 300 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 --cpu-sample 0 --steps 10 > $OUT/bench_bifpn_fp16_graph.json 2>> $OUT/bench.err
timeout 300 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --cpu-sample 0 --steps 10 --layers 1 > $OUT/bench_bifpn.json 2>> $OUT/bench.err
timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1 > $OUT/train_step.json
tail -3 $OUT/pytest.log; tail -1 $OUT/smoke.log
for f in bench bench_inline bench_graph bench_fpn bench_bifpn_fp16_graph bench_bifpn; do python - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", d["value"], d["ms_per_step"], d.get("verified"), d["roofline"]["frac"] if d.get("roofline") else None)
except Exception as e: print("$f FAILED", e)
PY
done
cat $OUT/train_step.json | cut -c1-160; tail -3 $OUT/bench.err
