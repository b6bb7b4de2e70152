# Validation after the depthwise training kernels changed (run through gpurun): tools/run/final_validation_r02b.sh <tag>
# GPU test-suite, smoke(), the default bench line, the per-layer depthwise table, the training step and its kernel split.
cd $GRAFT_REPO_ROOT
TAG=${1:-w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python tools/dw_probe.py > $OUT/dw_probe.txt 2>&1
timeout 300 python tools/bench_train.py --steps 8 --warmup 4 2>/dev/null | tail -1 > $OUT/train_step.json
bash tools/run/train_prof.sh > $OUT/train_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/trainprof/tail.txt $OUT/train_kernel_split.txt
tail -3 $OUT/pytest.log; tail -1 $OUT/smoke.log
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("bench", d["value"], d["ms_per_step"], d.get("verified"), d["roofline"]["frac"])
PY
cut -c1-160 $OUT/train_step.json; tail -1 $OUT/dw_probe.txt; tail -1 $OUT/train_kernel_split.txt
