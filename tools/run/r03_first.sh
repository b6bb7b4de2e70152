# Round 3, first GPU call: the new bench-size tests, the kernels touched by the advice fixes, the never-run flat2 BN kernel,
# and the default bench line with the oracle-checked `verified`.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a
mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_gpu_bench_sizes.py -q -x 2>&1 | tail -25 ) > $OUT/sizes.log 2>&1
( time timeout 400 python -m pytest tests/test_gpu_conv.py -q -k "block or stem or mbconv or extra" 2>&1 | tail -8 ) > $OUT/conv.log 2>&1
( SSDK_BN_FLAT=3 timeout 300 python -m pytest tests/test_gpu_train.py -q -k "batchnorm or bn" 2>&1 | tail -8 ) > $OUT/bn_flat3.log 2>&1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
SSDK_TAIL_STAMPS=1 timeout 200 python tools/scan_probe.py > $OUT/scan_probe.log 2>&1
tail -12 $OUT/sizes.log; tail -4 $OUT/conv.log; tail -4 $OUT/bn_flat3.log
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); print("bench", d["value"], d["ms_per_step"], d.get("verified")); print(d["config"]["verification"])
except Exception as e: print("bench FAILED", e)
PY
tail -5 $OUT/bench.err; tail -20 $OUT/scan_probe.log
