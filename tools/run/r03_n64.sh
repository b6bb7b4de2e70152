cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03n64
mkdir -p $OUT
for v in 1 0; do
SSDK_GEMM_N64=$v timeout 200 python bench.py --cpu-sample 0 --layers 1 > $OUT/b$v.json 2> $OUT/b$v.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b$v.json") if l.startswith("{")][-1])
print("N64=$v", d["value"], d["ms_per_step"], d["verified"], [ (r["layer"], r["kernel"], r["us"]) for r in d["layers"] if "320>256" in r["layer"]])
PY
done
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "dense or large_tile" 2>&1 | tail -2
