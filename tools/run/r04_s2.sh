# round 4, session 2: the split register-flow kernel (blocks 4-7) -- parity, then A/B of its two knobs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04s2; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "split_register or inverted_residual or mobilenetv2_plan" 2>&1 | tail -15 ) > $OUT/pytest_blocks.log 2>&1
tail -4 $OUT/pytest_blocks.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
    print("$tag", d["value"], d["ms_per_step"], " ".join("%s:%.1f" % (r["kernel"][:14], r["us"]) for r in d["layers"][:8]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run split_xb1 SSDK_MB_SPLIT=1
run split_xb2 SSDK_MB_SPLIT=1 SSDK_MB_SPLIT_XB=2
run split_rs8 SSDK_MB_SPLIT=1 SSDK_MB_SPLIT_RS=8
run split_rs32 SSDK_MB_SPLIT=1 SSDK_MB_SPLIT_RS=32
run nosplit SSDK_MB_SPLIT=0
tail -2 $OUT/bench_split_xb1.err
