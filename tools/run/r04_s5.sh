# round 4, session 5: decode stage as two launches (tail2_kernel) -- every box test, probe, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s5; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_box.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -12 ) > $OUT/pytest_box.log 2>&1
tail -3 $OUT/pytest_box.log
( timeout 600 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_nets.py -x -q -m gpu -k "all_ties or detector" 2>&1 | tail -6 ) > $OUT/pytest_b2.log 2>&1
tail -2 $OUT/pytest_b2.log
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "^SURVEY|^all equal|^trained|nmswalk wg0|levelsel wg" > $OUT/probe_tail2.txt
SSDK_TAIL2=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "^SURVEY|^all equal|^trained" > $OUT/probe_3launch.txt
cat $OUT/probe_tail2.txt; echo ---; cat $OUT/probe_3launch.txt
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sample 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1]); r=d["roofline"]["decode_nms_stage"]
    print("$tag", d["value"], d["ms_per_step"], d["verified"], r["bench_input_in_line"]["stage_ms"], r["realistic_heads_in_line"]["stage_ms"], r["realistic_heads_in_line"]["stage_frac"], r["realistic_heads_in_line"]["kernels_ms"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run three
run tail2 SSDK_TAIL2=1
run threeb
tail -3 $OUT/bench_tail2.err
