cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -6 ) > $OUT/conv.log 2>&1
tail -4 $OUT/conv.log
timeout 200 python tools/gemm_probe.py head_L0 head_L1 head_L2 head_L3 > $OUT/gemm.log 2>&1
SSDK_HALO_SPLITK=0 timeout 200 python tools/gemm_probe.py head_L2 >> $OUT/gemm.log 2>&1
grep -v amdgpu $OUT/gemm.log
timeout 300 python bench.py --layers 1 --cpu-sample 0 > $OUT/bench_layers.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_layers.json")); print(d["value"], d["ms_per_step"], d["roofline"]["head_convs_mfma"])
for l in d["layers"]:
    if l["kind"] in ("head","conv") : print("%-40s %-22s %7.1f us %7.1f TF" % (l["layer"], l["kernel"], l["us"], l["TFLOPs"]))
PY
