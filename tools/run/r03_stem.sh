cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " | cut -c1-200 | head
bash tools/run/r03_cfgs.sh
python - <<PY
import json
for f in ["bench_fpn_resnet50_640","bench_bifpn_regnetx008_896"]:
    d=json.loads([l for l in open("gpurun_out/r03cfgs/%s.json"%f) if l.startswith("{")][-1])
    print([ (r["layer"], r["kernel"], r["us"]) for r in d["layers"] if r["kernel"] in ("stem7","conv_first")])
PY
