cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "not soak" 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout 120 python tools/gemm_probe.py extras_3x3s2 head_L2 extras_1x1 head_L3 2>&1 | grep -v amdgpu.ids | cut -c1-75; }
run A=1
run SSDK_SPLITK=0
run SSDK_SPLITK_WGS=256
timeout 200 python bench.py --layers 1 --cpu-sample 0 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['stages']['forward_ms'], [round(r['us'],1) for r in d['layers'] if r['kind']!='mbconv'])"
