cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -x -k "mobilenetv2 or ssd_mnv2 or detector" 2>&1 | tail -3
for v in "SSDK_XPAIR=0 SSDK_HEAD_BALANCE=0" "SSDK_XPAIR=1 SSDK_HEAD_BALANCE=0" "SSDK_XPAIR=0 SSDK_HEAD_BALANCE=1" "SSDK_XPAIR=1 SSDK_HEAD_BALANCE=1"; do env $v timeout 200 python bench.py --cpu-sample 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], d['verified'])"; done
