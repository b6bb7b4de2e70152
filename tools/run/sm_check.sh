cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "large_tile or mobilenetv2" 2>&1 | tail -2
for v in "A=1" "SSDK_CONV_SMALLMAP_TINY=64" "SSDK_CONV_SMALLMAP_TINY=16"; do env $v timeout 200 python bench.py --cpu-sample 0 --steps 30 --layers 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], d['verified'], [(r['kernel'][:13], round(r['us'],1)) for r in d['layers'] if r['kind']=='head'])"; done
