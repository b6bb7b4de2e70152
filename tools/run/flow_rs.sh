cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "inverted or 16x16 or mobilenetv2 or stem_block" 2>&1 | tail -2
for v in 0 0; do SSDK_MB_FLOW_RS=$v timeout 200 python bench.py --layers 1 --cpu-sample 0 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('RS=$v', d['value'], d['ms_per_step'], d['verified'], [(r['kernel'][:8], round(r['us'],1)) for r in d['layers'][:5]])"; done
