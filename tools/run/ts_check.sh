cd $GRAFT_REPO_ROOT
for v in "A=1" "SSDK_SPLITK_WGS=256" "SSDK_SPLITK=0" "A=2" "SSDK_SPLITK_WGS=256"; do env $v timeout 200 python bench.py --cpu-sample 0 --steps 30 --layers 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], d['verified'], [(r['kernel'][:10], round(r['us'],1)) for r in d['layers'] if r['kind']=='conv'])"; done
