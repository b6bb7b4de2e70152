cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --layers 1 --cpu-sample 0 --steps 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], [round(r['us'],1) for r in d['layers'] if r['kind']=='mbconv'])"; }
run A=1
run SSDK_MB_HC=64
run SSDK_MB_HC=32
run SSDK_MB_8X16=2
run SSDK_MB_8X16=0
run SSDK_MB_RESIDENT=0
run SSDK_MB_TS=8
run SSDK_MB_LEAN=0
