# A/B of one environment switch inside one box: tools/run/ab_env.sh VAR  (runs VAR=0, VAR=1, VAR=0, VAR=1)
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do env $1=$v timeout 200 python bench.py --layers 1 --cpu-sample 0 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1=$v', d['value'], d['stages']['forward_ms'], [round(r['us'],1) for r in d['layers'] if r['kind']!='mbconv'])"; done
