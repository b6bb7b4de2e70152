cd $GRAFT_REPO_ROOT
SSDK_PWFLOW=2 timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pointwise_streaming" 2>&1 | tail -4
for v in "SSDK_PWFLOW=2" "SSDK_PWFLOW=1" "SSDK_PWFLOW=2" "SSDK_PWFLOW=1"; do
  echo -n "$v: "; env $v timeout 300 python bench.py --cpu-sample 0 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done
SSDK_PWFLOW=2 timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
for r in d['layers']:
    if ' k1 ' in r['layer'] and '256>' in r['layer']: print(r['layer'], r['kernel'], r['us'])" | sort | uniq -c | head -20
