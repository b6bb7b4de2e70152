cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "large_tile or dense or gemm or residual or resnet or fpn" 2>&1 | tail -3 )
for i in 1 2; do
for c in "--cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32" "--cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16"; do
  echo -n "$c: "; timeout 300 python bench.py --cpu-sample 0 $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done; done
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 2>/dev/null | tail -1 | python -c "
import sys,json,collections; d=json.loads(sys.stdin.read())
t=collections.OrderedDict()
for r in d['layers']:
    if r['kernel'].startswith('conv_gemm256'):
        a=t.setdefault((r['layer'],r['kernel']),[0,0.0]); a[0]+=1; a[1]+=r['us']
for k,v in sorted(t.items(),key=lambda kv:-kv[1][1])[:6]: print(k[0],k[1],v[0],round(v[1],1))"
