cd $GRAFT_REPO_ROOT
for v in "X=1" "SSDK_SMALL_LEVEL_PIXELS=65536" "SSDK_SMALL_LEVEL_PIXELS=4096" "X=1" "SSDK_SMALL_LEVEL_PIXELS=65536"; do
for c in "--cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32" "--cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16"; do
  echo -n "$v $c: "; env $v timeout 300 python bench.py --cpu-sample 0 $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done; done
