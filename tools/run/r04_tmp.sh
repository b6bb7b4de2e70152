cd $GRAFT_REPO_ROOT
for c in "--cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1" "--cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --graph 1" "--graph 1" "--tail-stream 1"; do
  echo -n "$c: "; timeout 300 python bench.py --cpu-sample 0 $c 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])" || tail -3 /tmp/err.txt
done
