cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "large_tile or dense or head or fpn or tower or halo or grouped" 2>&1 | tail -4 )
SSDK_H3_DBG_WG=-1 SSDK_H3_DBG=4 timeout 300 python bench.py --cpu-sample 0 --steps 1 --warmup 0 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 2>&1 >/dev/null | grep "h3 dbg\] setup" | head -4
for i in 1 2; do
for c in "" "--cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32" "--cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16"; do
  echo -n "$c: "; timeout 300 python bench.py --cpu-sample 0 $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done; done
