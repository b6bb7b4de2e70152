cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04lanes; mkdir -p $OUT
for v in "SSDK_OPS_TRACE=1"; do
  env $v timeout 300 python bench.py --cpu-sample 0 --steps 2 --warmup 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/dbg.json 2> $OUT/dbg_$v.err
  grep "plan\] head\|lane [12]" $OUT/dbg_$v.err | tail -60
done
