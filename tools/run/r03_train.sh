cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03train
mkdir -p $OUT
for v in 2 3 2 3; do
  SSDK_BN_FLAT=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 4 > $OUT/train_flat$v.json 2> $OUT/train_flat$v.err
  echo "BN_FLAT=$v $(tail -1 $OUT/train_flat$v.json | cut -c1-200)"
done
