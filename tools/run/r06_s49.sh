# round 6, session 49: the default bench line three times (the forward check's correlation rule with its sampling term)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s49; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  timeout 900 python bench.py > $OUT/bench_$i.json 2> $OUT/bench_$i.err; tail -1 $OUT/bench_$i.json | cut -c1-120; tail -1 $OUT/bench_$i.err | cut -c1-200
done
