cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04dbg; mkdir -p $OUT
SSDK_MB_DBG=1 timeout 300 python bench.py --cpu-sample 0 --layers 1 --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
grep "mbconv dbg" $OUT/bench.err | awk '!seen[$3 $4 $5 $6]++' | cut -c1-900
