# round 6, session 38: steady-state kernel split of the training step at HEAD (BatchNorm finalize folded), every kernel listed
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s38
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 88 400 > $OUT/split.txt
rm -rf $OUT/tr
cd $GRAFT_REPO_ROOT && timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 > $OUT/train.json; cat $OUT/train.json | cut -c1-200
