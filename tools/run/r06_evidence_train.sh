# round 6: the training-step evidence alone (bench lines eager / captured graph / 300 px, kernel split) -> gpurun_out/ev_train
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/ev_train; mkdir -p $OUT
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train.json 2> $OUT/train.err; tail -1 $OUT/train.json | cut -c1-300
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 --graph 1 > $OUT/train_graph.json 2>> $OUT/train.err; tail -1 $OUT/train_graph.json | cut -c1-300
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 --size 300 > $OUT/train_300.json 2>> $OUT/train.err; tail -1 $OUT/train_300.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 400 > $OUT/train_kernel_split.txt
head -12 $OUT/train_kernel_split.txt; grep -c Cijk $OUT/train_kernel_split.txt
rm -rf $OUT/tr
