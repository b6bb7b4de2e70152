# round 6, session 26: kernel split of the training step (what is left on the libraries)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s26; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 200 > $OUT/split.txt
rm -rf $OUT/tr
head -75 $OUT/split.txt | cut -c1-150
