# round 6, session 23: kernel split of the training step with and without the head pairs
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s23; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $OUT/tr
  SSDK_HEAD_PAIR=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 > $OUT/split_hp$v.txt
  rm -rf $OUT/tr
  echo "== HEAD_PAIR=$v"; grep -i -E "igemm|transpose|SubTensor|conv3x3|smallmap|pack_conv|CatArray|copy_kernel|kernel time" $OUT/split_hp$v.txt | head -40
done
