# round 6, session 24: head pairs with the per-module backward -- parity, A/B, kernel split
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s24; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair or whole_step or training_module or graphed" 2>&1 | tail -6 ) > $OUT/t_train.log 2>&1; tail -6 $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_HEAD_PAIR=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_hp$v.json 2> $OUT/train_hp$v.err
  tail -1 $OUT/train_hp$v.json | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
SSDK_HEAD_PAIR=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 200 > $OUT/split_hp1.txt
rm -rf $OUT/tr
grep -i -E "igemm|transpose|SubTensor|conv3x3|smallmap|pack_conv|CatArray|copy_kernel|kernel time" $OUT/split_hp1.txt | head -40
