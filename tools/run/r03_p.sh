cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_bench_sizes.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|^E  " > $OUT/conv_fail.log
cut -c1-220 $OUT/conv_fail.log | head -60
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 30 --warmup 5 > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep -h "smallmap\|xpair" $OUT/kernel_stats.csv | cut -d, -f1-9 | cut -c1-150
tail -1 $OUT/stats.log | cut -c1-300
